#!/usr/bin/env python3
"""Print a rocprofv3 --pmc counter_collection.csv as run-length groups of consecutive dispatches of the kernels that
match, in dispatch order (a microbenchmark launches one shape many times, then the next): mean counter value per run.

    python tools/pmc_runs.py <counter_collection.csv> <COUNTER> <kernel-substring> [scale]
"""
import csv
import sys

csv.field_size_limit(1 << 30)
path, counter, match = sys.argv[1:4]
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == counter and match in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
rows.sort()
runs = []
for _, g, v in rows:
    if runs and runs[-1][0] == g:
        runs[-1][1].append(v)
    else:
        runs.append((g, [v]))
for g, vs in runs:
    print(f"grid {g:9d}  n {len(vs):4d}  mean {sum(vs) / len(vs) * scale:14.1f}  min {min(vs) * scale:14.1f}  max {max(vs) * scale:14.1f}")
