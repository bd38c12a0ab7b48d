#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes of bench.py (FETCH_SIZE in one, WRITE_SIZE in the other: they do not fit one
pass on gfx950, MI355X_MICROARCH.md §rocprofv3 PMC slots) to the HBM-side bytes per launch of the dominant kernel.

    python tools/pmc_traffic.py --fetch <..._counter_collection.csv> --write <..._counter_collection.csv> \
        --key c2_B1024 --out profiles/traffic_gemm_nt.json [--match gemm_nt_v]

Corrections exactly as the guide's §HBM prescribes: the counters are in KiB per dispatch; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads (16 B / lane loads and LDS-DMA alike) so it is doubled; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores.  Infinity-Cache hits are counted by both, so `bytes_per_launch` is an upper
bound on HBM traffic.  bench.py copies the entry into `roofline.traffic`."""
import argparse
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)


def per_launch(path, counter, match):
    tot, n = 0.0, 0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter or match not in row.get("Kernel_Name", ""):
                continue
            tot += float(row["Counter_Value"])
            n += 1
    if n == 0:
        raise SystemExit(f"{path}: no dispatches of *{match}* with counter {counter}")
    return tot / n * 1024.0, n


def source_hash():
    """bench.gemm_source_hash(): the entry is only valid for the GEMM sources it was measured on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.gemm_source_hash()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--key", required=True, help="workload key bench.py looks up, e.g. c2_B1024")
    ap.add_argument("--out", default="profiles/traffic_gemm_nt.json")
    ap.add_argument("--match", default="gemm_nt_v")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    fetch_raw, nf = per_launch(a.fetch, "FETCH_SIZE", a.match)
    write, nw = per_launch(a.write, "WRITE_SIZE", a.match)
    entry = {"bytes_per_launch": round(2.0 * fetch_raw + write), "fetch_bytes_per_launch_x2": round(2.0 * fetch_raw),
             "fetch_size_raw_bytes_per_launch": round(fetch_raw), "write_bytes_per_launch": round(write),
             "dispatches_fetch_pass": nf, "dispatches_write_pass": nw, "kernel_match": a.match,
             "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB per dispatch "
                       "averaged over the kernel's dispatches; FETCH_SIZE x2 (gfx950 reports half of wide reads), "
                       "WRITE_SIZE exact; includes Infinity-Cache hits",
             "command": a.cmd, "source_sha16": source_hash()}
    data = {}
    if os.path.exists(a.out):
        with open(a.out) as f:
            data = json.load(f)
    data[a.key] = entry
    with open(a.out, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps({a.key: entry}))


if __name__ == "__main__":
    sys.exit(main())
