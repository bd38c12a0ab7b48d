#!/usr/bin/env python3
"""Reduce one or more rocprofv3 --pmc counter_collection.csv files to a per-kernel table (mean per dispatch over the
dispatches of each kernel whose name contains the match string; the first `--skip` dispatches of each kernel are
warm-up and dropped).

    python3 tools/pmc_table.py --match attn_ [--skip 1] a_counter_collection.csv b_counter_collection.csv ...

Prints markdown: one row per kernel, one column per counter, plus derived columns when their inputs are present:
  frac_*      = counter / SQ_WAVE_CYCLES (SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles like SQ_WAVE_CYCLES)
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES or GRBM_GUI_ACTIVE-derived CU cycles)
  fetch_GB    = 2 x FETCH_SIZE KiB (gfx950 reports half of wide reads, MI355X_MICROARCH.md §HBM), write_GB = WRITE_SIZE
  l2_hit      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"""
import argparse
import csv
import re
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("clipk::", "")
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--match", default="attn_")
    ap.add_argument("--skip", type=int, default=1)
    a = ap.parse_args()
    vals = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [(dispatch, value)]
    durs = defaultdict(dict)                               # kernel -> {dispatch: us} (under the profiler: slower)
    for path in a.files:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name", "")
                if a.match not in k:
                    continue
                vals[short(k)][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                if "Start_Timestamp" in r:
                    durs[short(k)][(path, int(r["Dispatch_Id"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    counters = sorted({c for k in vals for c in vals[k]})
    means = {}
    for k in vals:
        means[k] = {}
        for c, lst in vals[k].items():
            lst.sort()
            lst = lst[a.skip:] if len(lst) > a.skip else lst
            means[k][c] = sum(v for _, v in lst) / len(lst)
    derived = []
    for k, m in means.items():
        if durs[k]:
            d = sorted(durs[k].values())
            m["us_profiled(median)"] = d[len(d) // 2]
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in list(m):
                if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE_INST"):
                    m["frac_" + c[3:]] = m[c] / wc
        if "FETCH_SIZE" in m:
            m["fetch_GB(x2)"] = 2.0 * m["FETCH_SIZE"] * 1024 / 1e9
        if "WRITE_SIZE" in m:
            m["write_GB"] = m["WRITE_SIZE"] * 1024 / 1e9
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            m["l2_hit"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
            m["mfma_busy/cu_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CU_CYCLES"]
        if "GRBM_GUI_ACTIVE" in m:
            m["gui_active/8"] = m["GRBM_GUI_ACTIVE"] / 8.0
        derived = sorted({c for mm in means.values() for c in mm if c not in counters})
    cols = counters + derived
    kernels = sorted(means)
    print("| metric (mean per dispatch) | " + " | ".join(kernels) + " |")
    print("|---|" + "---|" * len(kernels))
    for c in cols:
        row = []
        for k in kernels:
            v = means[k].get(c)
            if v is None:
                row.append("-")
            elif c.startswith("frac_") or c in ("l2_hit", "mfma_busy/cu_busy") or "GB" in c:
                row.append(f"{v:.3f}")
            else:
                row.append(f"{v:.4g}")
        print(f"| {c} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
