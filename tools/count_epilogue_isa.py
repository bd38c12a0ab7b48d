#!/usr/bin/env python3
"""Count the instructions a kernel executes after its last MFMA (= the epilogue of the persistent GEMM kernels), by class.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only clip_dplm_amd/csrc/gemm_nt_v3.hip -o /tmp/v3.s
    python tools/count_epilogue_isa.py /tmp/v3.s _ZN12_GLOBAL__N_117gemm_nt_v3_kernelILi7EEEvNS_6ParamsE      (mode 7 = GELU_D8)"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
# epilogue = after the last v_mfma
last_mfma = max(i for i, l in enumerate(body) if "v_mfma" in l)
epi = [l.strip().split()[0] for l in body[last_mfma + 1:] if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
c = collections.Counter(epi)
tot = sum(c.values())
groups = collections.Counter()
for k, v in c.items():
    if k.startswith("v_pk_"): groups["v_pk_*"] += v
    elif k in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"): groups["transcendental"] += v
    elif k.startswith("v_cvt"): groups["v_cvt*"] += v
    elif k.startswith("v_"): groups["other VALU"] += v
    elif k.startswith("ds_"): groups["LDS"] += v
    elif k.startswith("buffer_") or k.startswith("global_"): groups["VMEM"] += v
    elif k.startswith("s_"): groups["SALU/ctl"] += v
    else: groups["other"] += v
print("instructions after the last MFMA:", tot)
for k, v in groups.most_common(): print(f"  {k:16s} {v}")
print("  top:", ", ".join(f"{k} {v}" for k, v in c.most_common(18)))
