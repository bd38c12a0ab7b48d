#!/usr/bin/env python3
"""Where does a K-step of the 256 x 256 weight-gradient kernel go?  (run on the GPU box)

    make tools/probes/libwgrad_trace.so && python tools/exp_wgrad_trace.py [N K M]

A -DCLIPK_WGRAD_TRACE build of gemm_wgrad_v3.hip stamps the cycle counter after each of the 8 barriers of steps 8..11
in workgroup 0, for wave 0 (n-wave group 0) and wave 4 (group 1, one barrier behind).  Printed: cycles between
consecutive barriers; the barrier after which a group's MFMA block runs is marked."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", os.environ.get("TRACE_LIB", "libwgrad_trace.so")))


class Args(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("lddy", C.c_long), ("X", C.c_void_p), ("ldx", C.c_long), ("slab", C.c_void_p),
                ("bslab", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ntn", C.c_int),
                ("ntk", C.c_int), ("splits", C.c_int), ("m_per_split", C.c_int)]


N, K, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 480, 262144)
SCHED = int(sys.argv[4]) if len(sys.argv) > 4 else 3          # 3: 8-phase schedule (stamps after its 8 barriers per
lib.clipk_set_option(b"wgrad_kernel", SCHED)                  # step), 4: software-pipelined (stamps at its 8 blocks)
dev = torch.device("cuda:0")
dy = (torch.randn(M, N, device=dev) * 0.1).to(torch.bfloat16)
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
a = Args()
ntn, ntk, splits, mps = C.c_int(), C.c_int(), C.c_int(), C.c_int()
lib.clipk_wgrad_v3_plan(M, N, K, C.byref(ntn), C.byref(ntk), C.byref(splits), C.byref(mps))
slab = torch.empty(splits.value * (N * K + ntk.value * N), dtype=torch.float32, device=dev)
a.dY, a.lddy, a.X, a.ldx = dy.data_ptr(), N, x.data_ptr(), K
a.slab, a.bslab = slab.data_ptr(), slab.data_ptr() + 4 * splits.value * N * K
a.M, a.N, a.K = M, N, K
a.ntn, a.ntk, a.splits, a.m_per_split = ntn.value, ntk.value, splits.value, mps.value
trace = torch.zeros(136, dtype=torch.int64, device=dev)
assert lib.clipk_wgrad_v3_set_trace(C.c_void_p(trace.data_ptr())) == 0
for _ in range(int(os.environ.get('TRACE_LAUNCHES', '3000'))):   # >= 2 s back to back: the clock the chip holds under this load
    assert lib.clipk_wgrad_v3_launch(C.byref(a), None) == 0
torch.cuda.synchronize()
t = trace.cpu().tolist()
print(f"schedule {SCHED}, N={N} K={K} M={M}: tiles {ntn.value}x{ntk.value}, {splits.value} splits of {mps.value} rows "
      f"({mps.value // 64} steps); ideal MFMA block = 16 x 16 = 256 cycles, 8 blocks per step = 2048")
cyc, ticks, nkt = t[128], t[129], t[130]
print(f"main loop of workgroup 0: {cyc} shader cycles in {ticks} ticks of 100 MHz = {cyc / ticks * 0.1:.3f} GHz in-kernel "
      f"clock; {cyc / max(nkt, 1):.0f} cycles per step = {2048.0 * nkt / cyc:.3f} of the MFMA pipe's cycles; at this clock "
      f"the dense bf16 peak is {256 * 4 * 1024 * cyc / ticks * 0.1 / 1e3:.0f} TFLOP/s")
for w, name in ((0, "wave 0 (group 0)"), (1, "wave 4 (group 1)")):
    st = t[64 * w:64 * w + 64]
    print(name)
    for s in range(4):
        row = st[16 * s:16 * s + 8]
        nxt = st[16 * (s + 1)] if s < 3 else None
        d = [row[i + 1] - row[i] for i in range(7)] + ([nxt - row[7]] if nxt else [])
        print(f"  step {8 + s}: " + " ".join(f"{v:5d}" for v in d) + f"   sum {sum(d)}")
