#!/usr/bin/env python3
"""A/B of the MFMA shape in the 256 x 256 Linear kernel's main loop: v_mfma_f32_16x16x32_bf16 (the
product) against v_mfma_f32_32x32x16_bf16 (half the issue slots and operand-register reads per FLOP) in the SAME
8-phase loop - same LDS-DMA, same 12 fragment reads and 256 matrix-pipe cycles per quadrant, same registers.

    make tools/probes/libgemm_trace16.so tools/probes/libgemm_trace32.so && python3 tools/exp_gemm_mfma_shape.py

Both builds run WITHOUT the epilogue (option gemm_abl = 1): the main loop alone, on random bf16 operands (the 32x32x16
build multiplies fragments fetched in the 16x16x32 lane layout: its products are garbage, its instruction mix is not).
Per shape and arm, after >= 2 s of back-to-back launches (the clock the chip holds under THIS load): wall time per
launch (HIP events, interleaved rounds), shader cycles per K-tile of workgroup 0 (s_memtime around its main loops) and
the in-kernel clock (s_memtime / s_memrealtime), as tools/exp_wgrad_trace.py does for the weight-gradient kernel.
"pipe busy" = 2048 / (cycles per K-tile): a 256 x 256 x 64 K-tile is 8.39 MFLOP = 2048 cycles of a CU's four matrix pipes
(1024 FLOP per cycle and SIMD for either MFMA shape)."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_dplm_amd._ffi import BF16, GemmArgs  # noqa: E402

dev = torch.device("cuda:0")
libs = {}
for name in ("16", "32"):
    lib = C.CDLL(os.path.join(ROOT, "tools", "probes", f"libgemm_trace{name}.so"))
    assert lib.clipk_set_option(b"gemm_abl", 1) == 0
    libs[name] = lib
SHAPES = [("esm fc2  K=1920", 262144, 480, 1920), ("rna qkv  K=768", 262144, 2304, 768), ("esm qkv  K=480", 262144, 1440, 480),
          ("rna fc2  K=2048", 262144, 768, 2048)]
SECONDS = float(os.environ.get("EXP_SECONDS", "2.0"))
print(f"{'shape':18s} {'arm':>10s} {'us/launch':>10s} {'TFLOP/s':>8s} {'cyc/K-tile':>11s} {'pipe busy':>10s} {'GHz in-kernel':>14s}")
for sname, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    args = GemmArgs()
    args.A, args.lda, args.B, args.ldb = a.data_ptr(), K, b.data_ptr(), K
    args.C, args.ldc, args.c_dtype = c.data_ptr(), N, BF16
    args.M, args.N, args.K = M, N, K
    args.alpha = 1.0
    res = {}
    for arm, lib in libs.items():
        trace = torch.zeros(8, dtype=torch.int64, device=dev)
        assert lib.clipk_gemm_v3_set_trace(C.c_void_p(trace.data_ptr())) == 0
        launch = lambda: lib.clipk_gemm_nt_v3_launch(C.byref(args), None)
        for _ in range(50):
            assert launch() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < SECONDS:                # warm: the clock this loop sustains
            for _ in range(100):
                launch()
            torch.cuda.synchronize()
            n += 100
        trace.zero_()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200):
            launch()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 200
        cyc, ticks, nkt = trace.cpu().tolist()[:3]
        res[arm] = us
        print(f"{sname:18s} {('16x16x32' if arm == '16' else '32x32x16'):>10s} {us:10.1f} {2.0 * M * N * K / us / 1e6:8.0f} "
              f"{cyc / max(nkt, 1):11.0f} {2048.0 * nkt / max(cyc, 1):10.3f} {cyc / max(ticks, 1) * 0.1:14.3f}", flush=True)
        lib.clipk_gemm_v3_set_trace(None)
    print(f"{'':18s} 32x32x16 / 16x16x32 wall = {res['32'] / res['16']:.3f}")
    del a, b, c
