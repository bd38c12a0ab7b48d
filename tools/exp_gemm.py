#!/usr/bin/env python3
"""Experiment: where does gemm_nt time go? vary K and the epilogue at M=131072."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops
from tools.bench_kernels import timeit, rnd, DEV, T
print(f"{'N':>5s} {'K':>5s} {'epi':12s} {'us':>8s} {'TF/s':>7s} {'GB moved':>9s} {'TB/s':>6s}")
for N in (1920, 480):
    for K in (32, 480, 1920):
        for epi in ("bf16", "bias", "gelu+pre", "res32"):
            a, b = rnd((T, K)), rnd((N, K), scale=0.05)
            kw = {}
            byts = T * K * 2 + T * N * 2
            if epi == "bias": kw = {"bias": torch.randn(N, device=DEV)}
            if epi == "gelu+pre": kw = {"bias": torch.randn(N, device=DEV), "act": "gelu", "out_preact": True}; byts += T * N * 2
            if epi == "res32": kw = {"residual": torch.randn(T, N, device=DEV), "out_dtype": torch.float32}; byts += T * N * 6
            med, _ = timeit(lambda: ops.gemm_nt(a, b, **kw), iters=3, rounds=3)
            print(f"{N:5d} {K:5d} {epi:12s} {med*1e3:8.1f} {2.0*T*N*K/med/1e9:7.0f} {byts/1e9:9.2f} {byts/med/1e9:6.2f}")
            del a, b, kw
