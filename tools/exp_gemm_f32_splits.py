#!/usr/bin/env python3
"""Kernel durations (HIP events around back-to-back launches, launch gaps amortised by queueing 200 launches) of the skinny
exact-f32 Linear for every cross-workgroup split count, at the sliced notebook model's big shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


def main():
    dev = torch.device("cuda:0")
    M = 32
    for name, K, N in (("linear2 fwd", 5120, 1280), ("linear1 fwd", 1280, 5120), ("in_proj fwd", 1280, 3840), ("head4 fwd", 2560, 2560)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.02
        dy = torch.randn(M, N, device=dev)
        row = []
        for S in (1, 2, 3, 4, 6, 8):
            ops.set_option("gemm_f32_splits", S)
            tf = timeit(lambda: ops.gemm_f32(x, w))
            td = timeit(lambda: ops.gemm_f32(dy, w, trans_b=True))
            row.append(f"S={S}: {tf:5.1f}/{td:5.1f}")
        ops.set_option("gemm_f32_splits", 0)
        ta = timeit(lambda: ops.gemm_f32(x, w))
        print(f"{name:12s} K={K} N={N} (fwd/dgrad us, host-bound floor ~10): " + "  ".join(row) + f"  auto fwd {ta:5.1f}", flush=True)


if __name__ == "__main__":
    main()
