#!/usr/bin/env python3
"""Per-shape timing of the exact-f32 GEMM at the position-0-sliced notebook model's shapes (M = batch = 32 rows):
forward (NT), input gradient (trans_b) and weight gradient (trans_a + trans_b) of every Linear of the RBP tower
(1280 wide) - each launch against the time its algorithmic bytes take at 5 TB/s.  Usage: python tools/exp_gemm_f32_skinny.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    tot = ideal = 0.0
    for name, K, N in (("in_proj", 1280, 3840), ("out_proj", 1280, 1280), ("linear1", 1280, 5120), ("linear2", 5120, 1280),
                       ("head 0", 1280, 2560), ("head 4", 2560, 2560), ("head 8", 2560, 512), ("rna linear1", 120, 480)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        t_f = timeit(lambda: ops.gemm_f32(x, w, bias=b, addend=res))
        t_d = timeit(lambda: ops.gemm_f32(dy, w, trans_b=True))
        t_w = timeit(lambda: ops.gemm_f32(dy, x, trans_a=True, trans_b=True))
        by = N * K * 4
        us = by / 5e12 * 1e6
        print(f"{name:12s} K={K:5d} N={N:5d}: fwd {t_f:7.1f} us  dgrad {t_d:7.1f} us  wgrad {t_w:7.1f} us   "
              f"(weight {by / 1e6:5.1f} MB = {us:5.1f} us at 5 TB/s)", flush=True)
        tot += t_f + t_d + t_w
        ideal += 3 * us
    print(f"sum {tot:.0f} us, at 5 TB/s {ideal:.0f} us")


if __name__ == "__main__":
    main()
