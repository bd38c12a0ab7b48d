#!/usr/bin/env python3
"""Weight-gradient kernel: time against M (fixed cost vs main-loop slope) and with / without the bias gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench_kernels as bk  # noqa: E402
from clip_dplm_amd import ops  # noqa: E402

ops.set_option("wgrad_kernel", int(os.environ.get("WGRAD_KERNEL", "3")))
for name, N, K in (("esm out", 480, 480), ("esm fc1", 1920, 480), ("rna fc1", 2048, 768)):
    for M in (32768, 65536, 131072, 262144):
        dy, x = bk.rnd((M, N), scale=0.1), bk.rnd((M, K))
        res = []
        for wb in (True, False):
            med, _ = bk.timeit(lambda: ops.gemm_wgrad(dy, x, want_bias=wb), iters=10, rounds=7)
            res.append(med)
        print(f"{name:8s} M={M:7d}  with bias {res[0] * 1e3:8.1f} us {2.0 * M * N * K / res[0] / 1e9:6.0f} TF/s | "
              f"without {res[1] * 1e3:8.1f} us {2.0 * M * N * K / res[1] / 1e9:6.0f} TF/s", flush=True)
