#!/usr/bin/env python3
"""A reference point for the dominant kernel's fraction of the bf16 peak: the plain bf16 product  C[M, N] = A[M, K] B[N, K]^T
(bf16 out, no bias - the `linear->bf16` epilogue class) of clipk_gemm_nt against torch.matmul (= the ROCm GEMM library this
PyTorch build dispatches to) on the shapes of the metric step, M = B L = 262144 rows.  Interleaved rounds after a warm-up.
Nothing in the product calls the library: this is context for the roofline numbers, not a dependency."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 1024 * 256
g = torch.Generator().manual_seed(0)


def timeit(f, n=10):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"{'shape':34s} {'clipk us':>9s} {'TF/s':>6s} {'library us':>11s} {'TF/s':>6s}   clipk / library time")
for name, N, K in (("esm qkv            N=1440 K= 480", 1440, 480), ("esm out_proj       N= 480 K= 480", 480, 480),
                   ("esm fc1            N=1920 K= 480", 1920, 480), ("esm fc2 / d_fc1    N= 480 K=1920", 480, 1920),
                   ("esm d_qkv          N= 480 K=1440", 480, 1440), ("rna qkv            N=2304 K= 768", 2304, 768),
                   ("rna fc1            N=3072 K= 768", 3072, 768), ("rna fc2 / d_fc1    N= 768 K=3072", 768, 3072)):
    a = (torch.randn(M, K, generator=g) * 1.0).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bt = b.t()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f1 = lambda: ops.gemm_nt(a, b, out=out)
    f2 = lambda: torch.matmul(a, bt, out=out)
    for _ in range(30):
        f1(); f2()
    t1, t2, t3 = [], [], []
    for _ in range(5):
        t1.append(timeit(f1)); t2.append(timeit(f2))
        ops.set_option("epi_nt", 1)                      # non-temporal output stores (off by default: nothing in the step)
        t3.append(timeit(f1))
        ops.set_option("epi_nt", 0)
    m1, m2, m3 = statistics.median(t1), statistics.median(t2), statistics.median(t3)
    fl = 2.0 * M * N * K
    print(f"{name:34s} {m1:9.1f} {fl / m1 / 1e6:6.0f} {m2:11.1f} {fl / m2 / 1e6:6.0f}   {m1 / m2:5.2f}   "
          f"clipk with streaming stores {m3:7.1f} us {fl / m3 / 1e6:5.0f} TF/s ({m3 / m2:4.2f})", flush=True)
    del a, b, bt, out
