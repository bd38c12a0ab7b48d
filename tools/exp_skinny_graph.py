#!/usr/bin/env python3
"""Cost of one skinny exact-f32 Linear INSIDE a captured step: 40 calls of the same product captured into one hipGraph,
replayed, time per call (kernel + cross-workgroup reduce kernel + inter-node gaps - what the training step pays), for every
split count S of the contraction and the library's own choice (S = 0).  Shapes: the rbp tower of the sliced notebook model
(M = 32 rows), forward and input gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, CALLS = 32, 40


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(CALLS):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / 20 / CALLS


for name, K, N in (("in_proj", 1280, 3840), ("out_proj", 1280, 1280), ("linear1", 1280, 5120), ("linear2", 5120, 1280),
                   ("head 1", 1280, 2560), ("head 2", 2560, 2560), ("head 3", 2560, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    dy = torch.randn(M, N, device=dev)
    yo, xo = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    rows = {"fwd": [], "dgrad": []}
    for S in (0, 1, 2, 3, 4, 6, 8):
        ops.set_option("gemm_f32_splits", S)
        rows["fwd"].append(graph_time(lambda: ops.gemm_f32(x, w, out=yo)))
        rows["dgrad"].append(graph_time(lambda: ops.gemm_f32(dy, w, trans_b=True, out=xo)))
    ops.set_option("gemm_f32_splits", 0)
    for k, v in rows.items():
        print(f"{name:9s} {k:5s} K={K if k == 'fwd' else N:5d} N={N if k == 'fwd' else K:5d}  us per call in a graph:  auto {v[0]:5.1f} | "
              + "  ".join(f"S={s}: {t:5.1f}" for s, t in zip((1, 2, 3, 4, 6, 8), v[1:])), flush=True)
