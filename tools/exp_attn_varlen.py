#!/usr/bin/env python3
"""Whole-head attention kernels on packed batches: time per (sequence, head) against the sequence length (run on the GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench_kernels as bk  # noqa: E402
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, D = 1024, 20, 24
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(256, dtype=torch.float32)[:, None] * inv[None]
rope = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
for L in (256, 192, 129, 96, 64):
    T = B * L
    qkv = (torch.randn(T, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
    dout = (torch.randn(T, H * D, device=dev) * 0.5).to(torch.bfloat16)
    cu = (torch.arange(B + 1, dtype=torch.int32) * L).to(dev)
    o, lse = ops.attn_varlen_fwd_rot_(qkv.clone(), cu, 256, H, D, rope, q_scale=D ** -0.5)
    f, _ = bk.timeit(lambda: ops.attn_varlen_fwd_rot_(qkv, cu, 256, H, D, rope, q_scale=D ** -0.5))
    b, _ = bk.timeit(lambda: ops.attn_varlen_bwd(qkv, o, dout, lse, cu, 256, H, D, rope=rope, q_scale=D ** -0.5, prerotated=True))
    print(f"L={L:4d}: fwd {f * 1e3:7.1f} us  bwd {b * 1e3:7.1f} us   ({B * H} heads)", flush=True)

# the general kernels (RNA encoder heads: 8 x 96, no RoPE) on the same packed batches
H, D = 8, 96
for L in (256, 192, 129, 96, 64):
    T = B * L
    qkv = (torch.randn(T, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
    dout = (torch.randn(T, H * D, device=dev) * 0.5).to(torch.bfloat16)
    cu = (torch.arange(B + 1, dtype=torch.int32) * L).to(dev)
    o, lse = ops.attn_varlen_fwd(qkv, cu, 256, H, D, q_scale=D ** -0.5)
    f, _ = bk.timeit(lambda: ops.attn_varlen_fwd(qkv, cu, 256, H, D, q_scale=D ** -0.5))
    b, _ = bk.timeit(lambda: ops.attn_varlen_bwd(qkv, o, dout, lse, cu, 256, H, D, q_scale=D ** -0.5))
    print(f"hd 96 general, L={L:4d}: fwd {f * 1e3:7.1f} us  bwd {b * 1e3:7.1f} us", flush=True)
