#!/usr/bin/env python3
"""Run every attention kernel of the metric configuration (B = 1024, L = 256: ESM-2-35M heads 20 x 24 with RoPE,
RNA heads 8 x 96) a few times, for rocprofv3 --pmc / --stats passes (tools/pmc_attn.sh):

    python3 tools/prof_attn.py [B] [launches]

The kernels are exactly the ones the training step launches (round 4): whole-head forward on q / k the qkv projection's
epilogue already rotated (pair-interleaved heads) + whole-head fused backward with the interleaved RoPE^T for the ESM
shape; general forward + eight-wave whole-head backward for the RNA shape.  PROF_ATTN_ROT=1: the round-3 pair
(`clipk_attn_fwd_rot`, rotate-half backward) instead."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = 256
rnd = lambda s, sc=1.0: (torch.randn(s, device=dev) * sc).to(torch.bfloat16)


def rope_tables(D):
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    return fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)


# ESM shape
H, D = 20, 24
r = rope_tables(D)
qkv0 = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
qkv = qkv0.clone()
ROT = os.environ.get("PROF_ATTN_ROT") == "1"
if ROT:
    out, lse = ops.attn_fwd_rot_(qkv, B, L, H, D, r, q_scale=D ** -0.5)
else:
    out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
for _ in range(N):
    q2 = qkv0.clone()
    if ROT:
        ops.attn_fwd_rot_(q2, B, L, H, D, r, q_scale=D ** -0.5)
    else:
        ops.attn_fwd(q2, B, L, H, D, rope=None, q_scale=D ** -0.5)
for _ in range(N):
    ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=True if ROT else 2)
torch.cuda.synchronize()
del qkv0, qkv, out, lse, dout, q2

# RNA shape
H, D = 8, 96
qkv = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
for _ in range(N):
    ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
for _ in range(N):
    ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=None, q_scale=D ** -0.5)
torch.cuda.synchronize()
print("prof_attn done", B, N)
