#!/usr/bin/env python3
"""Run ONE kernel shape a few times (for rocprofv3 --pmc passes):  python tools/prof_one.py gemm_fc1|gemm_qkv|wgrad_fc1|attn_esm"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops
dev = torch.device("cuda:0")
T = 131072
which = sys.argv[1] if len(sys.argv) > 1 else "gemm_fc1"
rnd = lambda s, sc=1.0: (torch.randn(s, device=dev) * sc).to(torch.bfloat16)
if which == "gemm_fc1":
    a, b, bias = rnd((T, 480)), rnd((1920, 480), 0.05), torch.randn(1920, device=dev)
    f = lambda: ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True)
elif which == "gemm_qkv":
    a, b, bias = rnd((T, 480)), rnd((1440, 480), 0.05), torch.randn(1440, device=dev)
    f = lambda: ops.gemm_nt(a, b, bias=bias)
elif which == "gemm_fc2":
    a, b, bias = rnd((T, 1920)), rnd((480, 1920), 0.05), torch.randn(480, device=dev)
    res = torch.randn(T, 480, device=dev)
    f = lambda: ops.gemm_nt(a, b, bias=bias, residual=res, out_dtype=torch.float32)
elif which == "gemm_rnaqkv":
    a, b, bias = rnd((T, 768)), rnd((2304, 768), 0.05), torch.randn(2304, device=dev)
    f = lambda: ops.gemm_nt(a, b, bias=bias)
elif which == "wgrad_fc1":
    dy, x = rnd((T, 1920), 0.1), rnd((T, 480))
    f = lambda: ops.gemm_wgrad(dy, x, want_bias=True)
elif which == "attn_esm":
    B, L, H, D = 512, 256, 20, 24
    qkv = rnd((B * L, 3 * H * D))
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    r = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
    f = lambda: ops.attn_fwd(qkv, B, L, H, D, rope=r, q_scale=D ** -0.5)
if which in ("attn_esm_bwd", "attn_rna_bwd"):
    B, L, H, D = (512, 256, 20, 24) if which == "attn_esm_bwd" else (512, 256, 8, 96)
    qkv = rnd((B * L, 3 * H * D))
    r = None
    if which == "attn_esm_bwd":
        inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
        r = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
    if r is not None:
        qkv = ops.rope_qk_(qkv, B, L, H, D, r)                # as the encoder does: rotate once, attention sees rotated q / k
    out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
    dout = rnd((B * L, H * D))
    f = lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=r is not None)
for _ in range(5):
    f()
torch.cuda.synchronize()
