#!/usr/bin/env python3
"""A/B of the row stores of the whole-head attention kernels (option attn_row_stores): the ESM-2-35M shape (20 heads of 24,
L = 256) at the metric batch - forward with in-place rotation (clipk_attn_fwd_rot: rotated q / k written back) and the fused
backward (dq / dk / dv rows) - one thread per row against four lanes per row from LDS (bit 0 of the option: backward, bit 1: forward).  Interleaved rounds, HIP events.

    python3 tools/exp_attn_row_stores.py [B] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, H, D = 256, 20, 24
rnd = lambda s, sc=1.0: (torch.randn(s, device=dev) * sc).to(torch.bfloat16)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
r = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
qkv0 = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
qkv = qkv0.clone()
out, lse = ops.attn_fwd_rot_(qkv, B, L, H, D, r, q_scale=D ** -0.5)
bufs = [qkv0.clone() for _ in range(8)]                 # the forward rotates in place: fresh inputs, cloned outside the timing


def fwd(flag):
    ops.set_option("attn_row_stores", 2 * flag)
    for b_ in bufs:
        b_.copy_(qkv0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b_ in bufs:
        o = ops.attn_fwd_rot_(b_, B, L, H, D, r, q_scale=D ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / len(bufs), (bufs[0].clone(), o[0].clone(), o[1].clone())


def bwd(flag, n=16):                     # (the backward's own A/B branch is gone: both arms run the kept kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, g


for _ in range(4):
    fwd(1); bwd(1)
tf, tb = {0: [], 1: []}, {0: [], 1: []}
for _ in range(R):
    for f in (0, 1):
        tf[f].append(fwd(f)[0])
        tb[f].append(bwd(f)[0])
f0, f1 = fwd(0)[1], fwd(1)[1]
g0, g1 = bwd(0)[1], bwd(1)[1]
ops.reset_options()
same_f = all(torch.equal(a, b) for a, b in zip(f0, f1))
same_b = torch.equal(g0, g1)
med = lambda v: sorted(v)[len(v) // 2]
print(f"ESM-2-35M attention, B={B} L={L}: us per launch (median of {R} interleaved rounds); bit-identical: forward {same_f}, backward {same_b}")
print(f"  forward + in-place rotation   one thread per row {med(tf[0]):7.1f}   four lanes per row {med(tf[1]):7.1f}   {['%.0f' % x for x in tf[0]]} {['%.0f' % x for x in tf[1]]}")
print(f"  whole-head backward           one thread per row {med(tb[0]):7.1f}   four lanes per row {med(tb[1]):7.1f}   {['%.0f' % x for x in tb[0]]} {['%.0f' % x for x in tb[1]]}")
