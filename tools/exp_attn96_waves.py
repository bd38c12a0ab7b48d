#!/usr/bin/env python3
"""A/B of the hd-96 whole-head attention backward with four waves of 64 keys (attn_bwd_fused96_kernel) against eight waves of
32 keys (attn_bwd_fused96w8_kernel, option attn_fused_waves = 8) and the dQ + dK/dV pair: the RNA encoder's shape (8 heads
of 96, L = 256) at the metric batch, interleaved rounds on a warm GPU, HIP events.

    python3 tools/exp_attn96_waves.py [B] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, H, D = 256, 8, 96
rnd = lambda s, sc=1.0: (torch.randn(s, device=dev) * sc).to(torch.bfloat16)
qkv = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
ARMS = {"pair (dQ + dK/dV kernels)": {"attn_fused_bwd": 0}, "4 waves x 64 keys": {"attn_fused_bwd": 1, "attn_fused_waves": 4},
        "8 waves x 32 keys": {"attn_fused_bwd": 1, "attn_fused_waves": 8}}


def run(opts, n):
    ops.reset_options()
    for k, v in opts.items():
        ops.set_option(k, v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=None, q_scale=D ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, g


for _ in range(3):
    run(ARMS["4 waves x 64 keys"], 20)
ts = {a: [] for a in ARMS}
for _ in range(R):
    for a, o in ARMS.items():
        ts[a].append(run(o, 10)[0])
g = {a: run(o, 1)[1].float() for a, o in ARMS.items()}
ops.reset_options()
ref = g["pair (dQ + dK/dV kernels)"]
med = lambda v: sorted(v)[len(v) // 2]
print(f"attention backward, 8 heads of 96, B={B} L={L}: us per launch (median of {R} interleaved rounds)")
for a in ARMS:
    err = (g[a] - ref).abs().max().item() / ref.abs().max().item()
    print(f"  {a:28s} {med(ts[a]):8.1f}   max |diff| vs the pair / max |grad| = {err:.1e}   {['%.0f' % x for x in ts[a]]}")
