"""Time the short-head attention backward on the bench's ESM-2-35M shape: whole-head kernel vs the dQ + dK/dV pair.

Method matters here: the first few hundred launches of a process run 10-20 % slower than steady state (clock
ramp), so configurations timed one after the other are not comparable.  Warm up first, then time the configurations
in interleaved rounds and look at the rounds side by side.
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, H, D = 512, 256, 20, 24
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * L, 3 * H * D, generator=g).to(torch.bfloat16).to(dev)
dout = torch.randn(B * L, H * D, generator=g).to(torch.bfloat16).to(dev)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
r = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
rot = ops.rope_qk_(qkv.clone(), B, L, H, D, r)
out, lse = ops.attn_fwd(rot, B, L, H, D, rope=None, q_scale=D ** -0.5)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def bwd():
    return ops.attn_bwd(rot, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=True)


for _ in range(200):
    bwd()
torch.cuda.synchronize()
res = collections.defaultdict(list)
for rnd in range(4):
    for mode, waves, name in (("1", "8", "whole-head, 8 waves"), ("1", "4", "whole-head, 4 waves"), ("0", "8", "dQ + dK/dV pair")):
        ops.set_option("attn_fused_bwd", int(mode))
        ops.set_option("attn_fused_waves", int(waves))
        res[name].append(timeit(bwd))
for k, v in res.items():
    print(f"{k:20s}", " ".join(f"{x:7.1f}" for x in v), "us")
ops.set_option("attn_fused_bwd", 1)
ops.set_option("attn_fused_waves", 4)
g1 = bwd().float()
ops.set_option("attn_fused_bwd", 0)
g0 = bwd().float()
print(f"max rel diff {(g1 - g0).abs().max().item() / g0.abs().max().item():.2e}")

# RNA encoder head shape (dQ + dK/dV pair: the whole-head kernel needs hd <= 32)
B2, L2, H2, D2 = 512, 256, 8, 96
q2 = torch.randn(B2 * L2, 3 * H2 * D2, generator=g).to(torch.bfloat16).to(dev)
do2 = torch.randn(B2 * L2, H2 * D2, generator=g).to(torch.bfloat16).to(dev)
o2, l2 = ops.attn_fwd(q2, B2, L2, H2, D2, rope=None, q_scale=D2 ** -0.5)
t = [timeit(lambda: ops.attn_bwd(q2, o2, do2, l2, B2, L2, H2, D2, rope=None, q_scale=D2 ** -0.5)) for _ in range(3)]
print("hd 96 backward    ", " ".join(f"{x:7.1f}" for x in t), "us")
t = [timeit(lambda: ops.attn_fwd(q2, B2, L2, H2, D2, rope=None, q_scale=D2 ** -0.5)) for _ in range(3)]
print("hd 96 forward     ", " ".join(f"{x:7.1f}" for x in t), "us")
fw = collections.defaultdict(list)
for rnd in range(3):
    for mode, name in (("1", "hd 24 forward, whole-head"), ("0", "hd 24 forward, general")):
        ops.set_option("attn_whole_fwd", int(mode))
        fw[name].append(timeit(lambda: ops.attn_fwd(rot, B, L, H, D, rope=None, q_scale=D ** -0.5)))
    ops.set_option("attn_whole_fwd", 1)
    scratch = qkv.clone()
    fw["hd 24 rope_qk_ + forward (two launches)"].append(timeit(
        lambda: ops.attn_fwd(ops.rope_qk_(scratch, B, L, H, D, r), B, L, H, D, rope=None, q_scale=D ** -0.5)))
    fw["hd 24 attn_fwd_rot_ (one kernel)"].append(timeit(
        lambda: ops.attn_fwd_rot_(scratch, B, L, H, D, r, q_scale=D ** -0.5)))
for k, v in fw.items():
    print(f"{k:40s}", " ".join(f"{x:7.1f}" for x in v), "us")
