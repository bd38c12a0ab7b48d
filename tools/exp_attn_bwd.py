"""Time the short-head attention backward: whole-head kernel vs the dQ + dK/dV pair (ESM-2-35M shape of the bench)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, H, D = 512, 256, 20, 24
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * L, 3 * H * D, generator=g) * 1.0).to(torch.bfloat16).to(dev)
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


res = {}
for mode in ("0", "1"):
    os.environ["CLIPK_ATTN_FUSED_BWD"] = mode
    f = lambda: ops.attn_bwd(rot, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=True)
    res[mode] = (timeit(f), f().float())
print(f"two kernels: {res['0'][0]:.1f} us   whole-head: {res['1'][0]:.1f} us")
d = (res["0"][1] - res["1"][1]).abs().max().item() / res["0"][1].abs().max().item()
print(f"max rel diff {d:.2e}")

os.environ["CLIPK_ATTN_FUSED_BWD"] = "1"
for st in ("0", "3", "5", "7", "10"):
    os.environ["CLIPK_ATTN_STAGGER"] = st
    f = lambda: ops.attn_bwd(rot, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5, prerotated=True)
    print("start-up stagger", st, f"{timeit(f):.1f} us")
