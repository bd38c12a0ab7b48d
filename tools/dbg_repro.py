#!/usr/bin/env python3
"""Is clipk_attn_bwd bit-reproducible?  Four launches on the same inputs per `prerotated` mode (1 = rotate-half columns, 2 =
pair-interleaved columns) at the ESM-2-35M shape; prints where two runs differ (row within the sequence, q / k / v section,
head, column).  This is the script that located the nondeterministic packed-f32 code of DESIGN.md 3.4."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, H, D = 64, 256, 20, 24
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(B * L, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dout = (torch.randn(B * L, H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
o, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
for pr in (1, 2):
    outs = [ops.attn_bwd(qkv, o, dout, lse, B, L, H, D, rope=(cos, sin), q_scale=D ** -0.5, prerotated=pr) for _ in range(4)]
    torch.cuda.synchronize()
    d = torch.stack([(outs[0].float() - outs[k].float()).abs() for k in (1, 2, 3)]).amax(0)
    idx = (d > 0).nonzero()
    print(f"prerotated {pr}: {idx.shape[0]} of {d.numel()} elements differ between runs, max {d.max().item():.3g}")
    if idx.shape[0]:
        rows, cols = idx[:, 0] % L, idx[:, 1]
        print("  rows (by 16):", collections.Counter((rows // 16 * 16).tolist()).most_common(8))
        print("  section:", dict(collections.Counter((cols // (H * D)).tolist())), " column in head:",
              sorted(collections.Counter((cols % D).tolist()).items()))
