#!/usr/bin/env python3
"""Single-stream vs dual-stream tower execution must give the same loss and gradients (GPU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clip_dplm_amd as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = K.ProteinRNACLIP().to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
B, L = 128, 256
g = torch.Generator().manual_seed(1)
ids = torch.randint(4, 24, (B, L), generator=g); ids[:, 0] = 0; ids[:, -1] = 2
rna = torch.randn(B, L, 768, generator=g)
ids, rna = ids.to(dev), rna.to(dev)
res = {}
for mode in (False, True, False, True):
    m.dual_stream = mode
    for p in m.parameters(): p.grad = None
    loss = m.loss(rna, ids); loss.backward(); torch.cuda.synchronize()
    flat = torch.cat([p.grad.flatten() for p in m.parameters()])
    res.setdefault(mode, []).append((loss.item(), flat.clone()))
    print("dual" if mode else "single", loss.item(), flat.norm().item())
a, b = res[False][0][1], res[True][0][1]
print("single vs single max diff", (res[False][0][1] - res[False][1][1]).abs().max().item())
print("single vs dual   max diff", (a - b).abs().max().item(), "rel", ((a - b).norm() / a.norm()).item())
