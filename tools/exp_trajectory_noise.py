#!/usr/bin/env python3
"""CPU-only: how fast do two AdamW trajectories separate when one of them sees gradients with iid relative noise?
The oracle trains the smoke-size model (two layers of each tower at the metric model's widths, B = 32, L = 64) for three
updates (clip 1.0, AdamW lr 1e-4, wd 0.01) with exact f32 gradients and with gradients perturbed per tensor by
noise * ||g|| / sqrt(n) * N(0, 1).  Yardstick for bench.py's parity.trajectory: the bf16 path's gradients carry ~1.5 %
relative error."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import clip_dplm_amd as K
from clip_dplm_amd.encoders import ESM2_SHAPES
from oracle import model_ref
ESM2_SHAPES["smoke"] = (2, 480, 20, 1920)
torch.manual_seed(0)
m = K.ProteinRNACLIP(esm="smoke", rna_dim=768, rna_layers=2, rna_heads=8, rna_ffn=2048, projection_dim=512)
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(1234)
B, L = 32, 64
ids = torch.randint(4, 24, (B, L), generator=g); ids[:, 0], ids[:, -1] = 0, 2
rna = torch.randn(B, L, 768, generator=g)
okw = dict(esm_layers=2, esm_heads=20, rna_layers=2, rna_heads=8)
def run(noise, seed=0):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    gn = torch.Generator().manual_seed(seed)
    out = []
    for it in range(4):
        opt.zero_grad()
        l = model_ref.protein_rna_clip_loss(sd, rna, ids, None, None, **okw)[0]
        out.append(l.item()); l.backward()
        if noise:
            for p in params:
                if p.grad is not None:
                    p.grad += noise * p.grad.norm() / (p.grad.numel() ** 0.5) * torch.randn(p.grad.shape, generator=gn)
        torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
    return out
base = run(0.0)
for nz, sd_ in ((0.015, 1), (0.015, 2), (0.005, 1), (1e-6, 1)):
    o = run(nz, sd_)
    print(f"gradient noise {nz:g} (seed {sd_}): |loss - exact| per step:", " ".join(f"{abs(a-b):.1e}" for a, b in zip(o, base)))
