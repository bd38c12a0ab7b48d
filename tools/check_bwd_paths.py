"""One training step of the bench model with the whole-head attention backward on and off: the flat gradients must
agree to bf16-rounding level (the two paths differ only in f32 summation order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clip_dplm_amd as K  # noqa: E402
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = K.ProteinRNACLIP(esm="esm2_t12_35M_UR50D").to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
g = torch.Generator().manual_seed(1)
ids = torch.randint(4, 24, (256, 256), generator=g).to(dev)
rna = torch.randn(256, 256, 768, generator=g).to(dev)
grads = {}
for mode in ("1", "0"):
    ops.set_option("attn_fused_bwd", int(mode))
    for p in m.parameters():
        p.grad = None
    loss = m.loss(rna, ids)
    loss.backward()
    grads[mode] = torch.cat([p.grad.detach().float().flatten() for p in m.parameters() if p.grad is not None])
    print("mode", mode, "loss", loss.item(), "|g|", grads[mode].norm().item())
a, b = grads["1"], grads["0"]
rel = (a - b).norm().item() / b.norm().item()
cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
print(f"relative L2 difference {rel:.3e}, cosine {cos:.8f}")
assert rel < 2e-2 and cos > 0.9998
