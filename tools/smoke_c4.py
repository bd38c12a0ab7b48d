#!/usr/bin/env python3
"""BASELINE config 4 shape check on one GPU: ESM-2-650M (33 x 1280, hd 64, RoPE) at L = 1024, small batch:
forward + fused loss + backward + fused AdamW run, loss finite and ~ ln B, timing printed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clip_dplm_amd as K
dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", 16)), 1024
torch.manual_seed(0)
m = K.ProteinRNACLIP(esm="esm2_t33_650M_UR50D").to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
opt = K.FusedAdamW(m, lr=1e-4)
g = torch.Generator().manual_seed(1)
ids = torch.randint(4, 24, (B, L), generator=g); ids[:, 0] = 0; ids[:, -1] = 2
rna = torch.randn(B, L, 768, generator=g)
ids, rna = ids.to(dev), rna.to(dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(); loss = m.loss(rna, ids); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: loss {loss.item():.5f} (ln B = {torch.log(torch.tensor(float(B))).item():.4f}) {1e3*(time.perf_counter()-t0):.1f} ms, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
assert torch.isfinite(loss)
print("C4 shape OK:", sum(p.numel() for p in m.parameters()) / 1e6, "M params")
