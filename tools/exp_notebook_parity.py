#!/usr/bin/env python3
"""Loss error of the notebook model at its own dims vs the CPU oracle over seeds / batch sizes / head precision
(exploration for tests/test_gpu_configs.py::test_notebook_model_at_its_own_dims_vs_oracle)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import clip_dplm_amd as K
from oracle import model_ref
dev = torch.device("cuda:0")
torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
for B, Lp, seed in ((32, 600, 7), (32, 64, 8), (32, 64, 9), (64, 64, 10), (128, 64, 11)):
    torch.manual_seed(seed)
    m = K.RNARBPCLIPModel(rna_dim=120, rbp_dim=1280, projection_dim=512).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    Lr = 48
    rna = torch.randn(B, Lr, 120, generator=g); rbp = torch.randn(B, Lp, 1280, generator=g)
    lr = torch.randint(10, Lr + 1, (B,), generator=g); lp = torch.randint(Lp // 3, Lp + 1, (B,), generator=g)
    lr[0], lp[0] = Lr, Lp
    for i in range(B):
        rna[i, lr[i]:] = float("nan"); rbp[i, lp[i]:] = float("nan")
    t0 = time.time()
    with torch.no_grad():
        _, _, ref = model_ref.rnarbp_clip_forward(sd, rna, rbp)
    tc = time.time() - t0
    m = m.to(dev)
    out = {}
    for prec in ("bf16", "f32"):
        K.set_linear_precision(m, prec)
        with torch.no_grad():
            out[prec] = m(rna.to(dev), rbp.to(dev))[2].item()
    print(f"B={B} Lp={Lp} seed={seed}: oracle {ref.item():.6f} ({tc:.1f} s)  bf16 heads err {abs(out['bf16']-ref.item()):.2e}  "
          f"f32 heads err {abs(out['f32']-ref.item()):.2e}", flush=True)
    del m
    torch.cuda.empty_cache()
