#!/usr/bin/env python3
"""Where do the fused path's and the oracle's TRAINING TRAJECTORIES separate (bench.py parity.trajectory)?

Full-depth metric model at the parity sub-batch (B = 32, L = 256): k optimiser updates on the GPU (fused path); at every
step the CPU oracle is evaluated AT THE GPU PATH'S OWN WEIGHTS (forward + backward), so gradient quality is compared
at identical weights, class by class: relative error, cosine, share of sign flips - and the Adam update direction the
two gradients would produce.  Usage (GPU box): python tools/exp_trajectory.py [updates]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import clip_dplm_amd as K                    # noqa: E402
from clip_dplm_amd.encoders import ESM2_SHAPES   # noqa: E402
from oracle import model_ref                 # noqa: E402
import bench                                 # noqa: E402


def klass(n):
    if n == "logit_scale":
        return "logit_scale"
    if "word_embeddings" in n:
        return "embedding"
    tower = "esm" if n.startswith("protein_model") else ("rna" if n.startswith("rna_model") else "heads")
    if "LayerNorm" in n or "norm" in n or "layernorm" in n or ".1." in n and tower == "heads" or ".5." in n and tower == "heads":
        return tower + ".ln"
    return tower + (".bias" if n.endswith("bias") else ".weight")


def main():
    updates = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    esm = os.environ.get("EXP_ESM", "esm2_t12_35M_UR50D")
    nl, d, h, f = ESM2_SHAPES[esm]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm=esm).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt = K.FusedAdamW(m, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
    rna, ids = bench.synth_batch(32, 256, 768, dev, 4321)
    rna_c, ids_c = rna.cpu(), ids.cpu()
    torch.set_num_threads(bench.host_threads())
    okw = dict(esm_layers=nl, esm_heads=h, rna_layers=6, rna_heads=8)
    for it in range(updates + 1):
        opt.zero_grad()
        loss = m.loss(rna, ids)
        loss.backward()
        gg = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters()}
        sd = {k: (v.detach().float().cpu().clone().requires_grad_(True) if v.is_floating_point() else v.detach().cpu())
              for k, v in m.state_dict().items()}
        t0 = time.perf_counter()
        lo, _, _ = model_ref.protein_rna_clip_loss(sd, rna_c, ids_c, None, None, **okw)
        lo.backward()
        print(f"step {it}: loss gpu {loss.item():.6f}  oracle at the same weights {lo.item():.6f}  |d| {abs(loss.item() - lo.item()):.1e}"
              f"   ({time.perf_counter() - t0:.1f} s)", flush=True)
        agg = {}
        for n, g in gg.items():
            r = sd[n].grad
            if r is None:
                continue
            a = agg.setdefault(klass(n), [0.0, 0.0, 0.0, 0, 0, 0.0, 0.0])
            a[0] += float(((g - r) ** 2).sum()); a[1] += float((r ** 2).sum()); a[2] += float((g * r).sum())
            a[6] += float((g ** 2).sum())
            fl = torch.sign(g) != torch.sign(r)
            a[3] += int(fl.sum()); a[4] += fl.numel(); a[5] += float(r.abs()[fl].sum())
        tot = sum(a[1] for a in agg.values())
        for k_, a in sorted(agg.items()):
            cos = a[2] / max((a[1] * a[6]) ** 0.5, 1e-30)
            print(f"    {k_:14s} |g|^2 share {a[1] / tot:8.2e}  rel err {((a[0] / max(a[1], 1e-30)) ** 0.5):8.2e}  cos {cos:.6f}  "
                  f"sign flips {a[3] / max(a[4], 1):7.3%} of {a[4]}")
        if it < updates:
            opt.step()


if __name__ == "__main__":
    main()
