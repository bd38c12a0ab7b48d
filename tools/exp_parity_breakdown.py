#!/usr/bin/env python3
"""Where does |loss_gpu - loss_oracle| come from?  Stage-by-stage comparison of the HIP path with the CPU oracle on the
smoke() model (two layers of each tower at the metric model's layer shapes): pooled encoder outputs, embeddings, loss;
and the loss recomputed with one tower / the heads swapped for the oracle's values (run on the GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import clip_dplm_amd as K  # noqa: E402
from clip_dplm_amd.encoders import ESM2_SHAPES  # noqa: E402
from clip_dplm_amd.modeling_seqclip import pool  # noqa: E402
from oracle import clip_ref, encoder_ref  # noqa: E402

dev = torch.device("cuda:0")
ESM2_SHAPES["smoke"] = (2, 480, 20, 1920)
B, L = int(os.environ.get("B", "128")), 64
for seed in (0, 1, 2):
    torch.manual_seed(seed)
    m = K.ProteinRNACLIP(esm="smoke", rna_dim=768, rna_layers=2, rna_heads=8, rna_ffn=2048, projection_dim=512).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234 + seed)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    rna = torch.randn(B, L, 768, generator=g)
    m = m.to(dev)
    with torch.no_grad():
        hr_g = pool(m.rna_model(rna.to(dev)), None, m.pooling).cpu()
        hp_g = pool(m.protein_model(ids.to(dev)), None, m.pooling).cpu()
        er_g = K.functional.l2_normalize(m.rna_projection(hr_g.to(dev))).cpu()
        ep_g = K.functional.l2_normalize(m.protein_projection(hp_g.to(dev))).cpu()
    esd = {k[len("protein_model."):]: v for k, v in sd.items() if k.startswith("protein_model.")}
    rsd = {"e." + k[len("rna_model."):]: v for k, v in sd.items() if k.startswith("rna_model.")}
    hr_o = encoder_ref.pool(encoder_ref.post_ln_encoder(rna, rsd, "e", 2, 8, None, "gelu", 1e-12, 1e-12), None, "mean")
    hp_o = encoder_ref.pool(encoder_ref.esm_encoder(ids, torch.ones_like(ids), esd, 2, 20, 1e-5), torch.ones_like(ids).bool(), "mean")
    head = lambda h, name: clip_ref.l2_normalize(clip_ref.projection_head(h, sd, name))
    er_o, ep_o = head(hr_o, "rna_projection"), head(hp_o, "protein_projection")
    sc = sd["logit_scale"].exp()
    L_ = lambda a, b: clip_ref.clip_loss_symmetric((a @ b.t()) * sc).item()
    ref = L_(er_o, ep_o)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"seed {seed}: loss_oracle {ref:.6f}")
    print(f"  pooled rel err  rna {rel(hr_g, hr_o):.2e}  protein {rel(hp_g, hp_o):.2e}")
    print(f"  embed  rel err  rna {rel(er_g, er_o):.2e}  protein {rel(ep_g, ep_o):.2e}")
    print(f"  dloss: gpu towers + gpu heads {L_(er_g, ep_g) - ref:+.2e} | gpu towers + ORACLE heads "
          f"{L_(head(hr_g, 'rna_projection'), head(hp_g, 'protein_projection')) - ref:+.2e} | only rna tower on gpu "
          f"{L_(er_g, ep_o) - ref:+.2e} | only protein tower on gpu {L_(er_o, ep_g) - ref:+.2e} | oracle towers + GPU heads "
          f"{L_(K.functional.l2_normalize(m.rna_projection(hr_o.to(dev))).detach().cpu(), K.functional.l2_normalize(m.protein_projection(hp_o.to(dev))).detach().cpu()) - ref:+.2e}")
    la = m.loss(rna.to(dev), ids.to(dev)).item()
    with torch.no_grad():
        lb = m.loss(rna.to(dev), ids.to(dev)).item()
        er2, ep2 = m.embed(rna.to(dev), ids.to(dev))
    er3, ep3 = m.embed(rna.to(dev), ids.to(dev))
    print(f"  m.loss with grad {la - ref:+.2e} | no_grad {lb - ref:+.2e} | embed(no_grad) vs staged: "
          f"{(er2.cpu() - er_g).abs().max().item():.1e} {(ep2.cpu() - ep_g).abs().max().item():.1e} | embed(grad) vs embed(no_grad): "
          f"{(er3.detach() - er2).abs().max().item():.1e} {(ep3.detach() - ep2).abs().max().item():.1e}")
