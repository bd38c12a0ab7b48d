"""CPU restatement of the full BASELINE-config-2 dual encoder (ESM-2 protein + post-LN RNA transformer +
ProjectionHeads + symmetric InfoNCE), assembled from encoder_ref / clip_ref.  Test infrastructure only: used by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (kind "port")."""
from __future__ import annotations

from typing import Dict

import torch

from . import clip_ref, encoder_ref

SD = Dict[str, torch.Tensor]


def protein_rna_clip_loss(sd: SD, rna_values, protein_ids, rna_mask, protein_mask, *, esm_layers: int, esm_heads: int,
                          rna_layers: int, rna_heads: int, rna_act: str = "gelu", rna_eps: float = 1e-12,
                          pooling: str = "mean", symmetric: bool = True):
    """sd uses clip_dplm_amd.ProteinRNACLIP's state_dict keys (protein_model.* = EsmModel keys, rna_model.* =
    RNARBPCLIPEncoder keys, *_projection.* = old/clip.py ProjectionHead keys, logit_scale)."""
    esd = {k[len("protein_model."):]: v for k, v in sd.items() if k.startswith("protein_model.")}
    rsd = {"e." + k[len("rna_model."):]: v for k, v in sd.items() if k.startswith("rna_model.")}
    pm = protein_mask if protein_mask is not None else torch.ones_like(protein_ids)
    hp = encoder_ref.esm_encoder(protein_ids, pm, esd, esm_layers, esm_heads, 1e-5)
    rv = rna_mask.bool() if rna_mask is not None else None
    hr = encoder_ref.post_ln_encoder(rna_values, rsd, "e", rna_layers, rna_heads, rv, rna_act, rna_eps, rna_eps)
    er = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hr, rv, pooling), sd, "rna_projection"))
    ep = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hp, pm.bool(), pooling), sd, "protein_projection"))
    logits = (er @ ep.t()) * sd["logit_scale"].exp()
    loss = clip_ref.clip_loss_symmetric(logits) if symmetric else clip_ref.ce_diag(logits)
    return loss, er, ep
