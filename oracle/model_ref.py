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


def contrastive_model_forward(sd: SD, cell_state, connectivity, gene_esm_embeddings, gene_values, protein_emb):
    """ContrastiveModel.forward of current/tf_clip_codes (1).ipynb:13113-13176 with the state_dict keys of that class.
    Encoders: CellStateEncoder :13074-13089 (MLP, then the post-LN stack over the B cells as ONE unbatched sequence with
    isolated cells masked as keys), PerturbationEncoder :13091-13111 (esm_projection + value_encoder, then the stack with
    attention over the batch axis per gene position, App. A-8), TransformerEncoder on the 2-D protein embeddings.
    `[:, 0]` is applied to 3-D encoder outputs only: on the 2-D ones it is the upstream defect A-19 (never executed)."""
    h = cell_state @ sd["cell_encoder.encoder.0.weight"].t() + sd["cell_encoder.encoder.0.bias"]
    h = clip_ref._gelu(clip_ref._ln(h, sd, "cell_encoder.encoder.1", 1e-5))
    h = clip_ref._linear(h, sd, "cell_encoder.encoder.3")
    keys_ok = ~(connectivity.sum(-1) == 0)                                            # [B]
    cell_enc = encoder_ref.post_ln_encoder(h[None], sd, "cell_encoder.graph_encoder", 3, 8, keys_ok[None], "relu",
                                           1e-5, 1e-5)[0]
    x = clip_ref._linear(gene_esm_embeddings, sd, "pert_encoder.esm_projection") + \
        clip_ref._linear(gene_values.unsqueeze(-1), sd, "pert_encoder.value_encoder")
    pert_enc = encoder_ref.post_ln_encoder(x.transpose(0, 1), sd, "pert_encoder.transformer", 3, 8, None, "relu",
                                           1e-5, 1e-5).transpose(0, 1)[:, 0]
    prot_enc = encoder_ref.post_ln_encoder(protein_emb[None], sd, "protein_encoder", 3, 8, None, "relu", 1e-5, 1e-5)[0]
    ce = clip_ref.l2_normalize(clip_ref.optimized_projection_head(cell_enc, sd, "cell_projection"))
    pe = clip_ref.l2_normalize(clip_ref.optimized_projection_head(pert_enc, sd, "pert_projection"))
    pr = clip_ref.l2_normalize(clip_ref.optimized_projection_head(prot_enc, sd, "protein_projection"))
    total, cp, cpr, ppr = clip_ref.tri_modal_losses(ce, pe, pr, sd["logit_scale"])
    return {"cell_embed": ce, "pert_embed": pe, "protein_embed": pr, "loss": total, "cell_pert_loss": cp,
            "cell_protein_loss": cpr, "pert_protein_loss": ppr}


def rnarbp_clip_forward(sd: SD, rna_emb, rbp_emb, num_layers: int = 3):
    """RNARBPCLIPModel.forward of current/rna_clip_codes.ipynb:1925-1954 (state_dict keys of that class): NaN-padded
    `[B, L, D]` inputs -> `create_padding_mask` (:1726-1736) -> 3 x nn.TransformerEncoderLayer(d, 8 heads, 4d, relu) +
    LayerNorm fed (B, L, D) although batch_first=False, mask transposed to match (:1936-1946): attention runs over the
    BATCH axis per position -> position 0 -> RNARBPCLIPProjectionHead -> normalise -> symmetric CE
    with exp(logit_scale).  Returns (rna_embed, rbp_embed, loss)."""
    def enc(x, prefix):
        valid = ~torch.isnan(x).any(-1)                      # [B, L]
        xt = torch.nan_to_num(x, 0.0).transpose(0, 1)        # [L, B, D]: "batch" = position, "sequence" = sample
        y = encoder_ref.post_ln_encoder(xt, sd, prefix, num_layers, 8, valid.transpose(0, 1), "relu", 1e-5, 1e-5)
        return y.transpose(0, 1)[:, 0]
    a = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(rna_emb, "rna_encoder"), sd, "rna_projection"))
    b = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(rbp_emb, "rbp_encoder"), sd, "rbp_projection"))
    loss = clip_ref.clip_loss_symmetric((a @ b.t()) * sd["logit_scale"].exp())
    return a, b, loss
