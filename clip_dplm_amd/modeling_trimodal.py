"""Tri-modal contrastive model (cell state x perturbation x protein) — SURVEY §8f rank 3.

Mirror of the classes of current/tf_clip_codes (1).ipynb:13026-13176 (cell 41: ProjectionHead, TransformerEncoder,
CellStateEncoder, PerturbationEncoder, ContrastiveModel) with the same names, constructor arguments, parameter names
(state_dict compatible) and output dict, on the libclipk kernels:
  * Linear layers = bf16-MFMA GEMMs; the perturbation encoder's `esm_projection(x) + value_encoder(v)` is ONE GEMM on
    the concatenated input [x | v] (K padded to a multiple of 32);
  * TransformerEncoder = the post-LN kernel stack (nn.TransformerEncoderLayer arithmetic, batch_first=False layout:
    attention runs over axis 0 of what it is given — SURVEY App. A-8 — so for [B, G, E] perturbation inputs the
    samples of the batch attend to each other per gene position, and 2-D [B, E] inputs are one unbatched sequence of
    B tokens, exactly as nn.TransformerEncoderLayer treats them);
  * the three pairwise symmetric losses on one logit_scale = loss.tri_modal_loss (one batched launch per pass).

Upstream defect (never executed in the notebook: the cell has no outputs): `cell_enc[:, 0]` / `protein_enc[:, 0]` index
a 2-D [B, E] encoder output and yield a [B] vector, on which the projection head cannot run.  The docstrings give the
intent ("embeddings for first token"): this module takes position 0 of 3-D encoder outputs and the rows themselves of
2-D ones (reference defects A-19).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as KF
from .loss import tri_modal_loss
from .modeling_clip import KLayerNorm, KLinear
from .modeling_seqclip import RNARBPCLIPEncoder, RNARBPCLIPProjectionHead


def _pad_k(x, w):
    """Zero-pad the contraction dimension of (x [.., K], w [N, K]) to a multiple of 32 (kernel tiles; exact)."""
    K = x.shape[-1]
    pad = (-K) % 32
    if pad:
        x, w = F.pad(x, (0, pad)), F.pad(w, (0, pad))
    return x, w


class ProjectionHead(RNARBPCLIPProjectionHead):
    """tf_clip_codes (1).ipynb:13031-13055: skip(x) + layer_scale * MLP3(x), hidden 2 * input_dim (identical to the
    RNA-RBP notebook's head)."""


class TransformerEncoder(RNARBPCLIPEncoder):
    """tf_clip_codes (1).ipynb:13056-13072: 3 x nn.TransformerEncoderLayer(d, nhead=8, ffn=4d, dropout=0.1) + LayerNorm,
    batch_first=False.  forward(x, mask): x [S, N, E] or unbatched [S, E]; mask = key padding mask [N, S] / [S],
    True = ignore."""

    def forward(self, x, mask=None):                       # noqa: D401  (signature of the notebook)
        unbatched = x.dim() == 2
        if unbatched:
            x = x.unsqueeze(1)                             # [S, 1, E]
            mask = None if mask is None else mask.unsqueeze(0)
        # kernels are batch-major [N, S, E]: attention over S for each n
        y = super().forward(x.transpose(0, 1).contiguous(), src_key_padding_mask=mask)
        y = y.transpose(0, 1)
        return y.squeeze(1) if unbatched else y


class CellStateEncoder(nn.Module):
    """tf_clip_codes (1).ipynb:13074-13089."""

    def __init__(self, gene_dim, hidden_dim, dropout: float = 0.1):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Linear(gene_dim + 1, hidden_dim),           # gene_expression + pseudotime
            KLayerNorm(hidden_dim),
            nn.GELU(),
            KLinear(hidden_dim, hidden_dim),
        )
        self._c0 = KF.WeightCache()
        self.graph_encoder = TransformerEncoder(hidden_dim, dropout=dropout)

    def forward(self, x, connectivity):
        e = self.encoder
        xp, wp = _pad_k(x, e[0].weight)                    # gene_dim + 1 is odd: pad K for the MFMA tiles
        h = KF.linear(xp, wp, e[0].bias, self._c0, version_fn=KF.params_version(e[0].weight))
        h = e[1](h, act="gelu")                            # LayerNorm + erf-GELU in one kernel
        h = e[3](h)
        graph_mask = (connectivity.sum(-1) == 0).bool()    # cells without neighbours are not attended to
        return self.graph_encoder(h, mask=graph_mask)


class PerturbationEncoder(nn.Module):
    """tf_clip_codes (1).ipynb:13091-13111."""

    def __init__(self, esm_dim=1280, hidden_dim=512, dropout: float = 0.1):
        super().__init__()
        self.value_encoder = nn.Linear(1, hidden_dim)
        self.esm_projection = nn.Linear(esm_dim, hidden_dim)
        self._c = KF.WeightCache()
        self.transformer = TransformerEncoder(hidden_dim, dropout=dropout)

    def forward(self, gene_esm_embeddings, values):
        # esm_projection(x) + value_encoder(v[..., None]) == [x | v] @ [W_esm | w_val]^T + (b_esm + b_val): one GEMM
        xin = torch.cat([gene_esm_embeddings, values.unsqueeze(-1).to(gene_esm_embeddings.dtype)], -1)
        w = torch.cat([self.esm_projection.weight, self.value_encoder.weight], 1)
        xin, w = _pad_k(xin, w)
        x = KF.linear(xin, w, self.esm_projection.bias + self.value_encoder.bias, self._c,
                      version_fn=KF.params_version(self.esm_projection.weight, self.value_encoder.weight))
        return self.transformer(x)


class ContrastiveModel(nn.Module):
    """tf_clip_codes (1).ipynb:13113-13176: same constructor, forward signature and output dict."""

    def __init__(self, gene_dim, protein_dim, projection_dim=512, esm_dim=1280, dropout: float = 0.1):
        super().__init__()
        self.cell_encoder = CellStateEncoder(gene_dim, projection_dim, dropout=dropout)
        self.pert_encoder = PerturbationEncoder(esm_dim, projection_dim, dropout=dropout)
        self.protein_encoder = TransformerEncoder(protein_dim, dropout=dropout)
        self.cell_projection = ProjectionHead(projection_dim, projection_dim)
        self.pert_projection = ProjectionHead(projection_dim, projection_dim)
        self.protein_projection = ProjectionHead(protein_dim, projection_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @staticmethod
    def _first(enc):
        return enc[:, 0] if enc.dim() == 3 else enc        # see the module docstring (upstream defect A-19)

    def forward(self, cell_state, connectivity, gene_esm_embeddings, gene_values, protein_emb, group=None):
        cell_enc = self.cell_encoder(cell_state, connectivity)
        pert_enc = self.pert_encoder(gene_esm_embeddings, gene_values)
        protein_enc = self.protein_encoder(protein_emb)
        cell_embed = KF.l2_normalize(self.cell_projection(self._first(cell_enc)))
        pert_embed = KF.l2_normalize(self.pert_projection(self._first(pert_enc)))
        protein_embed = KF.l2_normalize(self.protein_projection(self._first(protein_enc)))
        out = {"cell_embed": cell_embed, "pert_embed": pert_embed, "protein_embed": protein_embed}
        out.update(tri_modal_loss(cell_embed, pert_embed, protein_embed, self.logit_scale.exp(), group=group))
        return out
