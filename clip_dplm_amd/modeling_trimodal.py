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


class _CatPadColsFn(torch.autograd.Function):
    """[w0 | w1 | ... | 0-pad] along the input dimension, rebuilt only when a source Parameter changed (the
    concatenation used to be rebuilt by torch.cat on every forward; only its bf16 copy was version-cached).  `holder`
    keeps the f32 operand between calls; every call returns a fresh view of it, so the autograd edges of one forward
    never outlive that forward.  Backward: the column blocks of the gradient."""

    @staticmethod
    def forward(ctx, holder, pad_to, *ws):
        key = KF.params_version(*ws)() + (KF.weight_epoch(),)
        if holder.get("key") != key or KF.capture_force():
            w = torch.cat([t.detach() for t in ws], 1)
            pad = (-w.shape[1]) % pad_to
            holder["w"] = F.pad(w, (0, pad)) if pad else w
            holder["key"] = key
        ctx.cols = [t.shape[1] for t in ws]
        return holder["w"].view_as(holder["w"])

    @staticmethod
    def backward(ctx, dw):
        outs, c0 = [], 0
        for c in ctx.cols:
            outs.append(dw[:, c0:c0 + c])
            c0 += c
        return (None, None, *outs)


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

    def __init__(self, gene_dim, hidden_dim, dropout: float = 0.1, precision: str = "f32"):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Linear(gene_dim + 1, hidden_dim),           # gene_expression + pseudotime
            KLayerNorm(hidden_dim),
            nn.GELU(),
            KLinear(hidden_dim, hidden_dim),
        )
        self._c0 = KF.WeightCache()
        self._w0 = {}
        self.precision = precision
        self.graph_encoder = TransformerEncoder(hidden_dim, dropout=dropout, precision=precision)

    takes_precision = True

    def forward(self, x, connectivity):
        e = self.encoder
        if self.precision == "f32":                        # exact-f32 GEMM: any K, no padding
            h = KF.linear_f32(x, e[0].weight, e[0].bias)
        else:
            wp = _CatPadColsFn.apply(self._w0, 32, e[0].weight)            # gene_dim + 1 is odd: pad K for the MFMA tiles
            xp = F.pad(x, (0, wp.shape[1] - x.shape[-1]))
            h = KF.linear(xp, wp, e[0].bias, self._c0, version_fn=KF.params_version(e[0].weight))
        h = e[1](h, act="gelu")                            # LayerNorm + erf-GELU in one kernel
        h = e[3](h)
        graph_mask = (connectivity.sum(-1) == 0).bool()    # cells without neighbours are not attended to
        return self.graph_encoder(h, mask=graph_mask)


class PerturbationEncoder(nn.Module):
    """tf_clip_codes (1).ipynb:13091-13111."""

    takes_precision = True

    def __init__(self, esm_dim=1280, hidden_dim=512, dropout: float = 0.1, precision: str = "f32"):
        super().__init__()
        self.value_encoder = nn.Linear(1, hidden_dim)
        self.esm_projection = nn.Linear(esm_dim, hidden_dim)
        self._c = KF.WeightCache()
        self._w = {}
        self.precision = precision
        self.transformer = TransformerEncoder(hidden_dim, dropout=dropout, precision=precision)

    def forward(self, gene_esm_embeddings, values):
        lead = gene_esm_embeddings.shape[:-1]
        if self.precision == "f32":
            # esm_projection(x) + value_encoder(v[..., None]): two exact-f32 products, the second (K = 1) adding the first
            # in its epilogue
            x2 = gene_esm_embeddings.reshape(-1, gene_esm_embeddings.shape[-1])
            p = KF.linear_f32(x2, self.esm_projection.weight, self.esm_projection.bias)
            x = KF.linear_f32(values.reshape(-1, 1).to(x2.dtype), self.value_encoder.weight, self.value_encoder.bias,
                              addend=p).reshape(*lead, -1)
            return self.transformer(x)
        # bf16 kernels: [x | v] @ [W_esm | w_val]^T + (b_esm + b_val) is ONE GEMM (K padded to the MFMA tile)
        w = _CatPadColsFn.apply(self._w, 32, self.esm_projection.weight, self.value_encoder.weight)
        xin = torch.cat([gene_esm_embeddings, values.unsqueeze(-1).to(gene_esm_embeddings.dtype)], -1)
        xin = F.pad(xin, (0, w.shape[1] - xin.shape[-1]))
        x = KF.linear(xin, w, self.esm_projection.bias + self.value_encoder.bias, self._c,
                      version_fn=KF.params_version(self.esm_projection.weight, self.value_encoder.weight))
        return self.transformer(x)


class ContrastiveModel(nn.Module):
    """tf_clip_codes (1).ipynb:13113-13176: same constructor, forward signature and output dict.

    As in RNARBPCLIPModel, every encoder here attends over axis 0 (the batch) per position and only position 0 of a 3-D
    encoder output is read (`pert_enc[:, 0]`, ipynb:13140-13142): 3-D inputs are sliced to position 0 BEFORE their encoder
    (exact; `slice_first_position=False` computes every position as the notebook does), and the model runs in exact f32
    by default (`precision="f32"`; "bf16" = the bf16-MFMA kernels)."""

    def __init__(self, gene_dim, protein_dim, projection_dim=512, esm_dim=1280, dropout: float = 0.1,
                 precision: str = "f32", slice_first_position: bool = True):
        super().__init__()
        self.cell_encoder = CellStateEncoder(gene_dim, projection_dim, dropout=dropout, precision=precision)
        self.pert_encoder = PerturbationEncoder(esm_dim, projection_dim, dropout=dropout, precision=precision)
        self.protein_encoder = TransformerEncoder(protein_dim, dropout=dropout, precision=precision)
        self.cell_projection = ProjectionHead(projection_dim, projection_dim)
        self.pert_projection = ProjectionHead(projection_dim, projection_dim)
        self.protein_projection = ProjectionHead(protein_dim, projection_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.slice_first_position = bool(slice_first_position)
        self.multi_stream = False
        self._streams = None
        KF.set_linear_precision(self, precision)

    @staticmethod
    def _first(enc):
        if enc.dim() == 3:                                 # see the module docstring (upstream defect A-19)
            return enc.reshape(enc.shape[0], enc.shape[2]) if enc.shape[1] == 1 else enc[:, 0]
        return enc

    def _sl(self, x):
        return x[:, :1] if (self.slice_first_position and x.dim() == 3) else x

    def forward(self, cell_state, connectivity, gene_esm_embeddings, gene_values, protein_emb, group=None):
        if self.slice_first_position and gene_esm_embeddings.dim() == 3:
            gene_esm_embeddings, gene_values = gene_esm_embeddings[:, :1], gene_values[:, :1]
        cell = lambda: KF.l2_normalize(self.cell_projection(self._first(self.cell_encoder(cell_state, connectivity))))
        pert = lambda: KF.l2_normalize(self.pert_projection(self._first(self.pert_encoder(gene_esm_embeddings, gene_values))))
        prot = lambda: KF.l2_normalize(self.protein_projection(self._first(self.protein_encoder(self._sl(protein_emb)))))
        if self.multi_stream and cell_state.is_cuda:       # opt-in: the three towers on three HIP streams (RNARBPCLIPModel)
            if self._streams is None:
                self._streams = KF.branch_streams(3)
            cell_embed, pert_embed, protein_embed = KF.parallel_branches(
                self._streams, (cell, pert, prot),
                ((cell_state, connectivity), (gene_esm_embeddings, gene_values), (protein_emb,)))
        else:
            cell_embed, pert_embed, protein_embed = cell(), pert(), prot()
        out = {"cell_embed": cell_embed, "pert_embed": pert_embed, "protein_embed": protein_embed}
        out.update(tri_modal_loss(cell_embed, pert_embed, protein_embed, self.logit_scale.exp(), group=group))
        return out
