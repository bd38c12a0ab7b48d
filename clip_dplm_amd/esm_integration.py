"""ESM integration: frozen ESM-2 encoder + per-token projections to the transport space.

Mirror of triple_flow/3_esm_integration.py: ESMIntegration :45-135, ProteinProjection :137-159, GeneProjection
:161-182, ResidualBlock :184-197, AttentionBlock :199-213 — same class names, constructor arguments and state_dict
keys — on the libclipk kernels (bf16-MFMA Linear, fused LayerNorm+ReLU, flash attention).

The reference builds its ESM model with `AutoModel.from_pretrained(<name>)` — a network fetch, unavailable here and
never attempted.  ESMIntegration therefore takes the ESM-2 SHAPE by name (clip_dplm_amd.ESM2_SHAPES), initialises
randomly, and loads a caller-provided EsmModel state_dict if one is given; a built-in tokenizer for the 33-token
ESM-2 alphabet replaces AutoTokenizer.
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional

import torch
import torch.nn as nn

from . import functional as KF
from .encoders import ESM2Encoder, ESM2_SHAPES
from .modeling_clip import KLayerNorm, KLinear

ESM2_ALPHABET = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q",
                 "N", "F", "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>"]
_TOK = {t: i for i, t in enumerate(ESM2_ALPHABET)}


class BiologicalDataType(Enum):          # triple_flow/1_config.py (only the two members the path reads)
    PROTEIN_SEQUENCE = "protein"
    GENE_SEQUENCE = "gene"


def tokenize(sequences: List[str], max_length: int = 1024):
    """<cls> + residues + <eos>, padded with <pad>; truncation like tokenizer(..., truncation=True, max_length)."""
    rows = [[0] + [_TOK.get(c, 3) for c in s[: max_length - 2]] + [2] for s in sequences]
    L = max(len(r) for r in rows)
    ids = torch.full((len(rows), L), 1, dtype=torch.long)
    mask = torch.zeros((len(rows), L), dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = torch.tensor(r)
        mask[i, : len(r)] = 1
    return ids, mask


class ResidualBlock(nn.Module):
    """3_esm_integration.py:184-197: x + Linear(Dropout(ReLU(LN(Linear(x)))))."""

    def __init__(self, dim: int):
        super().__init__()
        self.layers = nn.Sequential(KLinear(dim, dim), KLayerNorm(dim), nn.ReLU(), nn.Dropout(0.1), KLinear(dim, dim))

    def forward(self, x):
        l = self.layers
        h = l[3](l[1](l[0](x), act="relu"))
        return x + l[4](h)


class _MHA(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names."""

    def __init__(self, dim):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = KLinear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)
        self._cache = KF.WeightCache()


class AttentionBlock(nn.Module):
    """3_esm_integration.py:199-213: LN(x + MHA(x, x, x)), 8 heads, batch_first."""

    def __init__(self, dim: int, num_heads: int = 8):
        super().__init__()
        self.attention = _MHA(dim)
        self.norm = KLayerNorm(dim)
        self.num_heads = num_heads

    def forward(self, x):
        B, L, E = x.shape
        H = self.num_heads
        D = E // H
        a = self.attention
        qkv = KF.linear(x, a.in_proj_weight, a.in_proj_bias, a._cache, out_dtype=torch.bfloat16)
        ctx = KF.AttnFn.apply(qkv.reshape(B * L, 3 * E), B, L, H, D, None, D ** -0.5)
        attended = a.out_proj(ctx.reshape(B, L, E))
        return self.norm(x + attended)


class ProteinProjection(nn.Module):
    """3_esm_integration.py:137-159."""

    def __init__(self, esm_dim: int, output_dim: int):
        super().__init__()
        self.projection = nn.Sequential(KLinear(esm_dim, output_dim * 2), KLayerNorm(output_dim * 2), nn.ReLU(),
                                        nn.Dropout(0.1), ResidualBlock(output_dim * 2),
                                        KLinear(output_dim * 2, output_dim), KLayerNorm(output_dim))

    def forward(self, x):
        p = self.projection
        h = p[3](p[1](p[0](x), act="relu"))
        return p[6](p[5](p[4](h)))


class GeneProjection(nn.Module):
    """3_esm_integration.py:161-182."""

    def __init__(self, esm_dim: int, output_dim: int):
        super().__init__()
        self.projection = nn.Sequential(KLinear(esm_dim, output_dim * 2), KLayerNorm(output_dim * 2), nn.ReLU(),
                                        nn.Dropout(0.1), AttentionBlock(output_dim * 2),
                                        KLinear(output_dim * 2, output_dim), KLayerNorm(output_dim))

    def forward(self, x):
        p = self.projection
        h = p[3](p[1](p[0](x), act="relu"))
        return p[6](p[5](p[4](h)))


class ESMOutput:
    def __init__(self, embeddings, attention_weights=None):
        self.embeddings, self.attention_weights = embeddings, attention_weights


class ESMIntegration(nn.Module):
    """3_esm_integration.py:45-135 with an explicit-shape ESM-2 backbone (frozen, like :83-84)."""

    def __init__(self, model_name: str = "esm2_t33_650M_UR50D", protein_dim: int = 512, gene_dim: int = 512,
                 max_sequence_length: int = 1024, esm_state_dict=None):
        super().__init__()
        self.model = ESM2Encoder.from_name(model_name)
        if esm_state_dict is not None:
            self.model.load_state_dict(esm_state_dict, strict=False)
        for p in self.model.parameters():
            p.requires_grad = False
        esm_dim = ESM2_SHAPES[model_name][1]
        self.protein_projection = ProteinProjection(esm_dim, protein_dim)
        self.gene_projection = GeneProjection(esm_dim, gene_dim)
        self.max_sequence_length = max_sequence_length
        self.cache = {}

    @torch.no_grad()
    def get_embeddings(self, sequences: List[str], data_type: BiologicalDataType) -> ESMOutput:
        key = str(hash(tuple(sequences)))
        if key in self.cache:
            return self.cache[key]
        ids, mask = tokenize(sequences, self.max_sequence_length)
        dev = next(self.model.parameters()).device
        h = self.model(ids.to(dev), attention_mask=mask.to(dev))
        proj = self.protein_projection if data_type == BiologicalDataType.PROTEIN_SEQUENCE else self.gene_projection
        out = ESMOutput(proj(h), None)
        self.cache[key] = out
        return out
