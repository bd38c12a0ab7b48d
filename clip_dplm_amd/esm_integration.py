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

from dataclasses import dataclass
from enum import Enum
from typing import List, Optional, Union

import torch
import torch.nn as nn

from . import functional as KF
from .encoders import ESM2Encoder, ESM2_SHAPES
from .modeling_clip import KLayerNorm, KLinear

ESM2_ALPHABET = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q",
                 "N", "F", "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>"]
_TOK = {t: i for i, t in enumerate(ESM2_ALPHABET)}


class BiologicalDataType(Enum):          # triple_flow/1_config.py:57-68, same members and values
    GENE_EXPRESSION = "gene_expression"
    PROTEIN_SEQUENCE = "protein_sequence"
    PERTURBATION = "perturbation"
    PSEUDOTIME = "pseudotime"
    CONNECTIVITY = "connectivity"
    GENE_SEQUENCE = "gene_sequence"      # extension: a readable name for "anything that is not a protein sequence"
                                         # (get_embeddings :124-127 sends every other member to the gene projection)


@dataclass
class ESMConfig:
    """triple_flow/1_config.py:153-183, same fields and defaults.  `model_name` selects an ESM-2 SHAPE from
    clip_dplm_amd.ESM2_SHAPES (a superset of the reference's three whitelisted names: BASELINE's ESM-2-35M is not in
    that whitelist); nothing is fetched by name.  `num_attention_heads`, `dropout` and
    `use_sequence_context` are declared by the reference and read by nothing on the path (kept for construction
    compatibility)."""
    model_name: str = "esm2_t33_650M_UR50D"
    esm_dim: int = 1280
    protein_dim: int = 512
    gene_dim: int = 512
    num_attention_heads: int = 8
    dropout: float = 0.1
    use_sequence_context: bool = True
    max_sequence_length: int = 1024
    tokenizer_path: Optional[str] = None

    def validate_model(self):
        """Same error as the reference (:175-183) for a name without a known shape."""
        if self.model_name not in ESM2_SHAPES:
            raise ValueError(f"Invalid ESM model: {self.model_name}")


_SPECIALS = [t for t in ESM2_ALPHABET if len(t) > 1]


def _encode(seq: str) -> List[int]:
    """EsmTokenizer's splitting (checked against transformers.EsmTokenizer built from this alphabet,
    tools/make_golden.py gen_esm_integration): vocabulary tokens are matched greedily wherever they occur ('<mask>' in
    the text is the mask token), whitespace separates and is dropped, and every maximal run of other characters
    becomes ONE <unk>."""
    out, i, n, in_unk = [], 0, len(seq), False
    while i < n:
        c = seq[i]
        if c.isspace():
            in_unk = False
            i += 1
            continue
        if c == "<":
            hit = next((t for t in _SPECIALS if seq.startswith(t, i)), None)
            if hit is not None:
                out.append(_TOK[hit])
                i += len(hit)
                in_unk = False
                continue
        if c in _TOK:
            out.append(_TOK[c])
            in_unk = False
        elif not in_unk:
            out.append(_TOK["<unk>"])
            in_unk = True
        i += 1
    return out


def tokenize(sequences: List[str], max_length: int = 1024):
    """tokenizer(sequences, padding=True, truncation=True, max_length=...) of 3_esm_integration.py:104-110: <cls> +
    tokens + <eos>, the TOKENS cut to max_length - 2, rows padded with <pad> to the longest.  Returns (ids, mask)."""
    keep = max(int(max_length) - 2, 0)
    rows = [[0] + _encode(s)[:keep] + [2] for s in sequences]
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


@dataclass
class ESMOutput:                         # 3_esm_integration.py:39-43
    embeddings: torch.Tensor
    attention_weights: Optional[torch.Tensor] = None


class ESMIntegration(nn.Module):
    """3_esm_integration.py:45-135: frozen ESM-2 (:83-84) + the two per-token projections, same constructor
    (`ESMIntegration(config: ESMConfig)`), attributes (`config`, `model`, `protein_projection`, `gene_projection`,
    `cache`) and `get_embeddings` behaviour — including its cache, which is keyed on the sequences only (:100-102: a
    second call with the same sequences and another data_type returns the first call's result).

    Differences forced by the environment: the backbone is built from the explicit shape `config.model_name` names and
    initialised randomly (`AutoModel.from_pretrained(<name>)` of :77 is a network fetch); `esm_state_dict` (an
    `EsmModel.state_dict()`, e.g. loaded by the caller from a local checkpoint) fills it.  The tokenizer is the built-in
    33-token ESM-2 alphabet (`tokenize`).  `ESMIntegration("esm2_t12_35M_UR50D", protein_dim=...)` — the keyword form of
    earlier rounds — is still accepted and builds the ESMConfig itself."""

    def __init__(self, config: Union[ESMConfig, str, None] = None, esm_state_dict=None, **kwargs):
        super().__init__()
        if config is None or isinstance(config, str):
            name = config if config is not None else kwargs.pop("model_name", ESMConfig.model_name)
            kwargs.setdefault("esm_dim", ESM2_SHAPES[name][1] if name in ESM2_SHAPES else ESMConfig.esm_dim)
            config = ESMConfig(model_name=name, **kwargs)
        elif kwargs:
            raise TypeError(f"unexpected keyword arguments next to an ESMConfig: {sorted(kwargs)}")
        config.validate_model()
        self.config = config
        hidden = ESM2_SHAPES[config.model_name][1]
        if hidden != config.esm_dim:
            raise ValueError(f"config.esm_dim = {config.esm_dim} but {config.model_name} has hidden size {hidden}")
        self._setup_esm(esm_state_dict)
        self.protein_projection = ProteinProjection(esm_dim=config.esm_dim, output_dim=config.protein_dim)
        self.gene_projection = GeneProjection(esm_dim=config.esm_dim, output_dim=config.gene_dim)
        self.cache = {}

    def _setup_esm(self, esm_state_dict=None) -> None:
        self.model = ESM2Encoder.from_name(self.config.model_name)
        if esm_state_dict is not None:
            self.model.load_state_dict(esm_state_dict, strict=False)
        for p in self.model.parameters():
            p.requires_grad = False

    @property
    def max_sequence_length(self) -> int:
        return self.config.max_sequence_length

    @torch.no_grad()
    def get_embeddings(self, sequences: List[str], data_type: BiologicalDataType) -> ESMOutput:
        key = str(hash(tuple(sequences)))
        if key in self.cache:
            return self.cache[key]
        ids, mask = tokenize(sequences, self.config.max_sequence_length)
        dev = next(self.model.parameters()).device
        h = self.model(ids.to(dev), attention_mask=mask.to(dev))
        proj = self.protein_projection if data_type == BiologicalDataType.PROTEIN_SEQUENCE else self.gene_projection
        out = ESMOutput(embeddings=proj(h), attention_weights=None)       # outputs.attentions is None at :129 too
        self.cache[key] = out
        return out


def create_esm_integration(model_name: str = "esm2_t33_650M_UR50D", esm_state_dict=None, **kwargs) -> ESMIntegration:
    """3_esm_integration.py:215-228: ESMConfig(model_name=..., **kwargs) -> ESMIntegration."""
    return ESMIntegration(ESMConfig(model_name=model_name, **kwargs), esm_state_dict=esm_state_dict)


def get_embeddings_batch(sequences: List[str], esm_model: ESMIntegration, data_type: BiologicalDataType,
                         batch_size: int = 32) -> torch.Tensor:
    """3_esm_integration.py:231-245: get_embeddings over slices of `batch_size` sequences, concatenated along dim 0.
    As in the reference every slice is padded to ITS longest sequence, so the slices must tokenise to one length for the
    concatenation to be defined (torch.cat raises otherwise, there and here)."""
    embeddings = []
    for i in range(0, len(sequences), batch_size):
        embeddings.append(esm_model.get_embeddings(sequences[i:i + batch_size], data_type).embeddings)
    return torch.cat(embeddings, dim=0)
