"""CPU restatement (PyTorch f32) of triple_flow/3_esm_integration.py's path: tokenise -> frozen ESM-2 -> per-token
projection (`ESMIntegration.get_embeddings` :90-135, `ProteinProjection` :137-159, `GeneProjection` :161-182,
`ResidualBlock` :184-197, `AttentionBlock` :199-213).  Test infrastructure only (tests/, tools/make_golden.py).

Pinned by tests/golden/esm_integration.npz, which tools/make_golden.py writes by running the REFERENCE's own
`ESMIntegration.get_embeddings` (its two `from_pretrained(<name>)` calls pointed at a locally constructed, seeded
`transformers.EsmModel` and a `transformers.EsmTokenizer` built from the 33-token alphabet: no fetch)."""
from __future__ import annotations

import re
from typing import Dict, List

import torch

from . import clip_ref, encoder_ref

SD = Dict[str, torch.Tensor]

# the ESM-2 vocabulary in id order (facebook/esm2_* vocab.txt; also transformers' EsmTokenizer default for ESM-2)
VOCAB = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q", "N", "F",
         "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>"]
_ID = {t: i for i, t in enumerate(VOCAB)}
# EsmTokenizer: every vocabulary entry is a no-split token (matched wherever it occurs); what lies between matches is
# split on whitespace and each piece that is not in the vocabulary becomes <unk>
_SPLIT = re.compile("(" + "|".join(re.escape(t) for t in sorted(VOCAB, key=len, reverse=True)) + ")")


def tokenize(sequences: List[str], max_length: int):
    """`self.tokenizer(sequences, padding=True, truncation=True, max_length=..., return_tensors="pt")` (:104-110)."""
    rows = []
    for s in sequences:
        toks = []
        for piece in _SPLIT.split(s):
            if piece in _ID:
                toks.append(_ID[piece])
            else:
                toks.extend(_ID["<unk>"] for _ in piece.split())
        rows.append([_ID["<cls>"]] + toks[: max(max_length - 2, 0)] + [_ID["<eos>"]])
    L = max(len(r) for r in rows)
    ids = torch.full((len(rows), L), _ID["<pad>"], dtype=torch.long)
    mask = torch.zeros((len(rows), L), dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
        mask[i, : len(r)] = 1
    return ids, mask


def _residual_block(x, sd: SD, p: str):
    """:184-197  x + Linear(Dropout(ReLU(LayerNorm(Linear(x)))))  (eval: dropout = identity)."""
    h = torch.relu(clip_ref._ln(clip_ref._linear(x, sd, p + ".layers.0"), sd, p + ".layers.1", 1e-5))
    return x + clip_ref._linear(h, sd, p + ".layers.4")


def _attention_block(x, sd: SD, p: str, heads: int = 8):
    """:199-213  LayerNorm(x + MultiheadAttention(x, x, x)), 8 heads, batch_first, no mask."""
    E = x.shape[-1]
    qkv = x @ sd[p + ".attention.in_proj_weight"].t() + sd[p + ".attention.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)
    a = encoder_ref._mha(q, k, v, heads, None, float(E // heads) ** -0.5)
    a = clip_ref._linear(a, sd, p + ".attention.out_proj")
    return clip_ref._ln(x + a, sd, p + ".norm", 1e-5)


def projection(x, sd: SD, prefix: str, kind: str):
    """ProteinProjection (:137-159, kind 'protein': ResidualBlock in the middle) / GeneProjection (:161-182, kind 'gene':
    AttentionBlock).  x: [B, L, esm_dim]."""
    p = prefix + ".projection"
    h = torch.relu(clip_ref._ln(clip_ref._linear(x, sd, p + ".0"), sd, p + ".1", 1e-5))
    h = _residual_block(h, sd, p + ".4") if kind == "protein" else _attention_block(h, sd, p + ".4")
    return clip_ref._ln(clip_ref._linear(h, sd, p + ".5"), sd, p + ".6", 1e-5)


def get_embeddings(sequences: List[str], sd: SD, *, esm_layers: int, esm_heads: int, max_sequence_length: int,
                   protein: bool):
    """ESMIntegration.get_embeddings (:90-135) without its cache.  sd: `model.*` = EsmModel keys,
    `protein_projection.*` / `gene_projection.*`."""
    ids, mask = tokenize(sequences, max_sequence_length)
    esd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    h = encoder_ref.esm_encoder(ids, mask, esd, esm_layers, esm_heads, 1e-5)
    if protein:
        return projection(h, sd, "protein_projection", "protein"), ids, mask
    return projection(h, sd, "gene_projection", "gene"), ids, mask
