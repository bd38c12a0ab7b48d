"""CPU restatement (PyTorch f32) of the two sequence encoders on the hot path.  Test infrastructure only.

  * post-LN transformer layer == torch.nn.TransformerEncoderLayer(norm_first=False) as instantiated at
    current/rna_clip_codes.ipynb:1915 (and the "transformer" architecture of
    run1/configuration_hybrid_clip.py:153-157,68-79: 6 x 768, 8 heads, ffn 2048, gelu, eps 1e-12);
  * ESM-2 encoder == transformers.EsmModel (third party; the reference calls it at
    triple_flow/3_esm_integration.py:77-80,118-119): modeling_esm.py:48-52 rotate_half, :74-79 RoPE,
    :82-86 erf-GELU, :252-268 token-dropout rescale + mask multiply, :362-374 q pre-scale, :306-314 softmax
    attention, :429-438 / :517-521 pre-LN residual blocks, :552-553 final LN.

Weights come in dicts keyed exactly like the state_dicts of those modules.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .clip_ref import _gelu, _linear, _ln

SD = Dict[str, torch.Tensor]


def _mha(q, k, v, nheads: int, key_valid: Optional[torch.Tensor], scale: float, attn_mult=None):
    """q,k,v: [B, L, E]; key_valid: [B, L] bool (True = attend).  softmax(q k^T * scale + mask) v.
    attn_mult: optional [B, H, L, L] dropout multiplier (keep / (1 - p)) on the attention probabilities."""
    B, L, E = q.shape
    hd = E // nheads
    q = q.view(B, L, nheads, hd).transpose(1, 2)
    k = k.view(B, L, nheads, hd).transpose(1, 2)
    v = v.view(B, L, nheads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * scale
    if key_valid is not None:
        s = s.masked_fill(~key_valid[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if attn_mult is not None:
        p = p * attn_mult
    return (p @ v).transpose(1, 2).reshape(B, L, E)


def post_ln_layer(x, sd: SD, prefix: str, nheads: int, key_valid, activation: str = "relu", eps: float = 1e-5,
                  drop=None):
    """One nn.TransformerEncoderLayer (batch-first layout [B, L, E] = per-sequence attention).
    x = LN1(x + drop1(MHA(x))); x = LN2(x + drop2(W2 drop(act(W1 x)))).  drop = None: dropout is the identity (eval /
    p = 0); else a dict of keep / (1 - p) multipliers {"attn": [B,H,L,L], "d1": [B,L,E], "ffn": [B,L,F], "d2": [B,L,E]}
    standing for nn.TransformerEncoderLayer's four nn.Dropout sites (torch/nn/modules/transformer.py _sa_block,
    _ff_block, MultiheadAttention dropout_p)."""
    E = x.shape[-1]
    qkv = x @ sd[f"{prefix}.self_attn.in_proj_weight"].t() + sd[f"{prefix}.self_attn.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)
    ctx = _mha(q, k, v, nheads, key_valid, (E // nheads) ** -0.5, None if drop is None else drop["attn"])
    sa = _linear(ctx, sd, f"{prefix}.self_attn.out_proj")
    if drop is not None:
        sa = sa * drop["d1"]
    x = _ln(x + sa, sd, f"{prefix}.norm1", eps)
    h = _linear(x, sd, f"{prefix}.linear1")
    h = torch.relu(h) if activation == "relu" else _gelu(h)
    if drop is not None:
        h = h * drop["ffn"]
    ff = _linear(h, sd, f"{prefix}.linear2")
    if drop is not None:
        ff = ff * drop["d2"]
    return _ln(x + ff, sd, f"{prefix}.norm2", eps)


def post_ln_encoder(x, sd: SD, prefix: str, num_layers: int, nheads: int, key_valid, activation="relu", eps=1e-5,
                    final_eps=1e-5, drops=None):
    """RNARBPCLIPEncoder.forward (rna_clip_codes.ipynb:1918-1923): layers then a final LayerNorm.
    drops: optional list of per-layer dropout multiplier dicts (post_ln_layer)."""
    for i in range(num_layers):
        x = post_ln_layer(x, sd, f"{prefix}.layers.{i}", nheads, key_valid, activation, eps,
                          None if drops is None else drops[i])
    return _ln(x, sd, f"{prefix}.layernorm", final_eps)


# ------------------------------------------------------------------------------------------------- ESM-2
def rope_tables(L: int, hd: int, theta: float = 10000.0):
    """modeling_esm.py:141,150-160: inv_freq = theta^(-2i/hd); emb = cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos(), fr.sin()          # [L, hd/2]


def _rope(t, cos, sin):
    """t: [B, H, L, hd]; rotate-half (modeling_esm.py:48-52,74-79)."""
    half = t.shape[-1] // 2
    cosf = torch.cat([cos, cos], -1)[None, None]
    sinf = torch.cat([sin, sin], -1)[None, None]
    rot = torch.cat([-t[..., half:], t[..., :half]], -1)
    return t * cosf + rot * sinf


def esm_embeddings(ids, attention_mask, sd: SD, mask_token_id: int = 32, token_dropout: bool = True):
    """modeling_esm.py:225-270 (rotary position type, no emb layer norm before)."""
    x = sd["embeddings.word_embeddings.weight"][ids]
    if token_dropout:
        x = x.masked_fill((ids == mask_token_id).unsqueeze(-1), 0.0)
        src_len = attention_mask.sum(-1) if attention_mask is not None else torch.full((ids.shape[0],), ids.shape[1])
        ratio = (ids == mask_token_id).sum(-1).float() / src_len
        x = x * (1 - 0.15 * 0.8) / (1 - ratio)[:, None, None]
    if attention_mask is not None:
        x = x * attention_mask.unsqueeze(-1).to(x.dtype)
    return x


def esm_layer(x, sd: SD, i: int, nheads: int, key_valid, cos, sin, eps: float):
    p = f"encoder.layer.{i}"
    B, L, E = x.shape
    hd = E // nheads
    h = _ln(x, sd, f"{p}.attention.LayerNorm", eps)
    q = _linear(h, sd, f"{p}.attention.self.query").view(B, L, nheads, hd).transpose(1, 2) * hd ** -0.5
    k = _linear(h, sd, f"{p}.attention.self.key").view(B, L, nheads, hd).transpose(1, 2)
    v = _linear(h, sd, f"{p}.attention.self.value").view(B, L, nheads, hd).transpose(1, 2)
    q, k = _rope(q, cos, sin), _rope(k, cos, sin)
    s = q @ k.transpose(-1, -2)
    if key_valid is not None:
        s = s.masked_fill(~key_valid[:, None, None, :], float("-inf"))
    ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, E)
    x = x + _linear(ctx, sd, f"{p}.attention.output.dense")
    h = _ln(x, sd, f"{p}.LayerNorm", eps)
    h = _gelu(_linear(h, sd, f"{p}.intermediate.dense"))
    return x + _linear(h, sd, f"{p}.output.dense")


def esm_encoder(ids, attention_mask, sd: SD, num_layers: int, nheads: int, eps: float = 1e-5,
                mask_token_id: int = 32, token_dropout: bool = True):
    """EsmModel(add_pooling_layer=False).forward(input_ids, attention_mask).last_hidden_state."""
    x = esm_embeddings(ids, attention_mask, sd, mask_token_id, token_dropout)
    L, E = x.shape[1], x.shape[2]
    cos, sin = rope_tables(L, E // nheads)
    key_valid = attention_mask.bool() if attention_mask is not None else None
    for i in range(num_layers):
        x = esm_layer(x, sd, i, nheads, key_valid, cos, sin, eps)
    return _ln(x, sd, "encoder.emb_layer_norm_after", eps)


def pool(x, valid: Optional[torch.Tensor], mode: str):
    """position-0 pooling (rna_clip_codes.ipynb:1948-1949) or the masked mean that
    configuration_hybrid_clip.py:109 `use_mean_pooling` declares (fair-esm mean: tf_clip_codes:1188)."""
    if mode == "first":
        return x[:, 0]
    if valid is None:
        return x.mean(1)
    w = valid.to(x.dtype)
    return (x * w[..., None]).sum(1) / w.sum(1, keepdim=True)
