"""CPU restatement (PyTorch f32) of the reference's CLIP modules and losses.  Test infrastructure only.

Functional style: every function takes a flat dict of tensors whose keys are the reference's own
state_dict keys, so golden weights exported from the reference load without renaming.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _linear(x, sd: SD, prefix: str):
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def _ln(x, sd: SD, prefix: str, eps: float):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def clip_encoder(x, sd: SD, prefix: str, num_layers: int, eps: float):
    """old/clip.py:8-17 — for layer: x = relu(layer(x)); return layernorm(x)."""
    for i in range(num_layers):
        x = torch.relu(_linear(x, sd, f"{prefix}.layers.{i}"))
    return _ln(x, sd, f"{prefix}.layernorm", eps)


def projection_head(x, sd: SD, prefix: str):
    """old/clip.py:20-36 — Linear -> LayerNorm -> GELU -> Dropout(eval: identity) -> Linear -> LayerNorm
    (nn.LayerNorm default eps 1e-5)."""
    h = _linear(x, sd, f"{prefix}.projection.0")
    h = _gelu(_ln(h, sd, f"{prefix}.projection.1", 1e-5))
    h = _linear(h, sd, f"{prefix}.projection.4")
    return _ln(h, sd, f"{prefix}.projection.5", 1e-5)


def optimized_projection_head(x, sd: SD, prefix: str):
    """old/clip_opt.py:9-44 and current/rna_clip_codes.ipynb:1901-1909 — skip(x) + layer_scale * MLP3(x)."""
    h = _gelu(_ln(_linear(x, sd, f"{prefix}.projection.0"), sd, f"{prefix}.projection.1", 1e-5))
    h = _gelu(_ln(_linear(h, sd, f"{prefix}.projection.4"), sd, f"{prefix}.projection.5", 1e-5))
    h = _ln(_linear(h, sd, f"{prefix}.projection.8"), sd, f"{prefix}.projection.9", 1e-5)
    return _linear(x, sd, f"{prefix}.skip") + sd[f"{prefix}.layer_scale"] * h


def l2_normalize(x, eps: float = 1e-12):
    """F.normalize(x, dim=-1) — old/clip.py:63-64."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def rna_protein_clip_forward(sd: SD, a_values, b_values, a: str = "rna", b: str = "protein",
                             num_layers=(2, 2), eps=(1e-12, 1e-12)):
    """old/clip.py:56-73 (RNAProteinCLIPModule.forward) and :93-110 (DiffMap variant, a='diffmap')."""
    ea = clip_encoder(a_values, sd, f"{a}_model", num_layers[0], eps[0])
    eb = clip_encoder(b_values, sd, f"{b}_model", num_layers[1], eps[1])
    ea = l2_normalize(projection_head(ea, sd, f"{a}_projection"))
    eb = l2_normalize(projection_head(eb, sd, f"{b}_projection"))
    scale = sd["logit_scale"].exp()
    return {f"logits_per_{a}_{b}": (ea @ eb.t()) * scale, f"{a}_embeds": ea, f"{b}_embeds": eb}


def ce_diag(logits):
    """nn.CrossEntropyLoss()(logits, arange(B)) — old/ablation.py:16."""
    return (torch.logsumexp(logits, dim=1) - logits.diag()).mean()


def clip_loss_symmetric(logits):
    """(CE(S) + CE(S^T)) / 2 — current/rna_clip_codes.ipynb:1952-1953."""
    return 0.5 * (ce_diag(logits) + ce_diag(logits.t()))


def optimized_clip_forward(sd: SD, diffmap_values, protein_values, cache: Optional[torch.Tensor], num_layers=(2, 2),
                           eps=(1e-12, 1e-12)):
    """old/clip_opt.py:83-128 without the (stateful) cache update: `cache` are the rows
    protein_embedding_cache[:cache_ptr] as they stand when the similarities are computed."""
    ed = clip_encoder(diffmap_values, sd, "diffmap_model", num_layers[0], eps[0])
    ep = clip_encoder(protein_values, sd, "protein_model", num_layers[1], eps[1])
    ed = l2_normalize(optimized_projection_head(ed, sd, "diffmap_projection"))
    ep = l2_normalize(optimized_projection_head(ep, sd, "protein_projection"))
    scale = sd["logit_scale"].exp().clamp(max=100)
    out = {"logits_per_diffmap_protein": (ed @ ep.t()) * scale, "diffmap_embeds": ed, "protein_embeds": ep}
    if cache is not None:
        out["logits_per_diffmap_cache"] = (ed @ cache.t()) * scale
    return out


def optimized_clip_loss(outputs):
    """old/clip_opt.py:130-151 — what actually runs: hard-label CE on [sim | sim_cache] plus CE on sim^T, /2
    (the smoothed labels are computed and discarded, App. A-6)."""
    s = outputs["logits_per_diffmap_protein"]
    comb = torch.cat([s, outputs["logits_per_diffmap_cache"]], dim=1) if "logits_per_diffmap_cache" in outputs else s
    return 0.5 * (ce_diag(comb) + ce_diag(s.t()))


def tri_modal_losses(cell_embed, pert_embed, protein_embed, logit_scale):
    """current/tf_clip_codes (1).ipynb:13143-13165: three pairwise symmetric CE losses on one logit_scale."""
    s = logit_scale.exp()
    cp = clip_loss_symmetric((cell_embed @ pert_embed.t()) * s)
    ce = clip_loss_symmetric((cell_embed @ protein_embed.t()) * s)
    pe = clip_loss_symmetric((pert_embed @ protein_embed.t()) * s)
    return cp + ce + pe, cp, ce, pe


def memory_queue_enqueue(queue: torch.Tensor, ptr: int, embeddings: torch.Tensor):
    """tong/utils/data.py:154-184 `MemoryQueue.enqueue_dequeue` as a pure function: FIFO write of the batch at `ptr`
    with true wrap-around (the tail of the batch continues at row 0); returns (queue, new ptr).  The reference returns
    the WHOLE queue, zero rows that were never written included."""
    size, n = queue.shape[0], embeddings.shape[0]
    q = queue.clone()
    e = embeddings.detach()
    if ptr + n > size:
        first = size - ptr
        q[ptr:] = e[:first]
        q[: n - first] = e[first:]
        return q, n - first
    q[ptr:ptr + n] = e
    return q, (ptr + n) % size


def contrastive_loss_queue(x, y, temperature: float = 0.1, queue: Optional[torch.Tensor] = None):
    """tong/utils/losses.py:4-19: normalise both, append the queue rows to the keys, one-sided CE(x y^T / tau, arange)."""
    x = l2_normalize(x)
    y = l2_normalize(y)
    if queue is not None:
        y = torch.cat([y, queue.detach()], dim=0)
    sim = (x @ y.t()) / temperature
    return (torch.logsumexp(sim, dim=1) - sim[:, : x.shape[0]].diagonal()).mean()
