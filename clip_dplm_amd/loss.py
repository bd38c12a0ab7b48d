"""Fused CLIP / InfoNCE loss on the simce kernels — the B x B (or B_local x B_global) logits never exist.

Reference losses covered (weights select the variant):
  * one-sided  CE(S, arange)                      old/ablation.py:16, run1/full.py:133      (w_row=1, w_col=0)
  * symmetric  (CE(S) + CE(S^T)) / 2              current/rna_clip_codes.ipynb:1952-1953    (0.5, 0.5)
  * cache-negative variant                        old/clip_opt.py:130-151                   (0.5, 0.5, cache=...)
  * global batch over ranks                       old/clip_opt.py:102-112 — but differentiable

Multi-GPU scheme: one all-gather of the stacked embeddings [2, B_l, P], one
all-gather of the two LSE vectors [2, B_l]; every rank then computes the COMPLETE gradient of the global
loss w.r.t. its own rows locally — no embedding-gradient reduce-scatter is needed.  Parameter gradients are
summed across ranks afterwards by the optimiser's reduce-scatter.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops

# kernel namespace; tests of the rank bookkeeping (gloo, CPU) substitute a torch restatement here — the product
# itself never does: ops.* raise on anything but device tensors.
_kernels = ops


def _gather_cat(t: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)          # concatenation along dim 0 (works on nccl and gloo)
    return out.view((world,) + tuple(t.shape))


_INF = {}


def _inf_like(t: torch.Tensor) -> torch.Tensor:
    """A read-only +inf vector of t's shape, made once per (device, length): the one-sided loss's unused direction."""
    key = (str(t.device), t.numel())
    v = _INF.get(key)
    if v is None:
        v = _INF[key] = torch.full((t.numel(),), float("inf"), dtype=torch.float32, device=t.device)
    return v.view(t.shape)


class ClipLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, scale, w_row, w_col, cache, group):
        a, b = a.contiguous(), b.contiguous()
        scale = scale.reshape(1).contiguous()
        bl = a.shape[0]
        if group is not None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            both = _gather_cat(torch.stack([a, b]), group)                 # [W, 2, Bl, P]
            a_g = both[:, 0].reshape(world * bl, -1)
            b_g = both[:, 1].reshape(world * bl, -1)
        else:
            world, rank, a_g, b_g = 1, 0, a, b
        off = rank * bl
        bg = world * bl
        lse_r, pos_r = _kernels.simce_lse(a, b_g, scale, label_offset=off, cache=cache)
        pos_c = None
        if w_col != 0.0:
            lse_c, pos_c = _kernels.simce_lse(b, a_g, scale, label_offset=off)
        else:
            lse_c = _inf_like(lse_r)                       # exp(s - inf) = 0: the unused direction contributes nothing
        if group is None:                                  # single process: sums, weights and the mean in one launch
            out = _kernels.ce_combine(lse_r, pos_r, lse_c if pos_c is not None else None, pos_c, w_row, w_col, bg)
            ctx.meta = (w_row, w_col, off, bg, cache)
            ctx.save_for_backward(a, b, a_g, b_g, scale, lse_r, lse_c, lse_r, lse_c)
            return out
        local = w_row * (lse_r - pos_r).sum()
        if pos_c is not None:
            local = local + w_col * (lse_c - pos_c).sum()
        if group is not None:
            lses = _gather_cat(torch.stack([lse_r, lse_c]), group)          # [W, 2, Bl]
            lse_r_g = lses[:, 0].reshape(-1).contiguous()
            lse_c_g = lses[:, 1].reshape(-1).contiguous()
            dist.all_reduce(local, group=group)
        else:
            lse_r_g, lse_c_g = lse_r, lse_c
        ctx.meta = (w_row, w_col, off, bg, cache)
        ctx.save_for_backward(a, b, a_g, b_g, scale, lse_r, lse_c, lse_r_g, lse_c_g)
        return local / bg

    @staticmethod
    def backward(ctx, dloss):
        a, b, a_g, b_g, scale, lse_r, lse_c, lse_r_g, lse_c_g = ctx.saved_tensors
        w_row, w_col, off, bg, cache = ctx.meta
        # rows of a: row-direction softmax uses their own LSE, column direction the keys' LSE
        # the incoming gradient (1.0 from loss.backward()) is folded into the kernels' 1 / Bg factor: no `grad * g` launches
        g = dloss.reshape(1).contiguous() if dloss.numel() == 1 else None
        da, dsa = _kernels.simce_grad(a, b_g, scale, lse_r, lse_c_g, w_row, w_col, 1.0 / bg, label_offset=off, cache=cache,
                                      upstream=g)
        # rows of b are the queries of the column direction
        db, _ = _kernels.simce_grad(b, a_g, scale, lse_c, lse_r_g, w_col, w_row, 1.0 / bg, label_offset=off, upstream=g)
        dscale = dsa.sum().reshape(1)          # this rank's rows only; the optimiser sums parameter grads over ranks
        return da, db, dscale, None, None, None, None


def clip_loss(a_embeds: torch.Tensor, b_embeds: torch.Tensor, logit_scale_exp: torch.Tensor, *,
              symmetric: bool = True, cache: Optional[torch.Tensor] = None, group=None,
              w_row: Optional[float] = None, w_col: Optional[float] = None) -> torch.Tensor:
    """InfoNCE over L2-normalised embeddings [B_local, P] (f32).  `logit_scale_exp` = exp(logit_scale)
    (already clamped if the model clamps, old/clip_opt.py:100).  With `group`, the batch is the concatenation
    over ranks in rank order and the returned value is the global-batch loss on every rank."""
    if w_row is None:
        w_row, w_col = (0.5, 0.5) if symmetric else (1.0, 0.0)
    if group is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        group = dist.group.WORLD
    if group is not None and dist.get_world_size(group) == 1 and not os.environ.get("CLIPK_FORCE_DIST"):
        group = None                         # (CLIPK_FORCE_DIST keeps the collective path for 1-rank RCCL rehearsals)
    return ClipLossFn.apply(a_embeds, b_embeds, logit_scale_exp, float(w_row), float(w_col), cache, group)


def contrastive_loss(x: torch.Tensor, y: torch.Tensor, temperature: float = 0.1, queue: Optional[torch.Tensor] = None,
                     group=None) -> torch.Tensor:
    """tong/utils/losses.py:4-19: InfoNCE with an optional memory queue — both inputs L2-normalised, the queue rows
    appended to the keys as extra negatives (detached), one-sided CE(x y^T / temperature, arange).  Same fused kernels as
    clip_loss (w_row = 1, w_col = 0, cache = queue): neither the [B, B + Q] logits nor the concatenated keys exist."""
    from . import functional as KF
    scale = torch.full((1,), 1.0 / float(temperature), dtype=torch.float32, device=x.device)
    # a private copy, as the reference's `queue.clone().detach()`: the queue is overwritten in place by the next
    # enqueue, which may come before this loss's backward
    cache = None if queue is None else queue.detach().to(dtype=torch.float32).clone()
    return clip_loss(KF.l2_normalize(x), KF.l2_normalize(y), scale, symmetric=False, cache=cache, group=group)


_TRI_PAIRS = ((0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1))      # (cell,pert) (pert,cell) (cell,prot) ...


class TriModalLossFn(torch.autograd.Function):
    """The three pairwise symmetric InfoNCE losses of current/tf_clip_codes (1).ipynb:13150-13163 on ONE logit scale:
    six directed similarity + LSE problems in one launch (clipk_simce_lse_pairs), six gradient problems in one more
    (clipk_simce_grad_pairs).  Returns (cell_pert, cell_protein, pert_protein) losses; any combination of upstream
    gradients is honoured."""

    @staticmethod
    def forward(ctx, cell, pert, prot, scale):
        E = torch.stack([cell, pert, prot]).contiguous()                # [3, B, P]
        sc = scale.reshape(1).contiguous()
        lse, pos = _kernels.simce_lse_pairs(E, _TRI_PAIRS, sc)
        per = (lse - pos).mean(1)                                       # six one-directional CE values
        ctx.save_for_backward(E, sc, lse)
        ctx.scale_shape = scale.shape
        return 0.5 * (per[0] + per[1]), 0.5 * (per[2] + per[3]), 0.5 * (per[4] + per[5])

    @staticmethod
    def backward(ctx, g_cp, g_ce, g_pe):
        E, sc, lse = ctx.saved_tensors
        B = E.shape[1]
        dX, dsc = _kernels.simce_grad_pairs(E, _TRI_PAIRS, sc, lse, 0.5, 0.5, 1.0 / B)
        # problem (a, b) holds d L_ab / d E_a complete (both directions): combine per modality — [B, P] adds: plumbing
        dcell = g_cp * dX[0] + g_ce * dX[2]
        dpert = g_cp * dX[1] + g_pe * dX[4]
        dprot = g_ce * dX[3] + g_pe * dX[5]
        dscale = (g_cp * dsc[0].sum() + g_ce * dsc[2].sum() + g_pe * dsc[4].sum()).reshape(ctx.scale_shape)
        return dcell, dpert, dprot, dscale


def tri_modal_loss(cell_embed: torch.Tensor, pert_embed: torch.Tensor, protein_embed: torch.Tensor,
                   logit_scale_exp: torch.Tensor, group=None):
    """Tri-modal contrastive objective of current/tf_clip_codes (1).ipynb:13150-13176: three pairwise symmetric
    InfoNCE losses sharing one logit_scale on the fused similarity + CE kernels (no B x B logits).  Single process:
    one batched launch per pass for all three pairs (TriModalLossFn); with a process group: three global-batch
    clip_loss calls.  Returns the loss entries of the reference's ContrastiveModel.forward dict."""
    if group is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        group = dist.group.WORLD
    if group is not None and dist.get_world_size(group) > 1:
        cp = clip_loss(cell_embed, pert_embed, logit_scale_exp, symmetric=True, group=group)
        ce = clip_loss(cell_embed, protein_embed, logit_scale_exp, symmetric=True, group=group)
        pe = clip_loss(pert_embed, protein_embed, logit_scale_exp, symmetric=True, group=group)
    else:
        cp, ce, pe = TriModalLossFn.apply(cell_embed.contiguous(), pert_embed.contiguous(), protein_embed.contiguous(),
                                          logit_scale_exp)
    return {"loss": cp + ce + pe, "cell_pert_loss": cp, "cell_protein_loss": ce, "pert_protein_loss": pe}
