"""Sequence encoders on the libclipk kernels: ESM-2 (pre-LN, RoPE) and the post-LN transformer of
nn.TransformerEncoderLayer.  Each encoder is ONE autograd.Function whose forward/backward walk the layers
by hand (explicit kernel sequence, explicit saved activations): no per-op autograd graph, bf16 activations
between GEMMs, f32 residual stream, f32 statistics.

Reference:
  * ESM-2: third-party transformers.EsmModel called at triple_flow/3_esm_integration.py:77-80,118-119
    (modeling_esm.py:225-270 embeddings, :362-384 attention, :429-438/:517-521 blocks, :552-553 final LN);
    state_dict keys follow EsmModel so HF checkpoints load as they are.
  * post-LN: nn.TransformerEncoderLayer as built at current/rna_clip_codes.ipynb:1911-1923 and described by the
    "transformer" architecture of run1/configuration_hybrid_clip.py:153-157 (6 x 768, 8 heads, ffn 2048, gelu).
    Keys follow RNARBPCLIPEncoder: layers.{i}.self_attn.in_proj_weight, ..., layernorm.weight.
"""
from __future__ import annotations

import math
from typing import List, Optional

import os

import torch
import torch.nn as nn

from . import functional as KF
from . import ops


def _rope_tables(L: int, hd: int, device, theta: float = 10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous().to(device), fr.sin().contiguous().to(device)


# Weight-gradient GEMMs are off the backward's critical path (only the optimiser consumes them), so they can run on
# a side HIP stream next to the dgrad / attention / LayerNorm chain of the same tower.  Off unless enabled.
WGRAD_SIDE_STREAM = False
_SIDE = {}


PREROTATE_QK = os.environ.get("CLIPK_PREROTATE", "1") != "0"   # ESM: RoPE on q / k once after the qkv GEMM, not at every staging
ROPE_IN_QKV_EPILOGUE = os.environ.get("CLIPK_ROPE_EPILOGUE", "1") != "0"   # hd in {16, 32, 64}: rotate in the qkv GEMM's epilogue
DIRECT_PARAM_GRADS = True      # weight-gradient kernels accumulate straight into existing .grad buffers
# ESM heads whose dim does not tile the GEMM epilogue's 64-column wave slices (ESM-2-8M / 35M / 150M: 16 handled above, 24,
# 32 ...): the q and k sections of the fused qkv projection are computed in PAIR-INTERLEAVED head order (bf16 weight copies
# with permuted rows, csrc/common.h il_src), so that RoPE runs in the projection's epilogue on neighbouring columns - on the
# f32 value, before its one bf16 rounding - and the attention forward neither rotates nor writes rotated rows back.  q . k
# is invariant under a common order of the head dim; the backward's RoPE^T and the weight-gradient rows undo the order.
# Whole-head attention shapes only (128 < L <= 256, head dim <= 32); CLIPK_ROPE_INTERLEAVED=0 keeps the in-place rotation.
ROPE_INTERLEAVED = os.environ.get("CLIPK_ROPE_INTERLEAVED", "1") != "0"


def _direct_ok(t):
    g = t.grad if (t is not None and t.is_leaf) else None      # e.g. ESM's fused qkv weight is a cat(), not a leaf
    return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == t.device


def _fused_views(parts):
    """[p0, p1, p2] stored back to back (FusedAdamW's flat buffer honours flat_param_groups) with .grad views laid
    out the same way -> (data view, grad view) of the concatenation along dim 0, zero-copy; else None."""
    p0 = parts[0]
    if not all(q.is_leaf and q.requires_grad and q.grad is not None and q.is_contiguous() and q.grad.is_contiguous()
               and q.dtype == torch.float32 and q.grad.dtype == torch.float32 for q in parts):
        return None
    d_ptr, g_ptr = p0.data_ptr(), p0.grad.data_ptr()
    for q in parts:
        if q.data_ptr() != d_ptr or q.grad.data_ptr() != g_ptr or q.shape[1:] != p0.shape[1:]:
            return None
        d_ptr += q.numel() * 4
        g_ptr += q.numel() * 4
    rows = sum(q.shape[0] for q in parts)
    shape = (rows,) + tuple(p0.shape[1:])
    stride = p0.stride()
    try:
        data = torch.as_strided(p0.data, shape, stride)
        grad = torch.as_strided(p0.grad, shape, stride)
    except RuntimeError:                                       # views would leave the storage: not back to back
        return None
    return data, grad


def _ln_bwd(dy, x, w, b, mean, rstd, pool=None, **kw):
    """ops.layernorm_bwd whose dgamma / dbeta go straight into w.grad / b.grad when those buffers exist (then the
    returned parameter grads are None, as in _wgrad).  pool = (row_weight, B, L): dy is the gradient of the mean-pooled
    rows [B, cols] (backward of ops.layernorm_meanpool_fwd)."""
    if pool is not None:
        wrow, B, L = pool
        if DIRECT_PARAM_GRADS and _direct_ok(w) and _direct_ok(b):
            dx, dxb, _, _ = ops.layernorm_meanpool_bwd(dy, wrow, x, w, mean, rstd, B, L, dgamma=w.grad, dbeta=b.grad,
                                                       accumulate=True, **kw)
            return dx, dxb, None, None
        return ops.layernorm_meanpool_bwd(dy, wrow, x, w, mean, rstd, B, L, **kw)
    if DIRECT_PARAM_GRADS and _direct_ok(w) and _direct_ok(b):
        dx, dxb, _, _ = ops.layernorm_bwd(dy, x, w, None, mean, rstd, dgamma=w.grad, dbeta=b.grad, accumulate=True, **kw)
        return dx, dxb, None, None
    return ops.layernorm_bwd(dy, x, w, None, mean, rstd, **kw)


def _wgrad(dy, x, lin=None, il=(0, 0)):
    """dW, db = wgrad(dy, x), optionally enqueued on the calling stream's side stream.  il: dy's leading columns are in
    pair-interleaved head order; the gradient rows land in the original order."""
    # With pre-allocated .grad buffers (FusedAdamW keeps them as views of one flat buffer) the reduce kernel adds
    # into them directly and autograd gets None: saves one torch `add` launch per parameter and step (289 of them
    # in the config-2 model) and the temporary.  Same semantics as AccumulateGrad: grad += dW.
    if DIRECT_PARAM_GRADS and lin is not None and lin.gw is not None and lin.gb is not None:
        direct, kw = True, dict(dw=lin.gw, dbias=lin.gb, accumulate=True)
    else:
        direct = DIRECT_PARAM_GRADS and lin is not None and _direct_ok(lin.w) and _direct_ok(lin.b)
        kw = dict(dw=lin.w.grad, dbias=lin.b.grad, accumulate=True) if direct else dict(want_bias=True)
    if il[1]:
        kw["il"] = il
    if not (WGRAD_SIDE_STREAM and dy.is_cuda):
        r = ops.gemm_wgrad(dy, x, **kw)
        return (None, None) if direct else r
    cur = torch.cuda.current_stream()
    side = _SIDE.get(cur.cuda_stream)
    if side is None:
        side = _SIDE[cur.cuda_stream] = torch.cuda.Stream()
    side.wait_event(cur.record_event())            # dy / x are ready at this point of the calling stream
    with torch.cuda.stream(side):
        dw, db = ops.gemm_wgrad(dy, x, **kw)
    dy.record_stream(side)
    x.record_stream(side)
    return (None, None) if direct else (dw, db)


def _join_side(*tensors):
    """Make the calling stream wait for its side stream (end of a stack's backward)."""
    if not WGRAD_SIDE_STREAM:
        return
    cur = torch.cuda.current_stream()
    side = _SIDE.get(cur.cuda_stream)
    if side is not None:
        cur.wait_stream(side)
        for t in tensors:
            if t is not None:
                t.record_stream(cur)


def _bucket_begin(module):
    """A forward of the stack that will be differentiated: one more backward has to finish before the stack's gradient
    bucket is final (micro-batches, a stack applied twice in one forward, several backward() calls per optimiser step).
    A bucket that a sharded FusedAdamW has already sent in this step is taken back (`_grad_bucket_begin`)."""
    module._bucket_pending = getattr(module, "_bucket_pending", 0) + 1
    cb = getattr(module, "_grad_bucket_begin", None)
    if cb is not None:
        cb()


def _bucket_backward_begin(module):
    """Start of a stack's backward: if this stack's bucket has ALREADY been sent in this step (a count that went wrong, a
    backward nobody announced), take it back before new gradient is written into it - it is reduced again when final."""
    cb = getattr(module, "_grad_bucket_begin", None)
    if cb is not None:
        cb()


def _bucket_done(module, grads):
    """End of a stack's backward: with every parameter gradient written straight into the flat .grad buffer (all
    returned grads None) and no other backward of this stack outstanding, the stack's gradient bucket is final, and a
    sharded FusedAdamW may start its reduce-scatter on a side stream under the rest of the backward
    (optim.FusedAdamW._reduce_bucket)."""
    module._bucket_pending = max(0, getattr(module, "_bucket_pending", 0) - 1)
    cb = getattr(module, "_grad_bucket_done", None)
    if cb is not None and module._bucket_pending == 0 and all(g is None for g in grads):
        cb()


class _Lin:
    """Kernel-side view of one Linear: f32 master params + cached bf16 W / W^T.  gw / gb: explicit gradient buffers
    (fused views of several parameters' .grad), used instead of w.grad / b.grad by the direct-accumulation path.
    il_cache (optional): a second cache whose copies have their leading rows in pair-interleaved head order (`il`)."""

    def __init__(self, w, b, cache: KF.WeightCache, gw=None, gb=None, version_fn=None, il_cache=None):
        self.w, self.b = w, b
        self.gw, self.gb = gw, gb
        self.il = (0, 0)
        if il_cache is not None:                               # this forward uses the interleaved copies only
            self.il = il_cache.il
            self.wb, self.wtb = il_cache.get(w, version_fn)
        else:
            self.wb, self.wtb = cache.get(w, version_fn)


# =================================================================================================
# ESM-2 (pre-LN) stack
# =================================================================================================
def _varlen_prerot(meta):
    """Packed batch whose sequences all fit the whole-head attention kernels: q / k are rotated in place by the forward
    kernel (clipk_attn_varlen_fwd_rot) and the backward runs one kernel per (sequence, head)."""
    B, L, H, D, mask, rope, eps, seq = meta
    return PREROTATE_QK and seq is not None and rope is not None and ops.varlen_whole_head_applies(seq[1], D)


# What the FFN keeps for its backward: the bf16 pre-activation u (GELU' recomputed from it), or - default - GELU'(u) itself
# as 8-bit codes written by the fc1 epilogue next to GELU(u) (clipk.h aux_dtype): 1 instead of 2 bytes per hidden
# element in the two store-bound FFN epilogues of every layer, no erf in the backward one.  The forward is untouched;
# the backward's GELU' factor carries an absolute error <= 0.0025.  CLIPK_GELU_AUX=bf16 keeps the pre-activation.
GELU_AUX_U8 = os.environ.get("CLIPK_GELU_AUX", "u8") != "bf16"


def _esm_layer_fwd(x, p, meta, keep=True):
    """x: f32 [T,d].  p: dict of this layer's tensors.  Returns y f32 [T,d] and the saved activations."""
    B, L, H, D, mask, rope, eps, seq = meta
    _, h1, m1, r1 = ops.layernorm_fwd(x, p["ln1_w"], p["ln1_b"], eps, want_f32=False, want_bf16=True)
    if seq is None and PREROTATE_QK and ROPE_IN_QKV_EPILOGUE and rope is not None and D in (16, 32, 64):
        # heads that tile the GEMM's 64-column wave slices (ESM-2-650M: 20 x 64): q / k leave the projection's epilogue
        # already rotated - no second pass over them.  (hd = 24 of the 35M model does not tile: branch below.)
        qkv = ops.gemm_nt(h1, p["qkv"].wb, bias=p["qkv"].b, rope=(rope[0], rope[1], L, D, 2 * H * D))
        ctx, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=None, q_scale=D ** -0.5)
        x2 = ops.gemm_nt(ctx, p["out"].wb, bias=p["out"].b, residual=x, out_dtype=torch.float32)
        return _esm_layer_ffn_fwd(x, x2, p, eps, keep, (h1, m1, r1, qkv, ctx, lse))
    if p["qkv"].il[1]:
        # pair-interleaved q / k (see ROPE_INTERLEAVED): rotated in the projection's epilogue; attention as it is
        il_hd, il_rows = p["qkv"].il
        qkv = ops.gemm_nt(h1, p["qkv"].wb, bias=p["qkv_b_il"], rope=(rope[0], rope[1], L, D, il_rows), rope_interleaved=True)
        ctx, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=None, q_scale=D ** -0.5)
        x2 = ops.gemm_nt(ctx, p["out"].wb, bias=p["out"].b, residual=x, out_dtype=torch.float32)
        return _esm_layer_ffn_fwd(x, x2, p, eps, keep, (h1, m1, r1, qkv, ctx, lse))
    qkv = ops.gemm_nt(h1, p["qkv"].wb, bias=p["qkv"].b)
    if seq is not None and _varlen_prerot(meta):           # packed batch of short sequences: whole-head kernel, q / k
        ctx, lse = ops.attn_varlen_fwd_rot_(qkv, seq[0], seq[1], H, D, rope, q_scale=D ** -0.5)   # rotated in place
    elif seq is not None:                                  # packed variable-length batch: rows [cu[b], cu[b+1])
        ctx, lse = ops.attn_varlen_fwd(qkv, seq[0], seq[1], H, D, rope=rope, q_scale=D ** -0.5)
    elif PREROTATE_QK and rope is not None:
        # RoPE once, in place: the attention kernels would otherwise rotate every K row 5x and every Q row 4x per
        # layer while staging it.  `qkv` (saved for backward) then holds rotated q / k.
        # (one call; for the short ESM heads also one kernel, which rotates the rows while it stages them.)
        ctx, lse = ops.attn_fwd_rot_(qkv, B, L, H, D, rope, key_mask=mask, q_scale=D ** -0.5)
    else:
        ctx, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5)
    x2 = ops.gemm_nt(ctx, p["out"].wb, bias=p["out"].b, residual=x, out_dtype=torch.float32)
    return _esm_layer_ffn_fwd(x, x2, p, eps, keep, (h1, m1, r1, qkv, ctx, lse))


def _esm_layer_ffn_fwd(x, x2, p, eps, keep, att):
    h1, m1, r1, qkv, ctx, lse = att
    _, h2, m2, r2 = ops.layernorm_fwd(x2, p["ln2_w"], p["ln2_b"], eps, want_f32=False, want_bf16=True)
    if keep:
        g, u = ops.gemm_nt(h2, p["fc1"].wb, bias=p["fc1"].b, act="gelu", out_preact=True,
                           aux_u8=GELU_AUX_U8 and ops.gelu_aux_u8_applies(h2.shape[1], p["fc1"].wb.shape[0]))
    else:                                                  # frozen encoder: nothing is kept for a backward
        g, u = ops.gemm_nt(h2, p["fc1"].wb, bias=p["fc1"].b, act="gelu"), None
    y = ops.gemm_nt(g, p["fc2"].wb, bias=p["fc2"].b, residual=x2, out_dtype=torch.float32)
    return y, (x, h1, m1, r1, qkv, ctx, lse, x2, h2, m2, r2, g, u)


# Pre-LN (ESM) backward: the residual-stream GRADIENT between the layers is bf16 (it is needed in bf16 anyway, as the
# operand of the next dgrad / wgrad GEMMs): the LayerNorm-backward kernels then read and write 10 instead of 16 bytes
# per element.  Gradients only — the forward residual stream stays f32 (loss parity).  Gradient cosines vs the oracle
# stay at 0.9999 (tests/test_gpu_configs.py); CLIPK_ESM_F32_GRAD_STREAM=1 restores the f32 stream.
ESM_BF16_GRAD_STREAM = os.environ.get("CLIPK_ESM_F32_GRAD_STREAM", "0") != "1"


def _esm_layer_bwd(dy, dyb, p, saved, meta, need_dx_bf16, need_dx_f32=True):
    """dy f32 [T,d] or None (bf16 gradient stream) + its bf16 copy dyb.  Returns dx f32 (or None), dx bf16 (or None)
    and the parameter grads."""
    B, L, H, D, mask, rope, eps, seq = meta
    x, h1, m1, r1, qkv, ctx, lse, x2, h2, m2, r2, g, u = saved
    if dyb is None:
        dyb = ops.to_bf16(dy)
    lowp = dy is None
    gr = {}
    du = ops.gemm_nt(dyb, p["fc2"].wtb, dact_aux=u, dact="gelu")               # dgrad fused with GELU'
    gr["fc2_w"], gr["fc2_b"] = _wgrad(dyb, g, p["fc2"])
    dh2 = ops.gemm_nt(du, p["fc1"].wtb)
    gr["fc1_w"], gr["fc1_b"] = _wgrad(du, h2, p["fc1"])
    dx2, dx2b, gr["ln2_w"], gr["ln2_b"] = _ln_bwd(dh2, x2, p["ln2_w"], p["ln2_b"], m2, r2, dx_add=dyb if lowp else dy,
                                                  want_f32=not lowp, want_bf16=True)
    dctx = ops.gemm_nt(dx2b, p["out"].wtb)
    gr["out_w"], gr["out_b"] = _wgrad(dx2b, ctx, p["out"])
    if seq is not None:
        dqkv = ops.attn_varlen_bwd(qkv, ctx, dctx, lse, seq[0], seq[1], H, D, rope=rope, q_scale=D ** -0.5,
                                   prerotated=_varlen_prerot(meta))
    else:
        dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5,
                            prerotated=2 if p["qkv"].il[1] else (PREROTATE_QK and rope is not None))
    dh1 = ops.gemm_nt(dqkv, p["qkv"].wtb)
    gr["qkv_w"], gr["qkv_b"] = _wgrad(dqkv, h1, p["qkv"], il=p["qkv"].il)
    want32 = need_dx_f32 or not lowp
    dx, dxb, gr["ln1_w"], gr["ln1_b"] = _ln_bwd(dh1, x, p["ln1_w"], p["ln1_b"], m1, r1, dx_add=dx2b if lowp else dx2,
                                                want_f32=want32, want_bf16=need_dx_bf16 or lowp)
    return dx, dxb, gr


_ESM_KEYS = ["ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]


class EsmStackFn(torch.autograd.Function):
    """ids -> final-LayerNorm hidden states [B*L, d] (f32); pool = True: -> their masked mean per sequence [B, d], the
    final LayerNorm and the pooling in one pass (clipk_layernorm_meanpool_*: the normalised rows are never written)."""

    @staticmethod
    def forward(ctx, module, ids, mask_u8, row_scale, seq, pool, *flat):
        nl = module.num_layers
        table, fin_w, fin_b = flat[0], flat[1], flat[2]
        B, L = ids.shape                                    # packed batches come as [T, 1] with seq = (cu, max_len)
        d, H = module.hidden_size, module.num_heads
        D = d // H
        rope = module.rope(L if seq is None else seq[1], ids.device)
        meta = (B, L, H, D, mask_u8, rope, module.eps, seq)
        x = ops.embed_fwd(ids, table, row_scale=row_scale, mask=mask_u8.view(-1) if mask_u8 is not None else None,
                          mask_token_id=module.mask_token_id if module.token_dropout else -1)
        layers, saved = [], []
        need_bwd = any(ctx.needs_input_grad)        # frozen encoder (3_esm_integration.py:83-84): keep no activations
        if need_bwd:
            _bucket_begin(module)
        # head dims the tiled RoPE epilogue cannot take (24: ESM-2-35M) get the pair-interleaved one (ROPE_INTERLEAVED)
        use_il = (ROPE_INTERLEAVED and PREROTATE_QK and ROPE_IN_QKV_EPILOGUE and seq is None and D % 8 == 0
                  and D not in (16, 32, 64) and d % 32 == 0)
        il_idx = module.il_index(ids.device) if use_il else None
        for i in range(nl):
            t = flat[3 + 12 * i: 3 + 12 * (i + 1)]
            caches = module.layer_caches[i]
            gw, gb = module._qkv_grads[i] if len(getattr(module, "_qkv_grads", ())) == nl else (None, None)
            # the fused qkv view's own _version never moves: its bf16 copy is keyed on the three source Parameters
            p = {"ln1_w": t[0], "ln1_b": t[1],
                 "qkv": _Lin(t[2], t[3], caches[0], gw, gb, module.qkv_version_fn(i),
                             il_cache=module.layer_caches_il[i] if use_il else None),
                 "out": _Lin(t[4], t[5], caches[1]),
                 "ln2_w": t[6], "ln2_b": t[7], "fc1": _Lin(t[8], t[9], caches[2]), "fc2": _Lin(t[10], t[11], caches[3])}
            if use_il:
                p["qkv_b_il"] = t[3].detach().index_select(0, il_idx)   # the bias in the projection's column order
            x, s = _esm_layer_fwd(x, p, meta, keep=need_bwd)
            layers.append(p)
            saved.append(s if need_bwd else None)
        ctx.pool = None
        if pool:
            y, mf, rf, wrow = ops.layernorm_meanpool_fwd(x, fin_w, fin_b, module.eps, B, L,
                                                            mask_u8.view(-1) if mask_u8 is not None else None)
            ctx.pool = (wrow, B, L)
        else:
            y, _, mf, rf = ops.layernorm_fwd(x, fin_w, fin_b, module.eps, want_f32=True)
        ctx.module, ctx.meta, ctx.layers, ctx.saved = module, meta, layers, saved
        ctx.fin = (x, fin_w, fin_b, mf, rf)
        ctx.ids, ctx.row_scale = ids, row_scale
        ctx.table = table
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved is None:
            raise RuntimeError("EsmStackFn: second backward through the same forward: the stack frees each layer's "
                               "activations as its backward consumes them (retain_graph is not supported)")
        module, meta = ctx.module, ctx.meta
        _bucket_backward_begin(module)
        x, fin_w, fin_b, mf, rf = ctx.fin
        nl = module.num_layers
        grads: List[Optional[torch.Tensor]] = [None] * (3 + 12 * nl)
        lowp = ESM_BF16_GRAD_STREAM
        dx, dxb, grads[1], grads[2] = _ln_bwd(dy.contiguous(), x, fin_w, fin_b, mf, rf, want_f32=not lowp, want_bf16=True,
                                              pool=ctx.pool)
        for i in reversed(range(nl)):
            # the last layer processed (i = 0) hands an f32 gradient to the embedding backward
            dx, dxb, gr = _esm_layer_bwd(dx, dxb, ctx.layers[i], ctx.saved[i], meta, need_dx_bf16=i > 0,
                                         need_dx_f32=(i == 0) or not lowp)
            ctx.saved[i] = None                                     # free this layer's activations now
            for j, k in enumerate(_ESM_KEYS):
                grads[3 + 12 * i + j] = gr[k]
        if ctx.needs_input_grad[6]:
            mask_u8 = meta[4]
            table = ctx.table
            direct = DIRECT_PARAM_GRADS and _direct_ok(table)       # the kernel accumulates: straight into table.grad
            dtable = table.grad if direct else torch.zeros(table.shape, dtype=torch.float32, device=dx.device)
            ops.embed_bwd(ctx.ids, dx, dtable, row_scale=ctx.row_scale,
                          mask=mask_u8.view(-1) if mask_u8 is not None else None,
                          mask_token_id=module.mask_token_id if module.token_dropout else -1)
            grads[0] = None if direct else dtable
        ctx.layers = ctx.saved = ctx.table = None
        _join_side(*grads)
        _bucket_done(module, grads)
        return (None, None, None, None, None, None, *grads)


class _EsmSelf(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)


class _Dense(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)


class _EsmAttention(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.self = _EsmSelf(d)
        self.output = _Dense(d, d)
        self.LayerNorm = nn.LayerNorm(d, eps=eps)


class _EsmLayer(nn.Module):
    def __init__(self, d, f, eps):
        super().__init__()
        self.attention = _EsmAttention(d, eps)
        self.intermediate = _Dense(d, f)
        self.output = _Dense(f, d)
        self.LayerNorm = nn.LayerNorm(d, eps=eps)


class _EsmEncoderModules(nn.Module):
    def __init__(self, nl, d, f, eps):
        super().__init__()
        self.layer = nn.ModuleList([_EsmLayer(d, f, eps) for _ in range(nl)])
        self.emb_layer_norm_after = nn.LayerNorm(d, eps=eps)


class _EsmEmbeddings(nn.Module):
    def __init__(self, vocab, d, pad):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, d, padding_idx=pad)


ESM2_SHAPES = {                      # (layers, hidden, heads, ffn) — SURVEY App. A-17
    "esm2_t6_8M_UR50D": (6, 320, 20, 1280),
    "esm2_t12_35M_UR50D": (12, 480, 20, 1920),
    "esm2_t30_150M_UR50D": (30, 640, 20, 2560),
    "esm2_t33_650M_UR50D": (33, 1280, 20, 5120),
    "esm2_t36_3B_UR50D": (36, 2560, 40, 10240),         # the other two names triple_flow/1_config.py:177-181 accepts
    "esm2_t48_15B_UR50D": (48, 5120, 40, 20480),
}


class ESM2Encoder(nn.Module):
    """ESM-2 encoder with transformers.EsmModel's parameter names (add_pooling_layer=False, rotary positions).

    forward(input_ids [B,L] int64, attention_mask [B,L] or None) -> last_hidden_state [B, L, d] (f32).
    q/k/v weights are kept as three nn.Linear (checkpoint compatible) and fused into one [3d, d] GEMM operand.
    """

    grad_bucket = True          # FlatParams stores this stack's parameters as one bucket (multi-GPU gradient overlap)

    def __init__(self, num_layers=12, hidden_size=480, num_heads=20, intermediate_size=1920, vocab_size=33,
                 pad_token_id=1, mask_token_id=32, layer_norm_eps=1e-5, token_dropout=True, initializer_range=0.02):
        super().__init__()
        self.num_layers, self.hidden_size, self.num_heads = num_layers, hidden_size, num_heads
        self.intermediate_size, self.eps = intermediate_size, layer_norm_eps
        self.mask_token_id, self.token_dropout, self.pad_token_id = mask_token_id, token_dropout, pad_token_id
        self.embeddings = _EsmEmbeddings(vocab_size, hidden_size, pad_token_id)
        self.encoder = _EsmEncoderModules(num_layers, hidden_size, intermediate_size, layer_norm_eps)
        self.layer_caches = [[KF.WeightCache() for _ in range(4)] for _ in range(num_layers)]
        hd = hidden_size // num_heads
        # bf16 copies of the fused qkv weight with the q and k rows in pair-interleaved head order (ROPE_INTERLEAVED)
        self.layer_caches_il = [KF.WeightCache(il=(hd, 2 * hidden_size)) for _ in range(num_layers)]
        self._il_idx = {}
        self._rope = {}
        self._fused = None
        self._init(initializer_range)

    @classmethod
    def from_name(cls, name: str, **kw):
        nl, d, h, f = ESM2_SHAPES[name]
        return cls(nl, d, h, f, **kw)

    def _init(self, std):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, std)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, 0.0, std)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()

    def rope(self, L, device):
        key = (L, str(device))
        if key not in self._rope:
            self._rope[key] = _rope_tables(L, self.hidden_size // self.num_heads, device)
        return self._rope[key]

    def il_index(self, device):
        """int64 [3 d]: the original output column each column of the pair-interleaved qkv projection holds."""
        key = str(device)
        if key not in self._il_idx:
            self._il_idx[key] = ops.il_source_rows(3 * self.hidden_size, self.hidden_size // self.num_heads,
                                                   2 * self.hidden_size).to(device)
        return self._il_idx[key]

    def qkv_version_fn(self, i):
        s = self.encoder.layer[i].attention.self
        return KF.params_version(s.query.weight, s.key.weight, s.value.weight)

    def flat_param_groups(self):
        """Parameters FusedAdamW's flat buffer should store back to back: q/k/v weights, q/k/v biases per layer."""
        groups = []
        for lyr in self.encoder.layer:
            s = lyr.attention.self
            groups.append([s.query.weight, s.key.weight, s.value.weight])
            groups.append([s.query.bias, s.key.bias, s.value.bias])
        return groups

    def _flat_params(self):
        """Per layer: ln1 w,b | fused qkv w,b | out w,b | ln2 w,b | fc1 w,b | fc2 w,b.  The fused qkv tensors are
        zero-copy views (plus their gradient views in self._qkv_grads) when the three Linears are stored back to
        back with .grad buffers in place (DIRECT_PARAM_GRADS), else torch.cat results that autograd splits again."""
        flat = [self.embeddings.word_embeddings.weight, self.encoder.emb_layer_norm_after.weight,
                self.encoder.emb_layer_norm_after.bias]
        self._qkv_grads = []
        for lyr in self.encoder.layer:
            s = lyr.attention.self
            fw = _fused_views([s.query.weight, s.key.weight, s.value.weight]) if DIRECT_PARAM_GRADS else None
            fb = _fused_views([s.query.bias, s.key.bias, s.value.bias]) if DIRECT_PARAM_GRADS else None
            if fw is not None and fb is not None and torch.is_grad_enabled():
                qkv_w, qkv_b = fw[0], fb[0]
                self._qkv_grads.append((fw[1], fb[1]))
            else:
                qkv_w = torch.cat([s.query.weight, s.key.weight, s.value.weight], 0)
                qkv_b = torch.cat([s.query.bias, s.key.bias, s.value.bias], 0)
                self._qkv_grads.append((None, None))
            flat += [lyr.attention.LayerNorm.weight, lyr.attention.LayerNorm.bias,
                     qkv_w, qkv_b,
                     lyr.attention.output.dense.weight, lyr.attention.output.dense.bias,
                     lyr.LayerNorm.weight, lyr.LayerNorm.bias,
                     lyr.intermediate.dense.weight, lyr.intermediate.dense.bias,
                     lyr.output.dense.weight, lyr.output.dense.bias]
        return flat

    def forward_pooled(self, input_ids, attention_mask=None):
        """Masked mean over the tokens of forward()'s last_hidden_state, [B, d]: same values as pooling the returned
        states, with the final LayerNorm and the mean in one kernel (nothing of size [B, L, d] is written)."""
        return self.forward(input_ids, attention_mask, _pool=True)

    def forward(self, input_ids, attention_mask=None, _pool=False):
        B, L = input_ids.shape
        mask_u8 = None
        row_scale = None
        if attention_mask is not None:
            mask_u8 = attention_mask.to(torch.uint8).contiguous()
        if self.token_dropout:
            # modeling_esm.py:252-259: scale by (1 - 0.15*0.8) / (1 - observed mask ratio); [B] scalars: plumbing
            src_len = attention_mask.sum(-1).float() if attention_mask is not None else \
                torch.full((B,), float(L), device=input_ids.device)
            ratio = (input_ids == self.mask_token_id).sum(-1).float() / src_len
            row_scale = ((1 - 0.15 * 0.8) / (1 - ratio)).contiguous()
        y = EsmStackFn.apply(self, input_ids.contiguous(), mask_u8, row_scale, None, bool(_pool), *self._flat_params())
        return y if _pool else y.view(B, L, self.hidden_size)

    def forward_packed(self, input_ids, cu_seqlens, max_len: int):
        """Packed variable-length batch: input_ids int64 [T] = the sequences back to back, cu_seqlens
        int32 [B+1] on the device, max_len = the longest sequence.  -> last_hidden_state [T, d].  No padded token
        exists anywhere: Linear / LayerNorm kernels see T real rows, attention runs per sequence."""
        T = input_ids.numel()
        row_scale = None
        if self.token_dropout:
            # per-sequence (1 - 0.15 * 0.8) / (1 - mask ratio), expanded per token ([T] scalars: plumbing)
            lens = (cu_seqlens[1:] - cu_seqlens[:-1]).long()
            seg = torch.repeat_interleave(torch.arange(lens.numel(), device=input_ids.device), lens)
            nmask = torch.zeros(lens.numel(), device=input_ids.device).index_add_(
                0, seg, (input_ids == self.mask_token_id).float())
            row_scale = ((1 - 0.15 * 0.8) / (1 - nmask / lens.float()))[seg].contiguous()
        y = EsmStackFn.apply(self, input_ids.reshape(T, 1).contiguous(), None, row_scale,
                             (cu_seqlens.contiguous(), int(max_len)), False, *self._flat_params())
        return y


# =================================================================================================
# post-LN (nn.TransformerEncoderLayer) stack
# =================================================================================================
# Post-LN layers: x = LN(x + sublayer(x)).  The LayerNorm output is needed in bf16 anyway (operand of the next GEMM); with
# this switch it is ALSO the residual of the next add (GEMM epilogue modes RES16 / PRES16) instead of a second, f32 copy,
# and the backward's residual-path gradient between LayerNorm-backward and the dgrad epilogues is the bf16 tensor too:
# 24 instead of 36 (forward) and 24 instead of 44 (backward) bytes per element and layer of HBM traffic around the
# LayerNorms.  Normalised O(1) values lose nothing that the bf16 GEMM operands had not lost already: full-depth loss
# parity 1.2e-5 / 1.9e-4 (ragged), gradient cosines 0.9999 (tests/test_gpu_configs.py).  The pre-LN ESM stream is
# un-normalised and stays f32 (bf16 there moved the loss by 6e-3).
# Per stack: `TransformerSeqEncoder(residual_dtype="bf16" | "f32")`.  bf16 is the default of the BASELINE towers (mean
# pooling over 256 rows averages the rounding away); the notebook / tri-modal encoders pool ONE row (position 0,
# rna_clip_codes.ipynb:1948-1949) and default to the f32 residual: at B = 32 the bf16 one alone moved their loss by
# ~1e-3 (round 3, tools/exp_notebook_parity.py).  CLIPK_POSTLN_F32_RESIDUAL=1 forces f32 everywhere.
POSTLN_BF16_RESIDUAL = os.environ.get("CLIPK_POSTLN_F32_RESIDUAL", "0") != "1"


# (the pre-LayerNorm sums s1 / s2 stay f32: as bf16 they cost 1.7e-3 of loss on the small-width test model for 0.4 % of
# the step, measured in round 2)
_SUM_DT = torch.float32


def _post_layer_fwd(x, xb, p, meta, dr=None):
    """dr = None or (p_drop, seed_attn, seed_drop1, seed_ffn, seed_drop2): the four nn.Dropout sites of
    nn.TransformerEncoderLayer (attention probabilities; out_proj output; FFN activation; linear2 output), each a
    counter-based mask recomputed in the backward from the same seed."""
    B, L, H, D, mask, act, eps, qs, seq, res16 = meta
    da, d1, df, d2 = (None,) * 4 if dr is None else tuple((dr[0], sd) for sd in dr[1:])
    qkv = ops.gemm_nt(xb, p["in"].wb, bias=p["in"].b)
    if seq is not None:
        ctx, lse = ops.attn_varlen_fwd(qkv, seq[0], seq[1], H, D, rope=None, q_scale=qs, dropout=da)
    else:
        ctx, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=None, q_scale=qs, dropout=da)
    s1 = ops.gemm_nt(ctx, p["out"].wb, bias=p["out"].b, residual=x, out_dtype=_SUM_DT, dropout=d1)
    if res16:
        _, x1b, m1, r1 = ops.layernorm_fwd(s1, p["n1_w"], p["n1_b"], eps, want_f32=False, want_bf16=True)
        x1 = x1b
    else:
        x1, x1b, m1, r1 = ops.layernorm_fwd(s1, p["n1_w"], p["n1_b"], eps, want_f32=True, want_bf16=True)
    if act == "gelu":
        g, u = ops.gemm_nt(x1b, p["fc1"].wb, bias=p["fc1"].b, act="gelu", out_preact=True, dropout=df,
                           aux_u8=GELU_AUX_U8 and df is None
                           and ops.gelu_aux_u8_applies(x1b.shape[1], p["fc1"].wb.shape[0]))
    else:
        g = ops.gemm_nt(x1b, p["fc1"].wb, bias=p["fc1"].b, act="relu", dropout=df)
        u = g                                    # relu'(pre) == relu'(relu(pre)); a dropped element has g = 0 either way
    s2 = ops.gemm_nt(g, p["fc2"].wb, bias=p["fc2"].b, residual=x1, out_dtype=_SUM_DT, dropout=d2)
    if res16:
        _, yb, m2, r2 = ops.layernorm_fwd(s2, p["n2_w"], p["n2_b"], eps, want_f32=False, want_bf16=True)
        y = yb
    else:
        y, yb, m2, r2 = ops.layernorm_fwd(s2, p["n2_w"], p["n2_b"], eps, want_f32=True, want_bf16=True)
    return y, yb, (xb, qkv, ctx, lse, s1, x1b, m1, r1, g, u, s2, m2, r2)


def _post_layer_bwd(dy, p, saved, meta, dr=None, need_dx=True):
    B, L, H, D, mask, act, eps, qs, seq, res16 = meta
    da, d1, df, d2 = (None,) * 4 if dr is None else tuple((dr[0], sd) for sd in dr[1:])
    xb, qkv, ctx, lse, s1, x1b, m1, r1, g, u, s2, m2, r2 = saved
    gr = {}
    # the bf16 copy of d s2 is the gradient of linear2's dropped-out output: masked (drop2); the f32 one is the
    # residual-path gradient
    # (bf16 mode: the LayerNorm-backward kernels emit only the bf16 gradient, which is both the next GEMM operand — with
    # the dropout mask when there is one — and the residual-path gradient added in the dgrad epilogue)
    lowp = res16 and dr is None
    ds2, ds2b, gr["n2_w"], gr["n2_b"] = _ln_bwd(dy, s2, p["n2_w"], p["n2_b"], m2, r2, want_f32=not lowp, want_bf16=True,
                                                dropout_bf16=d2)
    du = ops.gemm_nt(ds2b, p["fc2"].wtb, dact_aux=u, dact=act, dropout=df)     # x mask(ffn) x act'(u)
    gr["fc2_w"], gr["fc2_b"] = _wgrad(ds2b, g, p["fc2"])
    if lowp:
        dx1 = ops.gemm_nt(du, p["fc1"].wtb, residual=ds2b)                          # bf16 out = acc + bf16 residual
    else:
        dx1 = ops.gemm_nt(du, p["fc1"].wtb, residual=ds2, out_dtype=torch.float32)  # + residual-path gradient
    gr["fc1_w"], gr["fc1_b"] = _wgrad(du, x1b, p["fc1"])
    ds1, ds1b, gr["n1_w"], gr["n1_b"] = _ln_bwd(dx1, s1, p["n1_w"], p["n1_b"], m1, r1, want_f32=not lowp, want_bf16=True,
                                                dropout_bf16=d1)
    dctx = ops.gemm_nt(ds1b, p["out"].wtb)
    gr["out_w"], gr["out_b"] = _wgrad(ds1b, ctx, p["out"])
    if seq is not None:
        dqkv = ops.attn_varlen_bwd(qkv, ctx, dctx, lse, seq[0], seq[1], H, D, rope=None, q_scale=qs, dropout=da)
    else:
        dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, B, L, H, D, key_mask=mask, rope=None, q_scale=qs, dropout=da)
    if not need_dx:                                   # first layer of a stack fed with features that need no gradient
        dx = None                                     # (the RNA tower of ProteinRNACLIP): its input dgrad is never used
    elif lowp:
        dx = ops.gemm_nt(dqkv, p["in"].wtb, residual=ds1b)
    else:
        dx = ops.gemm_nt(dqkv, p["in"].wtb, residual=ds1, out_dtype=torch.float32)
    gr["in_w"], gr["in_b"] = _wgrad(dqkv, xb, p["in"])
    return dx, gr


_POST_KEYS = ["in_w", "in_b", "out_w", "out_b", "n1_w", "n1_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "n2_w", "n2_b"]


class PostLNStackFn(torch.autograd.Function):
    """x f32 [B*L, E] -> final-LayerNorm output f32 [B*L, E]; pool = True: -> its masked mean per sequence [B, E]
    (final LayerNorm + pooling in one pass, as EsmStackFn)."""

    @staticmethod
    def forward(ctx, module, x, mask_u8, B, L, seq, drop, pool, *flat):
        nl = module.num_layers
        E, H = module.embed_dim, module.nhead
        D = module.head_dim_padded
        meta = (B, L, H, D, mask_u8, module.activation, module.eps, float(E // H) ** -0.5, seq,
                module.residual_dtype == "bf16" and POSTLN_BF16_RESIDUAL)
        x = x.contiguous()
        xb = ops.to_bf16(x)
        fin_w, fin_b = flat[0], flat[1]
        layers, saved = [], []
        need_bwd = any(ctx.needs_input_grad)
        if need_bwd:
            _bucket_begin(module)
        for i in range(nl):
            t = flat[2 + 12 * i: 2 + 12 * (i + 1)]
            c = module.layer_caches[i]
            p = {"in": _Lin(t[0], t[1], c[0]), "out": _Lin(t[2], t[3], c[1]), "n1_w": t[4], "n1_b": t[5],
                 "fc1": _Lin(t[6], t[7], c[2]), "fc2": _Lin(t[8], t[9], c[3]), "n2_w": t[10], "n2_b": t[11]}
            x, xb, s = _post_layer_fwd(x, xb, p, meta, None if drop is None else (drop[0],) + tuple(drop[1][i]))
            layers.append(p)
            saved.append(s if need_bwd else None)
        ctx.pool = None
        if pool:
            y, mf, rf, wrow = ops.layernorm_meanpool_fwd(x, fin_w, fin_b, module.final_eps, B, L,
                                                            mask_u8.view(-1) if mask_u8 is not None else None)
            ctx.pool = (wrow, B, L)
        else:
            y, _, mf, rf = ops.layernorm_fwd(x, fin_w, fin_b, module.final_eps, want_f32=True)
        ctx.module, ctx.meta, ctx.layers, ctx.saved = module, meta, layers, saved
        ctx.fin = (x, fin_w, fin_b, mf, rf)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved is None:
            raise RuntimeError("PostLNStackFn: second backward through the same forward: the stack frees each layer's "
                               "activations as its backward consumes them (retain_graph is not supported)")
        module, meta = ctx.module, ctx.meta
        _bucket_backward_begin(module)
        x, fin_w, fin_b, mf, rf = ctx.fin
        nl = module.num_layers
        grads: List[Optional[torch.Tensor]] = [None] * (2 + 12 * nl)
        dx, _, grads[0], grads[1] = _ln_bwd(dy.contiguous(), x, fin_w, fin_b, mf, rf, want_f32=True, pool=ctx.pool)
        for i in reversed(range(nl)):
            drop = ctx.drop
            dx, gr = _post_layer_bwd(dx, ctx.layers[i], ctx.saved[i], meta,
                                     None if drop is None else (drop[0],) + tuple(drop[1][i]),
                                     need_dx=i > 0 or ctx.needs_input_grad[1])
            ctx.saved[i] = None
            for j, k in enumerate(_POST_KEYS):
                grads[2 + 12 * i + j] = gr[k]
        ctx.layers = ctx.saved = None
        _join_side(*grads)
        _bucket_done(module, grads)
        if ctx.needs_input_grad[1] and dx is not None and dx.dtype != torch.float32:
            dx = ops.to_f32(dx.contiguous())
        return (None, dx if ctx.needs_input_grad[1] else None, None, None, None, None, None, None, *grads)


class _MHAParams(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = nn.Linear(E, E)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _PostLayerParams(nn.Module):
    """Same parameter names as nn.TransformerEncoderLayer."""

    def __init__(self, E, F_, eps):
        super().__init__()
        self.self_attn = _MHAParams(E)
        self.linear1 = nn.Linear(E, F_)
        self.linear2 = nn.Linear(F_, E)
        self.norm1 = nn.LayerNorm(E, eps=eps)
        self.norm2 = nn.LayerNorm(E, eps=eps)


class TransformerSeqEncoder(nn.Module):
    """Stack of post-LN layers + final LayerNorm == RNARBPCLIPEncoder (rna_clip_codes.ipynb:1911-1923).

    forward(x [B, L, E] f32, key_padding_mask [B, L] bool with True = PAD, as nn.TransformerEncoderLayer) -> [B, L, E].
    Attention runs over axis 1 (per sequence).  To reproduce the notebook's batch-axis attention quirk
     pass x.transpose(0, 1) and the matching mask — see RNARBPCLIPModel.
    """

    grad_bucket = True          # see ESM2Encoder
    takes_precision = True      # KF.set_linear_precision reaches this stack

    def __init__(self, embed_dim=768, num_layers=6, nhead=8, dim_feedforward=2048, activation="gelu",
                 layer_norm_eps=1e-12, final_eps=None, dropout: float = 0.0, residual_dtype: str = "bf16",
                 precision: str = "bf16"):
        super().__init__()
        if residual_dtype not in ("bf16", "f32"):
            raise ValueError(f"residual_dtype must be 'bf16' or 'f32', got {residual_dtype!r}")
        if precision not in ("bf16", "f32"):
            raise ValueError(f"precision must be 'bf16' or 'f32', got {precision!r}")
        self.residual_dtype = residual_dtype
        # "bf16": bf16-MFMA GEMMs / attention, f32 statistics (the north-star arithmetic of the BASELINE towers).
        # "f32": every Linear on the exact-f32 MFMA GEMM, attention on attention_f32.hip, f32 LayerNorms and residual
        # adds - the arithmetic of the reference's un-autocast callers (rna_clip_codes.ipynb:2061-2089); the default of
        # the position-0-pooled notebook / tri-modal models, whose encoders see B rows once the dead positions are gone.
        self.precision = precision
        self.embed_dim, self.num_layers, self.nhead = embed_dim, num_layers, nhead
        # nn.TransformerEncoderLayer's dropout (rna_clip_codes.ipynb:1915 uses 0.1): attention probabilities, out_proj
        # output, FFN activation, linear2 output.  In training mode each site draws a counter-based mask inside the
        # attention kernels / GEMM epilogues from a per-call seed (torch's CPU generator: torch.manual_seed applies);
        # the backward recomputes the masks, nothing is stored.  Eval mode and p = 0 are the parity setting.
        self.dropout = float(dropout)
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout} (the kernels' masks scale by 1 / (1 - p))")
        self.activation, self.eps = activation, layer_norm_eps
        self.final_eps = layer_norm_eps if final_eps is None else final_eps
        self.layers = nn.ModuleList([_PostLayerParams(embed_dim, dim_feedforward, layer_norm_eps)
                                     for _ in range(num_layers)])
        self.layernorm = nn.LayerNorm(embed_dim, eps=self.final_eps)
        self.layer_caches = [[KF.WeightCache() for _ in range(4)] for _ in range(num_layers)]
        hd = embed_dim // nhead
        self.head_dim_padded = (hd + 7) // 8 * 8          # kernels want head_dim % 8 == 0 (notebook: 120/8 = 15)

    def _attn_params(self, l):
        """in_proj / out_proj as the kernels consume them.  When head_dim % 8 != 0 every head is zero-padded to
        head_dim_padded (zero q/k/v rows, zero out_proj columns): identical arithmetic, differentiable views."""
        E, H, Dp = self.embed_dim, self.nhead, self.head_dim_padded
        D = E // H
        a = l.self_attn
        if Dp == D:
            return a.in_proj_weight, a.in_proj_bias, a.out_proj.weight
        pad = Dp - D
        w = torch.nn.functional.pad(a.in_proj_weight.view(3, H, D, E), (0, 0, 0, pad)).reshape(3 * H * Dp, E)
        b = torch.nn.functional.pad(a.in_proj_bias.view(3, H, D), (0, pad)).reshape(3 * H * Dp)
        wo = torch.nn.functional.pad(a.out_proj.weight.view(E, H, D), (0, pad)).reshape(E, H * Dp)
        return w, b, wo

    def _flat_params(self):
        flat = [self.layernorm.weight, self.layernorm.bias]
        for l in self.layers:
            w_in, b_in, w_out = self._attn_params(l)
            flat += [w_in, b_in, w_out,
                     l.self_attn.out_proj.bias, l.norm1.weight, l.norm1.bias, l.linear1.weight, l.linear1.bias,
                     l.linear2.weight, l.linear2.bias, l.norm2.weight, l.norm2.bias]
        return flat

    def forward_pooled(self, x, src_key_padding_mask=None):
        """Masked mean over the positions of forward()'s output, [B, E]: final LayerNorm + mean in one kernel."""
        return self.forward(x, src_key_padding_mask, _pool=True)

    def _forward_f32(self, x, mask_u8, B, L, drop, pool_out):
        """The stack in exact f32 (precision = "f32"): per layer qkv Linear -> attention -> out_proj + residual ->
        LayerNorm -> linear1 -> activation -> linear2 + residual -> LayerNorm, every product on clipk_gemm_f32 (residual
        adds in its epilogue) and clipk_attn_f32_*; un-padded head dim (120 / 8 = 15 runs as it is).  Dropout sites and
        mask indices are those of the bf16 stack (same seeds -> same dropped elements)."""
        E, H = self.embed_dim, self.nhead
        D = E // H
        h = x.reshape(B * L, E).float().contiguous()
        for i, l in enumerate(self.layers):
            a = l.self_attn
            dr = None if drop is None else [(drop[0], sd) for sd in drop[1][i]]      # attn, drop1, ffn, drop2
            if dr is None:                 # no active dropout: each half of the layer is one autograd node
                s1 = KF.AttnBlockF32Fn.apply(h, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                             B, L, H, D, mask_u8, float(D) ** -0.5)
                x1 = KF.layer_norm(s1, l.norm1.weight, l.norm1.bias, self.eps)
                s2 = KF.FFNBlockF32Fn.apply(x1, l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                                            self.activation)
                h = KF.layer_norm(s2, l.norm2.weight, l.norm2.bias, self.eps)
                continue
            qkv = KF.linear_f32(h, a.in_proj_weight, a.in_proj_bias)
            ctx = KF.attention_f32(qkv, B, L, H, D, mask_u8, float(D) ** -0.5, dr[0] if dr else None)
            if dr:
                s1 = KF.dropout_f32(KF.linear_f32(ctx, a.out_proj.weight, a.out_proj.bias), dr[1], addend=h)
            else:
                s1 = KF.linear_f32(ctx, a.out_proj.weight, a.out_proj.bias, addend=h)
            x1 = KF.layer_norm(s1, l.norm1.weight, l.norm1.bias, self.eps)
            g = KF.ActFn.apply(KF.linear_f32(x1, l.linear1.weight, l.linear1.bias), self.activation)
            if dr:
                g = KF.dropout_f32(g, dr[2])
                s2 = KF.dropout_f32(KF.linear_f32(g, l.linear2.weight, l.linear2.bias), dr[3], addend=x1)
            else:
                s2 = KF.linear_f32(g, l.linear2.weight, l.linear2.bias, addend=x1)
            h = KF.layer_norm(s2, l.norm2.weight, l.norm2.bias, self.eps)
        y = KF.layer_norm(h, self.layernorm.weight, self.layernorm.bias, self.final_eps).view(B, L, E)
        if pool_out:
            return pool(y, None if mask_u8 is None else mask_u8.view(B, L), "mean")
        return y

    def forward(self, x, src_key_padding_mask=None, _pool=False, _valid_u8=None):
        B, L, E = x.shape
        mask_u8 = _valid_u8                 # (internal callers that already hold the kernels' form: uint8 [B, L], 1 = valid)
        if src_key_padding_mask is not None:
            mask_u8 = (~src_key_padding_mask.bool()).to(torch.uint8).contiguous()      # kernels take 1 = valid
        if self.precision == "f32":
            return self._forward_f32(x, mask_u8, B, L, self._draw_dropout(), bool(_pool))
        y = PostLNStackFn.apply(self, x.reshape(B * L, E), mask_u8, B, L, None, self._draw_dropout(), bool(_pool),
                                *self._flat_params())
        return y if _pool else y.view(B, L, E)

    def _draw_dropout(self):
        """None (eval mode / p = 0) or (p, [[seed_attn, seed_drop1, seed_ffn, seed_drop2] per layer]): 32-bit seeds from
        torch's default CPU generator, so torch.manual_seed() makes a training run reproducible."""
        if not (self.training and self.dropout > 0.0):
            return None
        seeds = torch.randint(0, 2 ** 31 - 1, (self.num_layers, 4), dtype=torch.int64).tolist()
        return (self.dropout, seeds)

    def forward_packed(self, x, cu_seqlens, max_len: int):
        """Packed variable-length batch: x f32 [T, E] = the sequences back to back, cu_seqlens int32
        [B+1] on the device, max_len = the longest sequence.  -> [T, E]; equals forward() on the padded batch with its
        key-padding mask, row for row, without any padded row going through a kernel."""
        T, E = x.shape
        B = cu_seqlens.numel() - 1
        if self.precision == "f32":
            raise NotImplementedError("packed variable-length batches run on the bf16 kernels only (precision='bf16'): "
                                      "the exact-f32 attention kernels take padded [B, L] batches")
        return PostLNStackFn.apply(self, x, None, B, int(max_len), (cu_seqlens.contiguous(), int(max_len)),
                                   self._draw_dropout(), False, *self._flat_params())


class PoolFn(torch.autograd.Function):
    """mode 'first' (rna_clip_codes.ipynb:1948) or masked 'mean' (configuration_hybrid_clip.py:109)."""

    @staticmethod
    def forward(ctx, x, mask_u8, mode):
        B, L, d = x.shape
        ctx.meta = (B, L, mode)
        ctx.mask = mask_u8
        return ops.pool_fwd(x.contiguous().view(B * L, d), B, L, mask=mask_u8.view(-1) if mask_u8 is not None else None,
                            mode=mode)

    @staticmethod
    def backward(ctx, dy):
        B, L, mode = ctx.meta
        m = ctx.mask
        dx = ops.pool_bwd(dy.contiguous(), B, L, mask=m.view(-1) if m is not None else None, mode=mode)
        return dx.view(B, L, -1), None, None


def pool(x, valid_mask=None, mode: str = "mean"):
    mask_u8 = valid_mask.to(torch.uint8).contiguous() if valid_mask is not None else None
    return PoolFn.apply(x, mask_u8, 0 if mode == "first" else 1)


class PoolPackedFn(torch.autograd.Function):
    """Pooling of a packed variable-length batch: x [T, d], sequence b = rows [cu[b], cu[b+1])."""

    @staticmethod
    def forward(ctx, x, cu_seqlens, mode):
        ctx.meta = (x.shape[0], mode)
        ctx.cu = cu_seqlens
        return ops.pool_varlen_fwd(x.contiguous(), cu_seqlens, mode)

    @staticmethod
    def backward(ctx, dy):
        T, mode = ctx.meta
        return ops.pool_varlen_bwd(dy.contiguous(), ctx.cu, T, mode), None, None


def pool_packed(x, cu_seqlens, mode: str = "mean"):
    return PoolPackedFn.apply(x, cu_seqlens, 0 if mode == "first" else 1)
