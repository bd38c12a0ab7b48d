"""autograd.Function wrappers of the libclipk kernels (per-op granularity, used by the drop-in modules).

Forward and backward both run hand-written HIP kernels; torch.autograd only routes tensors.  bf16 copies of
the f32 master weights (W and W^T) are cached per parameter and refreshed when the parameter changes.
"""
from __future__ import annotations

import os
from typing import Optional

import weakref

import torch

from . import ops

_WEIGHT_EPOCH = 0


def mark_weights_dirty() -> None:
    """Call after updating parameters outside torch's version counter (the fused optimiser kernels do)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


# hipGraph capture of a training step (training.GraphedTrainStep): whether a bf16 copy is stale is a HOST decision, which a
# replay never repeats.  While a step is being captured, (i) the copies the batched refresh covers (buffers in place, master
# weight = the parameters' own storage) are left to it - FusedAdamW.step() launches it right after the update, the launch
# becomes part of the graph and every replay leaves every such copy current for the next one; (ii) every other copy (per-call
# operands: padded / concatenated weights, first use) is rebuilt at its point of use, into the same buffers.
_CAPTURE_FORCE = False


def set_capture_force(on: bool) -> None:
    global _CAPTURE_FORCE
    _CAPTURE_FORCE = bool(on)


def capture_force() -> bool:
    return _CAPTURE_FORCE


def weight_epoch() -> int:
    """Counter of out-of-band parameter updates (mark_weights_dirty): part of every derived-operand cache key."""
    return _WEIGHT_EPOCH


def _cache_key(w: torch.Tensor, version=None):
    """Identity + version of a master weight.  For a fused zero-copy view of several Parameters (ESM's qkv:
    torch.as_strided of the flat buffer, whose own _version never moves) the caller passes the sum of the source
    Parameters' versions, so load_state_dict / re-initialisation / another optimiser invalidate the copy too."""
    if version is not None:                  # a DERIVED operand (fused view, torch.cat / pad of Parameters, rebuilt per
        return ("src", version, _WEIGHT_EPOCH, tuple(w.shape))      # call): its identity is that of its sources
    return (w.data_ptr(), w._version, _WEIGHT_EPOCH, tuple(w.shape))


def _batched_refresh_covers(c, w, require_cuda: bool = True) -> bool:
    """refresh_weight_caches() can rebuild this cache: buffers in place, a contiguous leaf master weight that IS the
    parameters' storage (never a per-call copy: `derived`)."""
    if w is None or c.derived or c.wb is None or c.wtb is None or (require_cuda and not w.is_cuda) \
            or not w.is_leaf or not w.is_contiguous():
        return False
    return w.dim() == 2 and c.wb.shape == w.shape


class WeightCache:
    """bf16 W [N,K] and W^T [K,N] of an f32 master weight, refreshed lazily (or all at once, refresh_weight_caches)."""

    __slots__ = ("wb", "wtb", "key", "src", "src_version", "derived", "il", "__weakref__")

    def __init__(self, il=(0, 0)):
        self.il = (int(il[0]), int(il[1]))   # (head rows, rows): those rows of both copies in pair-interleaved head order
        self.wb = self.wtb = None
        self.key = None
        self.src = None                      # the master weight this cache was last built from
        self.src_version = None              # callable -> version token of a fused view's source Parameters (or None)
        self.derived = False                 # src is a per-call COPY (torch.cat / pad of Parameters), not their storage
        _CACHES.add(self)

    def get(self, w: torch.Tensor, version_fn=None):
        self.src_version = version_fn
        ver = None if version_fn is None else version_fn()
        # A version_fn operand is either a zero-copy view of its source Parameters (ESM's fused qkv: same storage, the
        # batched refresh may re-read it) or a copy rebuilt per call (cat / pad): the copy of an EARLIER call holds
        # earlier weights, so refresh_weight_caches must never rebuild from it (a no_grad forward left a
        # leaf copy in `src`, the next optimiser step refreshed from it and stamped the new key on stale data).
        self.derived = ver is not None and w.data_ptr() not in {v[0] for v in ver if isinstance(v, tuple)}
        key = _cache_key(w, ver)
        self.src = w                         # always the operand of THIS call, hit or miss
        if key != self.key or (_CAPTURE_FORCE and not _batched_refresh_covers(self, w)):
            src = w.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            self.wb, self.wtb = ops.cast_transpose(src, w_out=self.wb if self.wb is not None and self.wb.shape == w.shape else None,
                                                   wt_out=self.wtb if self.wtb is not None and self.wtb.shape == (w.shape[1], w.shape[0]) else None,
                                                   il=self.il)
            self.key = key
            self.src = w
        return self.wb, self.wtb


_CACHES = weakref.WeakSet()
_BATCH_DESC = {}                             # (device, pointer tuple) -> int64 [n, 5] device descriptor table


def _stale_caches(require_cuda: bool = True):
    """(cache, master weight, new key) of every cache the batched refresh may rebuild: buffers in place, a contiguous
    leaf master weight that IS the parameters' storage (never a per-call copy: `derived`), key out of date."""
    todo = []
    for c in list(_CACHES):
        w = c.src
        if not _batched_refresh_covers(c, w, require_cuda):
            continue
        key = _cache_key(w, None if c.src_version is None else c.src_version())
        if key != c.key:
            todo.append((c, w, key))
    return todo


def refresh_weight_caches() -> int:
    """Rebuild every stale bf16 copy in ONE kernel launch on the current stream.  Called by FusedAdamW.step() right
    after the update: the lazy path costs one launch per weight (76 in the config-2 model) at the next forward.
    Covers caches that already have their buffers and a contiguous leaf master weight; the rest stay lazy."""
    todo = _stale_caches()
    if not todo:
        return 0
    by_dev = {}
    for item in todo:
        by_dev.setdefault(item[1].device, []).append(item)
    for dev, items in by_dev.items():
        rows = tuple((w.data_ptr(), c.wb.data_ptr(), c.wtb.data_ptr(), w.shape[0], w.shape[1], c.il[0] if c.il[1] else 2, c.il[1])
                     for c, w, _ in items)
        desc = _BATCH_DESC.get((dev, rows))
        if desc is None:
            _BATCH_DESC.clear()              # pointers changed (new model / reallocated flat buffer)
            desc = torch.tensor(rows, dtype=torch.int64).to(dev)
            _BATCH_DESC[(dev, rows)] = desc
        with torch.cuda.device(dev):
            ops.cast_transpose_batched(desc)
        for c, w, key in items:
            c.key = key
    return len(todo)


def _bf16(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.bfloat16:
        return x if x.is_contiguous() else x.contiguous()
    return ops.to_bf16(x.contiguous())


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b)   (old/clip.py:11,16,27,31).  x: [M,K] f32 or bf16; y dtype selectable."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache: WeightCache, act, out_dtype, version_fn=None):
        wb, wtb = cache.get(weight, version_fn)
        xb = _bf16(x)
        need_pre = act == "gelu"
        r = ops.gemm_nt(xb, wb, bias=bias, act=act, out_dtype=out_dtype, out_preact=need_pre)
        y, pre = r if need_pre else (r, None)
        ctx.act = act
        ctx.x_dtype = x.dtype
        ctx.has_bias = bias is not None
        # relu'(pre) == relu'(y): the output itself is the aux for ReLU
        aux = pre if need_pre else (y if act == "relu" and y.dtype == torch.bfloat16 else None)
        if act == "relu" and aux is None:
            aux = _bf16(y)
        ctx.save_for_backward(xb, wtb, aux)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wtb, aux = ctx.saved_tensors
        dy = dy.contiguous()
        g = ops.dact(dy, aux, ctx.act) if aux is not None else _bf16(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(g, wtb, out_dtype=ctx.x_dtype)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = ops.gemm_wgrad(g, xb, want_bias=ctx.has_bias)
        return dx, dw, db, None, None, None, None


class LinearF32Fn(torch.autograd.Function):
    """y = x W^T + b (+ addend) in EXACT f32 (clipk_gemm_f32: f32-input MFMA, bitwise an fmaf chain) — the arithmetic of
    the reference's own fp32 callers (old/ablation.py:9-18 runs old/clip.py without autocast; the notebook models of
    rna_clip_codes.ipynb:1925-1954 likewise).  `addend` [M, N]: the residual the Linear's output is added to
    (x + out_proj(ctx), x + linear2(h) of nn.TransformerEncoderLayer) in the GEMM's epilogue.  Opt-in per Linear
    (`KLinear.precision = "f32"`, `set_linear_precision`); the bf16-MFMA LinearFn is the default."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend=None):
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        return ops.gemm_f32(x, weight, bias=bias, addend=None if addend is None else addend.contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        bias = ctx.bias
        dy = dy.contiguous()
        dx = ops.gemm_f32(dy, weight, trans_b=True) if ctx.needs_input_grad[0] else None
        dw, db = _linear_f32_param_grads(dy, x, weight, bias, ctx.needs_input_grad[1],
                                         bias is not None and ctx.needs_input_grad[2])
        return dx, dw, db, (dy if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3] else None)


def _linear_f32_param_grads(dy, x, weight, bias, want_w, want_b):
    """(dW, db) of an exact-f32 Linear for autograd - or (None, None) after adding them straight into the parameters'
    .grad (views of FusedAdamW's flat buffer) where those exist: no temporary, no AccumulateGrad add launch per
    parameter.  Both gradients come from ONE launch when they go the same way (clipk_gemm_wgrad_f32)."""
    dw = db = None
    w_here, b_here = want_w and _grad_in_place(weight), want_b and _grad_in_place(bias)
    if want_w and want_b and w_here == b_here:
        if w_here:
            ops.gemm_wgrad_f32(dy, x, dw=weight.grad, dbias=bias.grad, accumulate=True)
        else:
            dw, db = ops.gemm_wgrad_f32(dy, x)
    else:
        if want_w:
            if w_here:
                ops.gemm_wgrad_f32(dy, x, dw=weight.grad, accumulate=True, want_bias=False)
            else:
                dw, _ = ops.gemm_wgrad_f32(dy, x, want_bias=False)
        if want_b:
            if b_here:
                ops.colsum_f32(dy, out=bias.grad, accumulate=True)
            else:
                db = ops.colsum_f32(dy)
    return dw, db


class JoinStreamsFn(torch.autograd.Function):
    """Identity on tensors that branches of the model produced on side HIP streams (the caller has already made its stream
    wait for them).  Its backward is the FIRST node of the backward pass to run; it asks the autograd engine to make the
    calling stream wait for the side streams once the pass is over.  The engine replays each branch on the stream of its
    forward and only syncs the streams of AccumulateGrad leaves - gradients the kernels add straight into `.grad` buffers
    on a side stream would otherwise race the optimiser."""

    @staticmethod
    def forward(ctx, streams, *tensors):
        ctx.streams = streams
        return tensors if len(tensors) > 1 else tensors[0]

    @staticmethod
    def backward(ctx, *grads):
        streams = ctx.streams

        def join():
            cur = torch.cuda.current_stream()
            for s in streams:
                cur.wait_stream(s)

        torch.autograd.Variable._execution_engine.queue_callback(join)
        return (None,) + grads


def branch_streams(n: int):
    """n HIP streams for parallel_branches (CLIPK_BRANCH_PRIORITY=high: created with high priority - experiment switch)."""
    pr = -1 if os.environ.get("CLIPK_BRANCH_PRIORITY") == "high" else 0
    return tuple(torch.cuda.Stream(priority=pr) for _ in range(n))


def parallel_branches(streams, thunks, inputs=()):
    """Run independent branches of a model (the towers of a contrastive model up to the loss) on HIP streams of their own:
    thunks[i]() is enqueued on streams[i] and returns ONE tensor; inputs[i] are the tensors it reads that the calling
    stream allocated.  The calling stream forks before and joins after (forward), autograd replays every branch on its
    stream (backward), JoinStreamsFn joins again when the backward pass ends.  Inside a hipGraph capture the branches become
    parallel branches of the graph.  Returns the outputs in order."""
    main = torch.cuda.current_stream()
    outs = []
    for s, f in zip(streams, thunks):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(f())
    for s, ins in zip(streams, inputs):
        for t in ins:
            if t is not None and t.is_cuda:
                t.record_stream(s)
    for s in streams:
        main.wait_stream(s)
    for t in outs:
        t.record_stream(main)
    return JoinStreamsFn.apply(tuple(streams), *outs)


class AttnBlockF32Fn(torch.autograd.Function):
    """s = h + out_proj(attention(in_proj(h))) of nn.TransformerEncoderLayer (rna_clip_codes.ipynb:1911-1921), exact f32,
    no active dropout, as ONE autograd node: the residual's gradient rides the in_proj input-gradient GEMM as its addend
    (per-op composition leaves that sum to an autograd `add` launch per block) and the host walks one node instead of three."""

    @staticmethod
    def forward(ctx, h, w_in, b_in, w_out, b_out, B, L, H, D, key_mask, q_scale):
        h = h.contiguous()
        qkv = ops.gemm_f32(h, w_in, bias=b_in)
        att, lse = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=key_mask, q_scale=q_scale)
        ctx.meta = (B, L, H, D, key_mask, q_scale)
        ctx.save_for_backward(h, w_in, b_in, w_out, b_out, qkv, att, lse)
        return ops.gemm_f32(att, w_out, bias=b_out, addend=h)

    @staticmethod
    def backward(ctx, ds):
        h, w_in, b_in, w_out, b_out, qkv, att, lse = ctx.saved_tensors
        B, L, H, D, key_mask, q_scale = ctx.meta
        ds = ds.contiguous()
        need = ctx.needs_input_grad
        dwo, dbo = _linear_f32_param_grads(ds, att, w_out, b_out, need[3], b_out is not None and need[4])
        datt = ops.gemm_f32(ds, w_out, trans_b=True)
        dqkv = ops.attn_f32_bwd(qkv, att, datt, lse, B, L, H, D, key_mask=key_mask, q_scale=q_scale)
        dwi, dbi = _linear_f32_param_grads(dqkv, h, w_in, b_in, need[1], b_in is not None and need[2])
        dh = ops.gemm_f32(dqkv, w_in, trans_b=True, addend=ds) if need[0] else None
        return dh, dwi, dbi, dwo, dbo, None, None, None, None, None, None


class FFNBlockF32Fn(torch.autograd.Function):
    """s = x + linear2(act(linear1(x))) of nn.TransformerEncoderLayer, exact f32, no active dropout, as one autograd node
    (see AttnBlockF32Fn)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        x = x.contiguous()
        u = ops.gemm_f32(x, w1, bias=b1)
        g = ops.act_fwd(u, act)
        ctx.act = act
        ctx.save_for_backward(x, w1, b1, w2, b2, u, g)
        return ops.gemm_f32(g, w2, bias=b2, addend=x)

    @staticmethod
    def backward(ctx, ds):
        x, w1, b1, w2, b2, u, g = ctx.saved_tensors
        ds = ds.contiguous()
        need = ctx.needs_input_grad
        dw2, db2 = _linear_f32_param_grads(ds, g, w2, b2, need[3], b2 is not None and need[4])
        du = ops.act_bwd(ops.gemm_f32(ds, w2, trans_b=True), u, ctx.act)
        dw1, db1 = _linear_f32_param_grads(du, x, w1, b1, need[1], b1 is not None and need[2])
        dx = ops.gemm_f32(du, w1, trans_b=True, addend=ds) if need[0] else None
        return dx, dw1, db1, dw2, db2, None


def _grad_in_place(p) -> bool:
    """A leaf parameter whose .grad buffer already exists (FusedAdamW.zero_grad keeps them as views of its flat buffer):
    kernels accumulate into it and autograd gets None - same semantics as AccumulateGrad (grad += dW)."""
    g = p.grad if (p is not None and p.is_leaf) else None
    return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device and g.shape == p.shape


def linear_f32(x, weight, bias=None, addend=None):
    """Exact-f32 Linear on 2-D rows (+ fused residual add)."""
    return LinearF32Fn.apply(x.float().contiguous(), weight, bias, addend)


class AttnF32Fn(torch.autograd.Function):
    """Self-attention on a fused f32 qkv tensor [B*L, 3*H*D] -> [B*L, H*D] in exact f32 (attention_f32.hip)."""

    @staticmethod
    def forward(ctx, qkv, B, L, H, D, key_mask, q_scale, dropout):
        qkv = qkv.contiguous()
        out, lse = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=key_mask, q_scale=q_scale, dropout=dropout)
        ctx.meta = (B, L, H, D, key_mask, q_scale, dropout)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, L, H, D, key_mask, q_scale, dropout = ctx.meta
        dqkv = ops.attn_f32_bwd(qkv, out, dout.contiguous(), lse, B, L, H, D, key_mask=key_mask, q_scale=q_scale,
                                dropout=dropout)
        return dqkv, None, None, None, None, None, None, None


def attention_f32(qkv, B, L, H, D, key_mask=None, q_scale=1.0, dropout=None):
    return AttnF32Fn.apply(qkv, B, L, H, D, key_mask, q_scale, dropout)


class DropoutF32Fn(torch.autograd.Function):
    """x * keep / (1 - p) (+ addend) with the kernels' counter-based mask (nothing stored: the backward re-draws it)."""

    @staticmethod
    def forward(ctx, x, dropout, addend=None):
        ctx.dropout = dropout
        return ops.dropout_f32(x.contiguous(), dropout, None if addend is None else addend.contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ops.dropout_f32(dy, ctx.dropout) if ctx.needs_input_grad[0] else None
        return dx, None, (dy if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] else None)


def dropout_f32(x, dropout, addend=None):
    return DropoutF32Fn.apply(x, dropout, addend)


def params_version(*params):
    """version_fn for an operand derived from these Parameters (concatenated / padded per call): the cached bf16 copies
    stay valid while none of the sources changed (keyed on the derived tensor's pointer the cache never hit)."""
    return lambda: tuple((p.data_ptr(), p._version) for p in params)


def linear(x, weight, bias, cache: WeightCache, act=None, out_dtype=torch.float32, precision: str = "bf16",
           version_fn=None):
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if precision == "f32":
        y = LinearF32Fn.apply(x2.float().contiguous(), weight, bias, None)
        if act is not None:
            y = ActFn.apply(y, act)                       # (out_dtype is a bf16-path hint: an f32 Linear stays f32)
    else:
        y = LinearFn.apply(x2, weight, bias, cache, act, out_dtype, version_fn)
    return y.reshape(*lead, y.shape[-1])


def set_linear_precision(module: torch.nn.Module, precision: str = "bf16") -> torch.nn.Module:
    """Select the arithmetic of every kernel-backed Linear under `module`: "bf16" (default of the BASELINE towers: bf16
    operands, f32 accumulate, the north-star arithmetic) or "f32" (exact-f32 MFMA: the MLP towers and heads of
    old/clip.py then reproduce the reference's fp32 forward to ~1e-6).  Post-LN transformer stacks
    (`TransformerSeqEncoder`) follow: "f32" runs their Linears, attention, LayerNorms and residual adds in f32
    (attention_f32.hip) - the default of the position-0-pooled notebook / tri-modal models.  The ESM-2 stack is bf16 only."""
    if precision not in ("bf16", "f32"):
        raise ValueError(f"precision must be 'bf16' or 'f32', got {precision!r}")
    for m in module.modules():
        if (hasattr(m, "_cache") and isinstance(m, torch.nn.Linear)) or getattr(m, "takes_precision", False):
            m.precision = precision
    return module


class LayerNormFn(torch.autograd.Function):
    """y = act(LayerNorm(x))  (old/clip.py:12,28-29,32); f32 in/out."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x = x.contiguous()
        y, _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps, act=act, want_f32=True)
        ctx.act = act
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _grad_in_place(gamma) and _grad_in_place(beta):
            # the column reduce adds straight into the parameters' .grad (no AccumulateGrad add launch per parameter) ...
            part = _ln_defer_buffer(gamma, x) if DEFER_LN_PARAM_GRADS else None
            if part is not None:
                # ... and it is DEFERRED: this LayerNorm's partial rows stay in a buffer of their own, one launch reduces
                # every LayerNorm of the pass when the pass is over
                dx = ops.layernorm_bwd(dy.contiguous(), x, gamma, beta, mean, rstd, act=ctx.act, want_f32=True,
                                       part_out=part[0])[0]
                _ln_defer(part, gamma, beta)
                return dx, None, None, None, None
            dx = ops.layernorm_bwd(dy.contiguous(), x, gamma, beta, mean, rstd, act=ctx.act, want_f32=True,
                                   dgamma=gamma.grad, dbeta=beta.grad, accumulate=True)[0]
            return dx, None, None, None, None
        dx, _, dg, db = ops.layernorm_bwd(dy.contiguous(), x, gamma, beta, mean, rstd, act=ctx.act, want_f32=True)
        return dx, dg, db, None, None


# Deferred LayerNorm parameter gradients.  The backward of a LayerNorm is two launches - the row pass, which leaves one
# [dgamma | dbeta] partial row per workgroup, and a column reduce over them - and in the models that pool one position the
# second is a 4-us launch on the critical chain, 20 times per step.  With `.grad` buffers in place (FusedAdamW) the row pass
# writes its partial rows to a buffer that belongs to this LayerNorm, and ONE batched launch reduces all of them when the
# backward pass is over (autograd-engine callback; after JoinStreamsFn's, so side-stream branches have been joined).
# MEASURED AND LEFT OFF (profiles/r04/ln_deferred_param_grads_ab_v1.txt): -3 % on config 1's one-stream captured step, nothing on
# the notebook step (1.891 vs 1.878 ms) and on the two-branch config-1 step - with the towers as parallel branches the
# reduces were already off the path that bounds the step.  CLIPK_DEFER_LN_GRADS=1 switches it on.
DEFER_LN_PARAM_GRADS = os.environ.get("CLIPK_DEFER_LN_GRADS", "0") == "1"
_LN_PART = {}          # (gamma storage ptr, rows, cols) -> (partial rows buffer, blocks, cols)
_LN_PENDING = []       # (partial rows buffer, blocks, cols, dgamma, dbeta)
_LN_PENDING_KEYS = set()


def _ln_defer_buffer(gamma, x):
    rows, cols = x.shape
    key = (gamma.data_ptr(), rows, cols)
    if key in _LN_PENDING_KEYS:            # the same LayerNorm twice in one pass (shared module): reduce this use right away
        return None
    part = _LN_PART.get(key)
    if part is None:
        if len(_LN_PART) > 256:
            _LN_PART.clear()
        blocks, nfloat = ops.layernorm_bwd_partial_shape(rows, cols)
        part = _LN_PART[key] = (torch.empty(nfloat, dtype=torch.float32, device=x.device), blocks, cols, key)
    return part


def _ln_defer(part, gamma, beta) -> None:
    buf, blocks, cols, key = part
    _LN_PENDING.append((buf, blocks, cols, gamma.grad, beta.grad))
    _LN_PENDING_KEYS.add(key)
    torch.autograd.Variable._execution_engine.queue_callback(flush_ln_param_grads)    # (later ones find nothing to do)


def flush_ln_param_grads() -> None:
    """Reduce every deferred LayerNorm parameter gradient of this backward pass (one launch per device)."""
    if not _LN_PENDING:
        return
    entries = list(_LN_PENDING)
    _LN_PENDING.clear()
    _LN_PENDING_KEYS.clear()
    ops.colreduce_entries(entries)


def discard_deferred_ln_param_grads() -> None:
    """Forget partial rows that were never reduced (a backward pass that raised): called by FusedAdamW.zero_grad()."""
    _LN_PENDING.clear()
    _LN_PENDING_KEYS.clear()


def layer_norm(x, gamma, beta, eps, act=None):
    lead = x.shape[:-1]
    if x.dtype != torch.float32:
        x = x.float()
    return LayerNormFn.apply(x.reshape(-1, x.shape[-1]), gamma, beta, eps, act).reshape(*lead, x.shape[-1])


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.act = act
        ctx.save_for_backward(x)
        return ops.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), x, ctx.act), None


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) (old/clip.py:63-64)."""

    @staticmethod
    def forward(ctx, x, eps):
        y, n = ops.l2norm_fwd(x.contiguous(), eps)
        ctx.eps = eps
        ctx.save_for_backward(y, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, n = ctx.saved_tensors
        return ops.l2norm_bwd(dy.contiguous(), y, n, ctx.eps), None


def l2_normalize(x, eps: float = 1e-12):
    return L2NormFn.apply(x, eps)


class SkipScaleFn(torch.autograd.Function):
    """skip + layer_scale * projected (old/clip_opt.py:41-44)."""

    @staticmethod
    def forward(ctx, skip, proj, scale):
        ctx.save_for_backward(proj, scale)
        return ops.axpby_dev(skip.contiguous(), proj.contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        proj, scale = ctx.saved_tensors
        dy = dy.contiguous()
        dproj = ops.axpby_dev(None, dy, scale)
        dscale = (dy * proj).sum().reshape(scale.shape)          # scalar reduction: plumbing
        return dy, dproj, dscale


class SimLogitsFn(torch.autograd.Function):
    """logits = scale * A B^T, materialised for the reference's module API (old/clip.py:66-67).
    Forward and backward are exact-f32 MFMA products (clipk_sim_logits / clipk_gemm_f32_nt on transposed operands
    prepared by clipk_transpose_scale_f32): dA = scale * dS B, dB = scale * dS^T A, dscale = <dA, A> / scale."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.contiguous(), b.contiguous()
        sc = scale.reshape(1).contiguous()
        ctx.save_for_backward(a, b, sc)
        ctx.scale_shape = scale.shape
        return ops.sim_logits(a, b, sc)

    @staticmethod
    def backward(ctx, ds):
        a, b, sc = ctx.saved_tensors
        ds = ds.contiguous()
        da = db = dscale = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            da = ops.matmul_f32_nt(ds, ops.transpose_scale_f32(b, sc))             # scale * dS @ B       [M, P]
        if ctx.needs_input_grad[1]:
            db = ops.matmul_f32_nt(ops.transpose_scale_f32(ds), ops.transpose_scale_f32(a, sc))   # scale dS^T A
        if ctx.needs_input_grad[2]:
            dscale = ((da * a).sum() / sc[0]).reshape(ctx.scale_shape)             # sum dS o (A B^T): scalar plumbing
        return da, db, dscale


class CEDiagFn(torch.autograd.Function):
    """w_row * CE(rows of [S | S_cache], diag) + w_col * CE(columns of S, diag) on MATERIALISED logits — the loss calls
    of old/ablation.py:16 (1, 0), rna_clip_codes.ipynb:1952-1953 (.5, .5) and old/clip_opt.py:130-151 (.5, .5 + cache)
    on the clipk_ce_logits kernels."""

    @staticmethod
    def forward(ctx, S, S2, w_row, w_col):
        S = S if S.stride(-1) == 1 else S.contiguous()
        if S2 is not None and (S2.shape[1] == 0):
            S2 = None
        if S2 is not None and S2.stride(-1) != 1:
            S2 = S2.contiguous()
        M, N = S.shape
        lse_r, pos_r = ops.ce_logits_lse(S, S2, columns=False)
        loss = w_row * (lse_r - pos_r).sum() / M
        lse_c = None
        if w_col != 0.0:
            lse_c, pos_c = ops.ce_logits_lse(S, None, columns=True)
            loss = loss + w_col * (lse_c - pos_c).sum() / N
        ctx.meta = (w_row / M, w_col / N)
        ctx.has_s2 = S2 is not None
        ctx.save_for_backward(S, S2, lse_r, lse_c)
        return loss

    @staticmethod
    def backward(ctx, g):
        S, S2, lse_r, lse_c = ctx.saved_tensors
        wr, wc = ctx.meta
        dS, dS2 = ops.ce_logits_bwd(S, S2, lse_r, lse_c, wr, wc, g.reshape(1).contiguous().float())
        return dS, dS2, None, None


def cross_entropy_diag(logits, cache_logits=None, symmetric=True):
    """F.cross_entropy(logits, arange) [+ the column direction] on the HIP kernels, for callers of the module API that
    hold materialised logits (square `logits`; `cache_logits` = extra negative columns of the row direction)."""
    if logits.shape[0] != logits.shape[1]:
        raise ValueError("cross_entropy_diag expects square logits [B, B]")
    w = (0.5, 0.5) if symmetric else (1.0, 0.0)
    return CEDiagFn.apply(logits.float(), None if cache_logits is None else cache_logits.float(), *w)


def sim_logits(a, b, scale):
    return SimLogitsFn.apply(a, b, scale)


class AttnFn(torch.autograd.Function):
    """Self-attention on a fused qkv tensor [B*L, 3*H*D] (bf16) -> [B*L, H*D] (bf16), flash kernels fwd/bwd."""

    @staticmethod
    def forward(ctx, qkv, B, L, H, D, key_mask, q_scale):
        qkv = qkv.contiguous()
        out, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=key_mask, rope=None, q_scale=q_scale)
        ctx.meta = (B, L, H, D, key_mask, q_scale)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, L, H, D, key_mask, q_scale = ctx.meta
        dqkv = ops.attn_bwd(qkv, out, _bf16(dout), lse, B, L, H, D, key_mask=key_mask, rope=None, q_scale=q_scale)
        return dqkv, None, None, None, None, None, None
