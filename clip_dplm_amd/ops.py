"""Thin tensor-level wrappers over the libclipk C ABI (no autograd here, no fallbacks).

Every function enqueues HIP kernels on torch's current stream and returns torch tensors that own the
output memory.  Inputs must live on a CUDA(HIP) device; anything else raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _ffi
from ._ffi import ACT, BF16, F32, U8, GemmArgs, check, ptr

_WS: dict = {}


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches on torch's current stream (the stream the
    kernels are enqueued on).  bench.py uses it for the live roofline numbers; off by default."""

    def __init__(self, only=None):
        self.records = {}          # name -> list of (start_event, end_event, algorithmic work)
        self.only = None if only is None else set(only)    # restrict to these kernel names (two events per launch cost
                                                           # ~1 % of a training step when every launch is timed)

    def add(self, name, s, e, work, nbytes=0.0, sub=None):
        self.records.setdefault(name, []).append((s, e, work, nbytes, sub))

    def summary_by_sub(self, name):
        """The launches of kernel `name` grouped by their sub-tag (gemm_nt: the epilogue mode) -> same fields as
        summary().  Call after torch.cuda.synchronize()."""
        groups = {}
        for r in self.records.get(name, []):
            groups.setdefault(r[4] or "-", []).append(r)
        out = {}
        for sub, recs in groups.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[sub] = {"launches": len(recs), "total_ms": ms, "avg_us": 1e3 * ms / max(len(recs), 1),
                        "work": float(sum(r[2] for r in recs)), "bytes": float(sum(r[3] for r in recs))}
        return out

    def summary(self):
        """name -> dict(launches, total_ms, avg_us, work) — call after torch.cuda.synchronize()."""
        out = {}
        for name, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[name] = {"launches": len(recs), "total_ms": ms, "avg_us": 1e3 * ms / max(len(recs), 1),
                         "work": float(sum(r[2] for r in recs)), "bytes": float(sum(r[3] for r in recs))}
        return out


_TIMER: Optional[KernelTimer] = None


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


def _timed(name: str, work: float, fn, nbytes: float = 0.0, sub=None):
    if _TIMER is None or (_TIMER.only is not None and name not in _TIMER.only):
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    _TIMER.add(name, s, e, work, nbytes, sub)
    return r


def _lib():
    return _ffi.load()


def set_option(name: str, value: int) -> None:
    """Kernel-selection option of libclipk (include/clipk.h: clipk_set_option).  Results never depend on options."""
    check(_lib().clipk_set_option(name.encode(), int(value)), f"clipk_set_option({name})")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(_lib().clipk_get_option(name.encode(), C.byref(v)), f"clipk_get_option({name})")
    return v.value


def reset_options() -> None:
    check(_lib().clipk_reset_options(), "clipk_reset_options")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts):
    """Every tensor on the GPU, and on the CURRENT one: the kernels are enqueued on the current device's stream, so a
    tensor of another device would be read through a foreign pointer.  Fails loudly instead."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _ffi.ClipkError("libclipk kernels need device tensors (there is no CPU fallback)")
        if dev is None:
            dev = t.device
            if dev.index is not None and dev.index != torch.cuda.current_device():
                raise _ffi.ClipkError(f"tensor on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                                      "wrap the call in `with torch.cuda.device(tensor.device):`")
        elif t.device != dev:
            raise _ffi.ClipkError(f"tensors on different devices: {dev} and {t.device}")


def workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """Grow-only per-(device, stream, tag) scratch buffer (kernels never allocate)."""
    key = (str(device), _stream(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# --------------------------------------------------------------------------------------------------
def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
            out_dtype=torch.bfloat16, residual: Optional[torch.Tensor] = None, out_preact: bool = False,
            dact_aux: Optional[torch.Tensor] = None, dact=None, alpha: float = 1.0,
            out: Optional[torch.Tensor] = None, dropout=None, rope=None, aux_u8: bool = False,
            rope_interleaved: bool = False):
    """C = epilogue(a[M,K] @ b[N,K]^T) with a, b bf16.  Returns C (and the bf16 pre-activation if asked).
    aux_u8 (act = "gelu", out_preact): the second output is GELU'(pre-activation) as 8-bit codes (uint8 [M, N]) instead of
    the bf16 pre-activation; a uint8 `dact_aux` (dact = "gelu") is read as such codes (clipk.h aux_dtype).
    dropout = (p, seed): nn.Dropout on the value after the activation (before act'(aux) and the residual add).
    rope = (cos, sin, L, hd, cols): rotate-half RoPE (tables f32 [L, hd/2], position = row mod L) on the first `cols`
    output columns in the epilogue (ESM-2's fused qkv projection: cols = 2 * hidden).  rope_interleaved: those columns'
    heads are in pair-interleaved order (b's rows permuted by cast_transpose(il=...)): partners are neighbours, any hd % 8 == 0."""
    _need_cuda(a, b, bias, residual, dact_aux, *(rope[:2] if rope else ()))
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1]
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    c = out if out is not None else torch.empty((M, N), dtype=out_dtype, device=a.device)
    aux8 = bool(aux_u8 and out_preact) or (dact_aux is not None and dact_aux.dtype == torch.uint8)
    if aux8:
        assert (act == "gelu" or not out_preact) and (dact == "gelu" or dact_aux is None), "8-bit aux: GELU only"
        assert dact_aux is None or dact_aux.dtype == torch.uint8
    pre = torch.empty((M, N), dtype=torch.uint8 if aux8 else torch.bfloat16, device=a.device) if out_preact else None
    args = GemmArgs()
    args.A, args.lda = a.data_ptr(), a.stride(0)
    args.B, args.ldb = b.data_ptr(), b.stride(0)
    args.C, args.ldc, args.c_dtype = c.data_ptr(), c.stride(0), _dt(c)
    args.M, args.N, args.K = M, N, K
    args.bias = ptr(bias)
    args.act = ACT[act]
    args.out_preact, args.ldp = ptr(pre), (pre.stride(0) if pre is not None else 0)
    args.dact_aux, args.ldd = ptr(dact_aux), (dact_aux.stride(0) if dact_aux is not None else 0)
    args.dact = ACT[dact]
    if residual is not None:
        args.residual, args.ldr, args.r_dtype = residual.data_ptr(), residual.stride(0), _dt(residual)
    else:
        args.residual, args.ldr, args.r_dtype = None, 0, 0
    args.alpha = alpha
    args.aux_dtype = U8 if aux8 else BF16
    args.drop_p, args.drop_seed = (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF) if dropout else (0.0, 0)
    if rope is not None:
        cos, sin, rl, hd, cols = rope
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape == (rl, hd // 2) and sin.shape == (rl, hd // 2)
        args.rope_cos, args.rope_sin = cos.data_ptr(), sin.data_ptr()
        args.rope_L, args.rope_hd, args.rope_cols, args.rope_row0 = int(rl), int(hd), int(cols), 0
        args.rope_interleaved = int(bool(rope_interleaved))
    else:
        args.rope_interleaved = 0
        args.rope_cos, args.rope_sin = None, None
        args.rope_L = args.rope_hd = args.rope_cols = args.rope_row0 = 0
    # algorithmic bytes: both operands once, every output / epilogue operand once
    nb = 2.0 * (M * K + N * K) + M * N * (c.element_size() + (pre.element_size() if out_preact else 0) +
                                          (dact_aux.element_size() if dact_aux is not None else 0)
                                          + (residual.element_size() if residual is not None else 0))
    sub = None
    if _TIMER is not None:                                   # epilogue mode of this launch (bench.py: per-mode rates)
        sub = ("dact_" + str(dact) + ("8" if aux8 else "") if dact_aux is not None else ("act_" + str(act) if act else "linear")) + \
              (("+dact8" if aux8 else "+preact") if out_preact else "") + ("+rope" if rope is not None else "") + \
              ("+res_" + ("f32" if residual.dtype == torch.float32 else "bf16") if residual is not None else "") + \
              ("+drop" if dropout else "") + ("->f32" if c.dtype == torch.float32 else "->bf16")
    check(_timed("gemm_nt", 2.0 * M * N * K, lambda: _lib().clipk_gemm_nt(C.byref(args), _stream()), nb, sub),
          "clipk_gemm_nt")
    return (c, pre) if out_preact else c


def gelu_aux_u8_applies(k_in: int, n_ffn: int) -> bool:
    """Shapes for which the 8-bit GELU' code can be the FFN pair's saved operand: the specialised epilogues need the
    LDS-DMA main loop (K % 32 == 0; the generic-K kernel returns CLIPK_ERR_UNSUPPORTED for CLIPK_U8, gemm_nt.hip) and
    code rows of whole 8-byte chunks (ldp / ldd % 8 == 0).  Both GEMMs of the pair - fc1 forward [M, n_ffn] and the
    fc2 input gradient [M, n_ffn] - contract over k_in = the model width.  Other widths (the notebook's 120) keep the
    bf16 pre-activation."""
    return k_in % 32 == 0 and n_ffn % 8 == 0


def gemm_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: Optional[torch.Tensor] = None,
               dbias: Optional[torch.Tensor] = None, accumulate: bool = False, want_bias: bool = False, il=(0, 0)):
    """dW[N,K] (+)= dy[M,N]^T @ x[M,K] (f32), optionally db[N] = colsum(dy).  il = (head columns, il_cols): dy's first
    il_cols columns are in pair-interleaved head order; dW / db come out in the original order (clipk.h)."""
    _need_cuda(dy, x)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
    M, N = dy.shape
    K = x.shape[1]
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        accumulate = False
    if want_bias and dbias is None:
        dbias = torch.empty((N,), dtype=torch.float32, device=dy.device)
    lib = _lib()
    nbytes = lib.clipk_gemm_wgrad_workspace(M, N, K)
    ws = workspace(nbytes, dy.device, "wgrad")
    check(_timed("gemm_wgrad", 2.0 * M * N * K,
                 lambda: lib.clipk_gemm_wgrad(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                              dw.stride(0), ptr(dbias), M, N, K, int(accumulate), int(il[0]), int(il[1]),
                                              ws.data_ptr(), ws.numel(), _stream())), "clipk_gemm_wgrad")
    return dw, dbias


# --------------------------------------------------------------------------------------------------
def simce_lse(x, y, scale, label_offset=0, cache=None):
    """lse[i] = logsumexp_j scale*<x_i, keys_j>, pos[i] = scale*<x_i, y_{label_offset+i}>."""
    _need_cuda(x, y, scale, cache)
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    Mx, P = x.shape
    Ny = y.shape[0]
    Nc = 0 if cache is None else cache.shape[0]
    lse = torch.empty(Mx, dtype=torch.float32, device=x.device)
    pos = torch.zeros(Mx, dtype=torch.float32, device=x.device)
    lib = _lib()
    nbytes = lib.clipk_simce_workspace(Mx, Ny + Nc, P)
    if nbytes == 0:
        raise _ffi.ClipkError(f"simce: unsupported shape Mx={Mx} Nkeys={Ny + Nc} P={P}")
    ws = workspace(nbytes, x.device, "simce")
    check(lib.clipk_simce_lse(x.data_ptr(), Mx, y.data_ptr(), Ny, ptr(cache), Nc, P, scale.data_ptr(), label_offset,
                              lse.data_ptr(), pos.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "clipk_simce_lse")
    return lse, pos


def ce_combine(lse_r, pos_r, lse_c, pos_c, w_row, w_col, bg):
    """(w_row * sum(lse_r - pos_r) + w_col * sum(lse_c - pos_c)) / bg as a 0-d device tensor, one launch
    (lse_c = pos_c = None: the one-sided loss)."""
    _need_cuda(lse_r, pos_r, lse_c, pos_c)
    out = torch.empty((), dtype=torch.float32, device=lse_r.device)
    check(_lib().clipk_ce_combine(lse_r.data_ptr(), pos_r.data_ptr(), ptr(lse_c), ptr(pos_c), lse_r.numel(), float(w_row),
                                  float(w_col), float(bg), out.data_ptr(), _stream()), "clipk_ce_combine")
    return out


def simce_grad(x, y, scale, lse_x, lse_y, w_row, w_col, inv_bg, label_offset=0, cache=None, upstream=None):
    """upstream: 1-element device tensor multiplied into the gradient inside the kernel (the loss' grad_output)."""
    _need_cuda(x, y, scale, lse_x, lse_y, cache, upstream)
    Mx, P = x.shape
    Ny = y.shape[0]
    Nc = 0 if cache is None else cache.shape[0]
    dx = torch.empty_like(x)
    dsc = torch.empty(Mx, dtype=torch.float32, device=x.device)
    lib = _lib()
    nbytes = lib.clipk_simce_workspace(Mx, Ny + Nc, P)
    ws = workspace(nbytes, x.device, "simce")
    check(lib.clipk_simce_grad_scaled(x.data_ptr(), Mx, y.data_ptr(), Ny, ptr(cache), Nc, P, scale.data_ptr(), label_offset,
                                      lse_x.data_ptr(), lse_y.data_ptr(), float(w_row), float(w_col), float(inv_bg),
                                      ptr(upstream), dx.data_ptr(), dsc.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "clipk_simce_grad_scaled")
    return dx, dsc


def _pairs_arrays(pairs):
    flat = [int(v) for pr in pairs for v in pr]
    rev = [pairs.index((b, a)) if (b, a) in pairs else i for i, (a, b) in enumerate(pairs)]
    return (C.c_int * len(flat))(*flat), (C.c_int * len(rev))(*rev)


def simce_lse_pairs(E, pairs, scale):
    """E f32 [nmod, B, P]; pairs = [(x_mod, y_mod), ...] (<= 6).  One launch: lse, pos each [npairs, B]."""
    _need_cuda(E, scale)
    assert E.dtype == torch.float32 and E.dim() == 3 and E.is_contiguous()
    nmod, B, P = E.shape
    n = len(pairs)
    pa, _ = _pairs_arrays(list(pairs))
    lse = torch.empty((n, B), dtype=torch.float32, device=E.device)
    pos = torch.zeros((n, B), dtype=torch.float32, device=E.device)
    lib = _lib()
    nbytes = lib.clipk_simce_pairs_workspace(n, B, P)
    if nbytes == 0:
        raise _ffi.ClipkError(f"simce pairs: unsupported shape npairs={n} B={B} P={P}")
    ws = workspace(nbytes, E.device, "simce")
    check(lib.clipk_simce_lse_pairs(E.data_ptr(), nmod, B, P, pa, n, scale.data_ptr(), lse.data_ptr(), pos.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _stream()), "clipk_simce_lse_pairs")
    return lse, pos


def simce_grad_pairs(E, pairs, scale, lse, w_row, w_col, inv_bg):
    """Gradients of every directed problem w.r.t. its X rows: dX [npairs, B, P], dscale partials [npairs, B]."""
    _need_cuda(E, scale, lse)
    nmod, B, P = E.shape
    n = len(pairs)
    pa, ra = _pairs_arrays(list(pairs))
    dX = torch.empty((n, B, P), dtype=torch.float32, device=E.device)
    dsc = torch.empty((n, B), dtype=torch.float32, device=E.device)
    lib = _lib()
    ws = workspace(lib.clipk_simce_pairs_workspace(n, B, P), E.device, "simce")
    check(lib.clipk_simce_grad_pairs(E.data_ptr(), nmod, B, P, pa, ra, n, scale.data_ptr(), lse.data_ptr(), float(w_row),
                                     float(w_col), float(inv_bg), dX.data_ptr(), dsc.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _stream()), "clipk_simce_grad_pairs")
    return dX, dsc


def sim_logits(x, y, scale):
    _need_cuda(x, y, scale)
    Mx, P = x.shape
    Ny = y.shape[0]
    s = torch.empty((Mx, Ny), dtype=torch.float32, device=x.device)
    check(_lib().clipk_sim_logits(x.data_ptr(), Mx, y.data_ptr(), Ny, P, scale.data_ptr(), s.data_ptr(), s.stride(0),
                                  _stream()), "clipk_sim_logits")
    return s


def ce_logits_lse(S, S2=None, columns=False, label_offset=0):
    """LSE over the rows (optionally of [S | S2]) or the columns of materialised f32 logits + the diagonal logit."""
    _need_cuda(S, S2)
    assert S.dtype == torch.float32 and S.dim() == 2 and S.stride(1) == 1
    M, N = S.shape
    N2 = 0 if S2 is None else S2.shape[1]
    if N2:
        assert S2.dtype == torch.float32 and S2.stride(1) == 1 and S2.shape[0] == M
    n = N if columns else M
    lse = torch.empty(n, dtype=torch.float32, device=S.device)
    pos = torch.empty(n, dtype=torch.float32, device=S.device)
    check(_lib().clipk_ce_logits_lse(S.data_ptr(), S.stride(0), M, N, ptr(S2) if N2 else None, S2.stride(0) if N2 else 0,
                                     N2, int(bool(columns)), label_offset, lse.data_ptr(), pos.data_ptr(), _stream()),
          "clipk_ce_logits_lse")
    return lse, pos


def ce_logits_bwd(S, S2, lse_row, lse_col, w_row, w_col, g, off_row=0, off_col=0):
    _need_cuda(S, S2, lse_row, lse_col, g)
    M, N = S.shape
    N2 = 0 if S2 is None else S2.shape[1]
    dS = torch.empty((M, N), dtype=torch.float32, device=S.device)
    dS2 = torch.empty((M, N2), dtype=torch.float32, device=S.device) if N2 else None
    check(_lib().clipk_ce_logits_bwd(S.data_ptr(), S.stride(0), M, N, ptr(S2) if N2 else None, S2.stride(0) if N2 else 0,
                                     N2, ptr(lse_row), ptr(lse_col), float(w_row), float(w_col), off_row, off_col,
                                     g.data_ptr(), dS.data_ptr(), dS.stride(0), ptr(dS2), dS2.stride(0) if N2 else 0,
                                     _stream()), "clipk_ce_logits_bwd")
    return dS, dS2


def transpose_scale_f32(x, scale=None):
    """scale[0] * x^T for a contiguous f32 [R, C] matrix (scale: device scalar tensor or None)."""
    _need_cuda(x, scale)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    R, Cc = x.shape
    out = torch.empty((Cc, R), dtype=torch.float32, device=x.device)
    check(_lib().clipk_transpose_scale_f32(x.data_ptr(), R, Cc, ptr(scale), out.data_ptr(), _stream()),
          "clipk_transpose_scale_f32")
    return out


_DROP_EPOCH = None


def set_dropout_epoch(epoch: Optional[torch.Tensor]) -> None:
    """Register (or, with None, clear) the device word every dropout site adds to its seed (clipk_set_dropout_epoch): int32 /
    uint32 tensor with one element, kept alive here while registered.  Used by training.GraphedTrainStep so that a replayed
    step draws new masks; eager code never needs it."""
    global _DROP_EPOCH
    if epoch is not None:
        _need_cuda(epoch)
        assert epoch.numel() == 1 and epoch.element_size() == 4 and epoch.dtype in (torch.int32, torch.uint32)
    check(_lib().clipk_set_dropout_epoch(0 if epoch is None else epoch.data_ptr()), "clipk_set_dropout_epoch")
    _DROP_EPOCH = epoch


def colsum_f32(x, out=None, accumulate=False):
    """out[c] (+)= sum_r x[r, c] (f32): bias gradients, written straight into `out` (e.g. a parameter's .grad)."""
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    if out is None:
        out, accumulate = torch.empty(x.shape[1], dtype=torch.float32, device=x.device), False
    check(_lib().clipk_colsum_f32(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), int(accumulate), _stream()),
          "clipk_colsum_f32")
    return out


def gemm_wgrad_f32(dy, x, dw=None, dbias=None, accumulate=False, want_bias=True):
    """Parameter gradients of an exact-f32 Linear: dW[N, K] (+)= dy[M, N]^T @ x[M, K], db[N] (+)= dy.sum(0), one launch when
    M <= 64.  dw / dbias: write (accumulate: add) into these tensors, e.g. the parameters' .grad; otherwise new tensors
    (want_bias=False and no dbias: no bias gradient).  Returns (dw, dbias)."""
    _need_cuda(dy, x, dw, dbias)
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.dim() == 2 and x.dim() == 2
    assert dy.shape[0] == x.shape[0]
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    if x.stride(1) != 1:
        x = x.contiguous()
    M, N = dy.shape
    K = x.shape[1]
    fresh = dw is None
    if fresh:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    if dbias is None and want_bias:
        assert fresh or not accumulate, "accumulate with a new dbias tensor"
        dbias = torch.empty(N, dtype=torch.float32, device=dy.device)
    assert dw.dtype == torch.float32 and dw.shape == (N, K) and dw.stride(1) == 1
    assert dbias is None or (dbias.dtype == torch.float32 and dbias.numel() == N and dbias.is_contiguous())
    acc = bool(accumulate and not fresh)
    if M > 64:                 # many rows: the tiled kernel (its wrapper pads odd leading dimensions) + the column reduce
        gemm_f32(dy, x, trans_a=True, trans_b=True, addend=dw if acc else None, out=dw)
        if dbias is not None:
            colsum_f32(dy.contiguous(), out=dbias, accumulate=acc)
        return dw, dbias
    lib = _lib()
    check(_timed("gemm_f32", 2.0 * M * N * K,
                 lambda: lib.clipk_gemm_wgrad_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                                  dw.stride(0), ptr(dbias), M, N, K, int(acc), _stream()),
                 4.0 * (M * (N + K) + N * K * (2 if acc else 1))), "clipk_gemm_wgrad_f32")
    return dw, dbias


def gemm_f32(a, b, trans_a=False, trans_b=False, bias=None, addend=None, addend_scale=None, alpha=None, out=None):
    """Exact-f32 out[M, N] = alpha * opA(a) @ opB(b) (+ bias) (+ addend_scale * addend) on the tiled f32-MFMA kernel
    (alpha / addend_scale: 1-element device tensors).
    trans_a: a is stored [K, M];  trans_b: b is stored [K, N] (else [N, K], the nn.Linear layout).
    out: write into this [M, N] f32 tensor (row stride = its stride(0)); out may BE the addend (in-place accumulation:
    every element is read and written by the same thread)."""
    _need_cuda(a, b, bias, addend, addend_scale, alpha)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
    if a.stride(1) != 1 or a.stride(0) % 4:
        a = a.contiguous()
    if b.stride(1) != 1 or b.stride(0) % 4:
        b = b.contiguous()
    K, M = (a.shape if trans_a else (a.shape[1], a.shape[0]))
    Kb, N = (b.shape if trans_b else (b.shape[1], b.shape[0]))
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    # float4 staging wants 16-byte aligned rows: zero-pad the stored rows of an operand with an odd leading dimension
    # (plumbing; M, N, K stay what they are, the kernel never reads the pad as data)
    if a.stride(0) % 4:
        a = torch.nn.functional.pad(a, (0, (-a.shape[1]) % 4))
    if b.stride(0) % 4:
        b = torch.nn.functional.pad(b, (0, (-b.shape[1]) % 4))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        assert out.shape == (M, N) and out.dtype == torch.float32 and out.stride(1) == 1 and out.device == a.device
    if addend is not None:
        assert addend.shape == (M, N) and addend.stride(1) == 1
    nb = 4.0 * (M * K + N * K + M * N * (2 if addend is not None else 1))     # operands once, output (+ addend) once
    lib = _lib()
    wsb = lib.clipk_gemm_f32_workspace(M, N, K, int(trans_a), int(trans_b))
    ws = workspace(wsb, a.device, "gemm_f32") if wsb else None
    check(_timed("gemm_f32", 2.0 * M * N * K,
                 lambda: lib.clipk_gemm_f32(a.data_ptr(), a.stride(0), int(trans_a), b.data_ptr(), b.stride(0),
                                            int(trans_b), M, N, K, ptr(alpha), ptr(bias), ptr(addend),
                                            addend.stride(0) if addend is not None else 0, ptr(addend_scale),
                                            out.data_ptr(), out.stride(0), ptr(ws), ws.numel() if ws is not None else 0,
                                            _stream()), nb), "clipk_gemm_f32")
    return out


def matmul_f32_nt(a, b):
    """a[M, K] @ b[N, K]^T in exact f32, any K."""
    return gemm_f32(a, b)


def gemm_f32_nt(x, w, bias=None, addend=None, addend_scale=None):
    """Exact-f32 out = x @ w.T (+ bias) (+ addend_scale * addend): the ICNN's Linear layers."""
    return gemm_f32(x, w, bias=bias, addend=addend, addend_scale=addend_scale)


# --------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, act=None, want_f32=True, want_bf16=False, want_stats=True):
    _need_cuda(x, gamma, beta)
    rows, cols = x.shape
    y32 = torch.empty((rows, cols), dtype=torch.float32, device=x.device) if want_f32 else None
    y16 = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    nbytes = rows * cols * (x.element_size() + (4 if want_f32 else 0) + (2 if want_bf16 else 0))
    check(_timed("layernorm_fwd", nbytes,
                 lambda: _lib().clipk_layernorm_fwd(x.data_ptr(), _dt(x), x.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                                    float(eps), ACT[act], ptr(y32), ptr(y16), cols, ptr(mean),
                                                    ptr(rstd), rows, cols, _stream())), "clipk_layernorm_fwd")
    return y32, y16, mean, rstd


def layernorm_bwd_partial_shape(rows: int, cols: int):
    """(blocks, floats) of the per-block dgamma / dbeta partial rows clipk_layernorm_bwd leaves in its workspace."""
    nbytes = _lib().clipk_layernorm_bwd_workspace(rows, cols)
    return nbytes // (2 * cols * 4), nbytes // 4


def colreduce_batched(desc: torch.Tensor, n: int, max_cols: int) -> None:
    """Reduce the partial rows of n LayerNorm backwards in one launch (clipk_colreduce_batched; desc int64 [n, 6] on the
    device: partial rows, blocks, cols, dgamma, dbeta, accumulate)."""
    _need_cuda(desc)
    assert desc.dtype == torch.int64 and desc.is_contiguous() and desc.numel() >= 6 * n
    check(_lib().clipk_colreduce_batched(desc.data_ptr(), int(n), int(max_cols), _stream()), "clipk_colreduce_batched")


_COLRED_DESC = {}


def colreduce_entries(entries) -> None:
    """entries: (partial rows tensor, blocks, cols, dgamma, dbeta) per LayerNorm; dgamma / dbeta += column sums, one launch per
    device.  The descriptor table lives on the device and is cached by the pointers it holds (a captured step finds the table
    its warm-up built)."""
    by_dev = {}
    for e in entries:
        by_dev.setdefault(e[0].device, []).append(e)
    for dev, es in by_dev.items():
        rows = tuple((p.data_ptr(), int(b), int(c), g.data_ptr(), bt.data_ptr(), 1) for p, b, c, g, bt in es)
        desc = _COLRED_DESC.get((dev, rows))
        if desc is None:
            if len(_COLRED_DESC) > 32:
                _COLRED_DESC.clear()
            desc = _COLRED_DESC[(dev, rows)] = torch.tensor(rows, dtype=torch.int64).to(dev)
        with torch.cuda.device(dev):
            colreduce_batched(desc, len(es), max(int(c) for _, _, c, _, _ in es))


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, act=None, dx_add=None, want_f32=True, want_bf16=False,
                  dgamma=None, dbeta=None, accumulate=False, want_param_grads=True, dropout_bf16=None, part_out=None):
    """part_out (f32, >= layernorm_bwd_partial_shape(rows, cols)[1] elements): the kernel's workspace is THIS buffer and no
    parameter gradient is reduced - the caller reduces the partial rows left in it later (colreduce_batched)."""
    _need_cuda(dy, x, gamma, mean, rstd, part_out)
    rows, cols = x.shape
    dx32 = torch.empty((rows, cols), dtype=torch.float32, device=x.device) if want_f32 else None
    dx16 = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if part_out is not None:
        want_param_grads = False
    if not want_param_grads:                               # input gradient only (transport maps in eval mode)
        dgamma = dbeta = None
    elif dgamma is None:
        dgamma = torch.empty(cols, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
    lib = _lib()
    nbytes = lib.clipk_layernorm_bwd_workspace(rows, cols)
    if part_out is not None:
        assert part_out.dtype == torch.float32 and part_out.is_contiguous() and part_out.numel() * 4 >= nbytes
        ws = part_out.view(torch.uint8) if part_out.dim() == 1 else part_out.reshape(-1).view(torch.uint8)
    else:
        ws = workspace(nbytes, x.device, "ln")
    nb = rows * cols * (dy.element_size() + x.element_size() + (dx_add.element_size() if dx_add is not None else 0) +
                        (4 if want_f32 else 0) + (2 if want_bf16 else 0))
    check(_timed("layernorm_bwd", nb,
                 lambda: lib.clipk_layernorm_bwd(dy.data_ptr(), _dt(dy), dy.stride(0), x.data_ptr(), _dt(x), x.stride(0),
                                                 gamma.data_ptr(), ptr(beta), mean.data_ptr(), rstd.data_ptr(), ACT[act],
                                                 ptr(dx_add), _dt(dx_add) if dx_add is not None else F32, ptr(dx32),
                                                 ptr(dx16), cols, ptr(dgamma),
                                                 ptr(dbeta), int(accumulate), rows, cols,
                                                 float(dropout_bf16[0]) if dropout_bf16 else 0.0,
                                                 (int(dropout_bf16[1]) & 0xFFFFFFFF) if dropout_bf16 else 0, ws.data_ptr(),
                                                 ws.numel(), _stream())), "clipk_layernorm_bwd")
    return dx32, dx16, dgamma, dbeta


def layernorm_bwd2(g, dy, a, gamma, beta, mean, rstd, act=None, dgamma=None, dbeta=None, accumulate=False):
    """Backward of layernorm_bwd (second order, f32; act in {None, "celu", "softplus"}): cotangent g of da ->
    (d_dy, d_a, d_gamma, d_beta).  See clipk_layernorm_bwd2."""
    _need_cuda(g, dy, a, gamma, mean, rstd)
    rows, cols = a.shape
    g, dy, a = g.contiguous(), dy.contiguous(), a.contiguous()
    assert g.dtype == dy.dtype == a.dtype == torch.float32
    d_dy, d_a = torch.empty_like(a), torch.empty_like(a)
    if dgamma is None:
        dgamma = torch.empty(cols, dtype=torch.float32, device=a.device)
        dbeta = torch.empty(cols, dtype=torch.float32, device=a.device)
        accumulate = False
    lib = _lib()
    nbytes = lib.clipk_layernorm_bwd_workspace(rows, cols)
    ws = workspace(nbytes, a.device, "ln")
    check(_timed("layernorm_bwd2", 5.0 * rows * cols * 4,
                 lambda: lib.clipk_layernorm_bwd2(g.data_ptr(), dy.data_ptr(), a.data_ptr(), a.stride(0), gamma.data_ptr(),
                                                  ptr(beta), mean.data_ptr(), rstd.data_ptr(), ACT[act], d_dy.data_ptr(),
                                                  d_a.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate),
                                                  rows, cols, ws.data_ptr(), nbytes, _stream())), "clipk_layernorm_bwd2")
    return d_dy, d_a, dgamma, dbeta


def meanpool_fused_supported(cols: int) -> bool:
    """Row widths clipk_layernorm_meanpool_fwd takes (its four partial rows must fit 64 KiB of LDS)."""
    return cols % 4 == 0 and (4 * cols + 4) * 4 <= 65536


def layernorm_meanpool_fwd(x, gamma, beta, eps, B, L, mask=None):
    """pooled[b] = mean over the valid rows l of LayerNorm(x[b * L + l]) in one pass (the normalised rows are never
    written).  x f32 / bf16 [B*L, cols]; mask u8 [B*L] (1 = valid) or None.  Returns pooled [B, cols], mean, rstd [B*L] and
    row_weight [B*L] (each row's weight in its sample's mean: what the backward needs)."""
    _need_cuda(x, gamma, beta, mask)
    rows, cols = x.shape
    assert rows == B * L and x.dtype in (torch.float32, torch.bfloat16) and x.stride(1) == 1
    pooled = torch.empty((B, cols), dtype=torch.float32, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    wrow = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_timed("layernorm_fwd", rows * cols * x.element_size(),
                 lambda: _lib().clipk_layernorm_meanpool_fwd(x.data_ptr(), _dt(x), x.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                                             float(eps), ptr(mask), B, L, cols, pooled.data_ptr(),
                                                             mean.data_ptr(), rstd.data_ptr(), wrow.data_ptr(),
                                                             _stream())), "clipk_layernorm_meanpool_fwd")
    return pooled, mean, rstd, wrow


def layernorm_meanpool_bwd(dpooled, row_weight, x, gamma, mean, rstd, B, L, want_f32=True, want_bf16=False,
                           dgamma=None, dbeta=None, accumulate=False):
    """Backward of layernorm_meanpool_fwd: dx f32 and / or bf16 [B*L, cols], dgamma, dbeta."""
    _need_cuda(dpooled, row_weight, x, gamma, mean, rstd)
    rows, cols = x.shape
    assert rows == B * L and dpooled.shape == (B, cols) and dpooled.dtype == torch.float32 and dpooled.is_contiguous()
    dx32 = torch.empty((rows, cols), dtype=torch.float32, device=x.device) if want_f32 else None
    dx16 = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if dgamma is None:
        dgamma = torch.empty(cols, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
    lib = _lib()
    ws = workspace(lib.clipk_layernorm_bwd_workspace(rows, cols), x.device, "ln")
    nb = rows * cols * (x.element_size() + (4 if want_f32 else 0) + (2 if want_bf16 else 0))
    check(_timed("layernorm_bwd", nb,
                 lambda: lib.clipk_layernorm_meanpool_bwd(dpooled.data_ptr(), row_weight.data_ptr(), B, L,
                                                          x.data_ptr(), _dt(x), x.stride(0), gamma.data_ptr(), mean.data_ptr(),
                                                          rstd.data_ptr(), ptr(dx32), ptr(dx16), cols, ptr(dgamma),
                                                          ptr(dbeta), int(accumulate), cols, ws.data_ptr(), ws.numel(),
                                                          _stream())), "clipk_layernorm_meanpool_bwd")
    return dx32, dx16, dgamma, dbeta


def l2norm_fwd(x, eps=1e-12):
    _need_cuda(x)
    rows, cols = x.shape
    y = torch.empty_like(x)
    n = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib().clipk_l2norm_fwd(x.data_ptr(), y.data_ptr(), n.data_ptr(), rows, cols, eps, _stream()), "clipk_l2norm_fwd")
    return y, n


def l2norm_bwd(dy, y, n, eps=1e-12):
    rows, cols = y.shape
    dx = torch.empty_like(y)
    check(_lib().clipk_l2norm_bwd(dy.data_ptr(), y.data_ptr(), n.data_ptr(), dx.data_ptr(), rows, cols, eps, _stream()),
          "clipk_l2norm_bwd")
    return dx


# --------------------------------------------------------------------------------------------------
def to_bf16(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib().clipk_cast_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "clipk_cast_f32_to_bf16")
    return y


def to_f32(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib().clipk_cast_bf16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "clipk_cast_bf16_to_f32")
    return y


def il_source_rows(n_rows: int, hd: int, il_rows: int) -> torch.Tensor:
    """int64 [n_rows]: the ORIGINAL row each row of a pair-interleaved copy holds (csrc/common.h il_src): within every
    head of `hd` rows of the first `il_rows` rows, copy row 2 j / 2 j + 1 = original row j / j + hd / 2."""
    r = torch.arange(n_rows, dtype=torch.int64)
    d = r % hd
    src = (r - d) + (d // 2) + (d % 2) * (hd // 2)
    return torch.where(r < il_rows, src, r)


def cast_transpose(w: torch.Tensor, want_w=True, want_wt=True, w_out=None, wt_out=None, il=(0, 0)):
    """il = (head rows, il_rows): the first il_rows rows of both copies in pair-interleaved head order (clipk.h)."""
    _need_cuda(w)
    rows, cols = w.shape
    wb = w_out if w_out is not None else (torch.empty((rows, cols), dtype=torch.bfloat16, device=w.device) if want_w else None)
    wt = wt_out if wt_out is not None else (torch.empty((cols, rows), dtype=torch.bfloat16, device=w.device) if want_wt else None)
    check(_lib().clipk_cast_transpose(w.data_ptr(), ptr(wb), ptr(wt), rows, cols, int(il[0]), int(il[1]), _stream()),
          "clipk_cast_transpose")
    return wb, wt


def cast_transpose_batched(desc: torch.Tensor):
    """desc: int64 [n, 7] device tensor of (w_ptr, wb_ptr, wt_ptr, rows, cols, il_hd, il_rows): all n weights in one launch."""
    _need_cuda(desc)
    assert desc.dtype == torch.int64 and desc.dim() == 2 and desc.shape[1] == 7 and desc.is_contiguous()
    check(_lib().clipk_cast_transpose_batched(desc.data_ptr(), desc.shape[0], _stream()), "clipk_cast_transpose_batched")


def act_fwd(x, act):
    y = torch.empty_like(x)
    check(_lib().clipk_act_fwd(x.data_ptr(), y.data_ptr(), ACT[act], x.numel(), _stream()), "clipk_act_fwd")
    return y


def act_bwd(dy, x, act):
    dx = torch.empty_like(x)
    check(_lib().clipk_act_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), ACT[act], x.numel(), _stream()), "clipk_act_bwd")
    return dx


def dact(dy, aux_bf16, act):
    """bf16( dy * act'(aux) )"""
    out = torch.empty(aux_bf16.shape, dtype=torch.bfloat16, device=aux_bf16.device)
    check(_lib().clipk_dact(dy.data_ptr(), _dt(dy), aux_bf16.data_ptr(), ACT[act], out.data_ptr(), aux_bf16.numel(),
                            _stream()), "clipk_dact")
    return out


def axpby_dev(a, b, s):
    """a + s * b with s a 1-element device tensor (a = None: s * b)."""
    y = torch.empty_like(b)
    check(_lib().clipk_axpby_dev(ptr(a), b.data_ptr(), s.data_ptr(), y.data_ptr(), b.numel(), _stream()),
          "clipk_axpby_dev")
    return y


# --------------------------------------------------------------------------------------------------
def _drop(dropout):
    return (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF) if dropout else (0.0, 0)


def attn_fwd(qkv, B, L, H, D, key_mask=None, rope=None, q_scale=1.0, dropout=None):
    _need_cuda(qkv, key_mask)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (B * L, 3 * H * D)
    out = torch.empty((B * L, H * D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    cos, sin = rope if rope is not None else (None, None)
    check(_timed("attn_fwd", 4.0 * B * H * L * L * D,
                 lambda: _lib().clipk_attn_fwd(qkv.data_ptr(), ptr(key_mask), ptr(cos), ptr(sin), out.data_ptr(),
                                               lse.data_ptr(), B, L, H, D, float(q_scale), *_drop(dropout),
                                               _stream())), "clipk_attn_fwd")
    return out, lse


def rope_qk_(qkv, B, L, H, D, rope):
    """In-place RoPE of the q and k sections of qkv (then: attn_fwd(rope=None), attn_bwd(rope=..., prerotated=True))."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (B * L, 3 * H * D)
    cos, sin = rope
    check(_lib().clipk_rope_qk(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, L, H, D, _stream()), "clipk_rope_qk")
    return qkv


def attn_fwd_rot_(qkv, B, L, H, D, rope, key_mask=None, q_scale=1.0):
    """rope_qk_ + attn_fwd(rope=None) in one call (one kernel for short heads): rotates q / k of `qkv` in place and
    returns (out, lse) computed from the rotated values; backward: attn_bwd(rope=..., prerotated=True)."""
    _need_cuda(qkv, key_mask)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (B * L, 3 * H * D)
    out = torch.empty((B * L, H * D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    cos, sin = rope
    check(_timed("attn_fwd", 4.0 * B * H * L * L * D,
                 lambda: _lib().clipk_attn_fwd_rot(qkv.data_ptr(), ptr(key_mask), cos.data_ptr(), sin.data_ptr(),
                                                   out.data_ptr(), lse.data_ptr(), B, L, H, D, float(q_scale),
                                                   _stream())), "clipk_attn_fwd_rot")
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=None, rope=None, q_scale=1.0, prerotated=False, dropout=None):
    _need_cuda(qkv, out, dout, lse)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    cos, sin = rope if rope is not None else (None, None)
    check(_timed("attn_bwd", 10.0 * B * H * L * L * D,
                 lambda: _lib().clipk_attn_bwd(qkv.data_ptr(), ptr(key_mask), ptr(cos), ptr(sin), out.data_ptr(),
                                               dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, L,
                                               H, D, float(q_scale), int(prerotated), *_drop(dropout),
                                               _stream())), "clipk_attn_bwd")
    return dqkv


def attn_f32_fwd(qkv, B, L, H, D, key_mask=None, q_scale=1.0, dropout=None):
    """Exact-f32 self-attention (clipk_attn_f32_fwd): qkv f32 [B*L, 3*H*D] -> out f32 [B*L, H*D], lse f32 [B, H, L]."""
    _need_cuda(qkv, key_mask)
    assert qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv.shape == (B * L, 3 * H * D)
    out = torch.empty((B * L, H * D), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    check(_timed("attn_f32_fwd", 4.0 * B * H * L * L * D,
                 lambda: _lib().clipk_attn_f32_fwd(qkv.data_ptr(), ptr(key_mask), out.data_ptr(), lse.data_ptr(), B, L, H,
                                                   D, float(q_scale), *_drop(dropout), _stream())), "clipk_attn_f32_fwd")
    return out, lse


def attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=None, q_scale=1.0, dropout=None):
    _need_cuda(qkv, out, dout, lse, key_mask)
    assert qkv.dtype == torch.float32 and dout.dtype == torch.float32 and dout.is_contiguous() and out.is_contiguous()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    check(_timed("attn_f32_bwd", 14.0 * B * H * L * L * D,
                 lambda: _lib().clipk_attn_f32_bwd(qkv.data_ptr(), ptr(key_mask), out.data_ptr(), dout.data_ptr(),
                                                   lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, L, H, D,
                                                   float(q_scale), *_drop(dropout), _stream())), "clipk_attn_f32_bwd")
    return dqkv


def dropout_f32(x, dropout, addend=None):
    """x * keep / (1 - p) (+ addend) with the kernels' counter-based mask over x's row-major element index."""
    _need_cuda(x, addend)
    assert x.dtype == torch.float32 and x.is_contiguous() and (addend is None or (addend.is_contiguous() and
                                                                                  addend.shape == x.shape))
    y = torch.empty_like(x)
    p_, seed = _drop(dropout)
    check(_lib().clipk_dropout_f32(x.data_ptr(), ptr(addend), y.data_ptr(), x.numel(), p_, seed, _stream()),
          "clipk_dropout_f32")
    return y


def attn_varlen_fwd(qkv, cu_seqlens, max_len, H, D, rope=None, q_scale=1.0, dropout=None):
    """Packed variable-length self-attention: qkv bf16 [T, 3*H*D], cu_seqlens int32 [B+1] on the device."""
    _need_cuda(qkv, cu_seqlens)
    T = qkv.shape[0]
    B = cu_seqlens.numel() - 1
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (T, 3 * H * D)
    assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous()
    out = torch.empty((T, H * D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    cos, sin = rope if rope is not None else (None, None)
    check(_timed("attn_fwd", 0.0,
                 lambda: _lib().clipk_attn_varlen_fwd(qkv.data_ptr(), cu_seqlens.data_ptr(), ptr(cos), ptr(sin),
                                                      out.data_ptr(), lse.data_ptr(), B, T, int(max_len), H, D,
                                                      float(q_scale), *_drop(dropout), _stream())),
          "clipk_attn_varlen_fwd")
    return out, lse


def attn_varlen_fwd_rot_(qkv, cu_seqlens, max_len, H, D, rope, q_scale=1.0):
    """attn_fwd_rot_ for a packed batch (whole-head kernel, D in {16, 24, 32}, 128 < max_len <= 256): q / k of `qkv`
    are rotated in place; backward: attn_varlen_bwd(..., prerotated=True)."""
    _need_cuda(qkv, cu_seqlens)
    T = qkv.shape[0]
    B = cu_seqlens.numel() - 1
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (T, 3 * H * D)
    assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous()
    out = torch.empty((T, H * D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    cos, sin = rope
    check(_timed("attn_fwd", 0.0,
                 lambda: _lib().clipk_attn_varlen_fwd_rot(qkv.data_ptr(), cu_seqlens.data_ptr(), cos.data_ptr(),
                                                          sin.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T,
                                                          int(max_len), H, D, float(q_scale), _stream())),
          "clipk_attn_varlen_fwd_rot")
    return out, lse


def varlen_whole_head_applies(max_len: int, D: int) -> bool:
    """Shapes clipk_attn_varlen_fwd_rot accepts (the whole-head short-sequence kernels)."""
    return D in (16, 24, 32) and 128 < int(max_len) <= 256


def attn_varlen_bwd(qkv, out, dout, lse, cu_seqlens, max_len, H, D, rope=None, q_scale=1.0, dropout=None,
                    prerotated=False):
    _need_cuda(qkv, out, dout, lse, cu_seqlens)
    T = qkv.shape[0]
    B = cu_seqlens.numel() - 1
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    cos, sin = rope if rope is not None else (None, None)
    check(_timed("attn_bwd", 0.0,
                 lambda: _lib().clipk_attn_varlen_bwd(qkv.data_ptr(), cu_seqlens.data_ptr(), ptr(cos), ptr(sin),
                                                      out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                                      dqkv.data_ptr(), B, T, int(max_len), H, D, float(q_scale),
                                                      int(bool(prerotated)), *_drop(dropout), _stream())),
          "clipk_attn_varlen_bwd")
    return dqkv


# --------------------------------------------------------------------------------------------------
def embed_fwd(ids, table, row_scale=None, mask=None, mask_token_id=-1):
    _need_cuda(ids, table)
    B, L = ids.shape
    d = table.shape[1]
    x = torch.empty((B * L, d), dtype=torch.float32, device=table.device)
    check(_lib().clipk_embed_fwd(ids.data_ptr(), table.data_ptr(), ptr(row_scale), ptr(mask), mask_token_id, x.data_ptr(),
                                 B, L, d, table.shape[0], _stream()), "clipk_embed_fwd")
    return x


def embed_bwd(ids, dx, dtable, row_scale=None, mask=None, mask_token_id=-1):
    B, L = ids.shape
    V, d = dtable.shape
    lib = _lib()
    nbytes = lib.clipk_embed_bwd_workspace(B, L, d, V)
    ws = workspace(nbytes, dx.device, "embed") if nbytes else None
    check(lib.clipk_embed_bwd(ids.data_ptr(), dx.data_ptr(), ptr(row_scale), ptr(mask), mask_token_id,
                              dtable.data_ptr(), B, L, d, V, ptr(ws), ws.numel() if ws is not None else 0, _stream()),
          "clipk_embed_bwd")
    return dtable


def pool_fwd(x, B, L, mask=None, mode=1):
    d = x.shape[-1]
    y = torch.empty((B, d), dtype=torch.float32, device=x.device)
    check(_lib().clipk_pool_fwd(x.data_ptr(), ptr(mask), y.data_ptr(), B, L, d, mode, _stream()), "clipk_pool_fwd")
    return y


def pool_bwd(dy, B, L, mask=None, mode=1):
    d = dy.shape[-1]
    dx = torch.empty((B * L, d), dtype=torch.float32, device=dy.device)
    check(_lib().clipk_pool_bwd(dy.data_ptr(), ptr(mask), dx.data_ptr(), B, L, d, mode, _stream()), "clipk_pool_bwd")
    return dx


def pool_varlen_fwd(x, cu_seqlens, mode=1):
    _need_cuda(x, cu_seqlens)
    B, d = cu_seqlens.numel() - 1, x.shape[-1]
    y = torch.empty((B, d), dtype=torch.float32, device=x.device)
    check(_lib().clipk_pool_varlen_fwd(x.data_ptr(), cu_seqlens.data_ptr(), y.data_ptr(), B, d, mode, _stream()),
          "clipk_pool_varlen_fwd")
    return y


def pool_varlen_bwd(dy, cu_seqlens, T, mode=1):
    _need_cuda(dy, cu_seqlens)
    B, d = dy.shape
    dx = torch.empty((T, d), dtype=torch.float32, device=dy.device)
    check(_lib().clipk_pool_varlen_bwd(dy.data_ptr(), cu_seqlens.data_ptr(), dx.data_ptr(), B, d, mode, _stream()),
          "clipk_pool_varlen_bwd")
    return dx


# --------------------------------------------------------------------------------------------------
def sumsq(g: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(g)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=g.device)
    lib = _lib()
    ws = workspace(lib.clipk_sumsq_workspace(g.numel()), g.device, "sumsq")
    check(lib.clipk_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "clipk_sumsq")
    return out


def adamw_step(w, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_norm=0.0,
               grad_scale=1.0, w_bf16=None, hyper=None):
    """hyper: device f32 [3] = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read by the kernel instead of lr / step (graph replay)."""
    _need_cuda(w, g, m, v, hyper)
    check(_lib().clipk_adamw_step(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), ptr(w_bf16), w.numel(),
                                  float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                  ptr(grad_norm_sq), float(max_norm), float(grad_scale), ptr(hyper), _stream()),
          "clipk_adamw_step")
