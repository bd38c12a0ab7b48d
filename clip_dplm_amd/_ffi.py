"""ctypes binding of libclipk.so (the C ABI declared in include/clipk.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel returns a
non-zero status this module raises.  PyTorch is used by the callers only for device memory, streams
and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libclipk.so")

BF16, F32, U8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_CELU, ACT_SOFTPLUS = 0, 1, 2, 3, 4
ACT = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU, "celu": ACT_CELU, "softplus": ACT_SOFTPLUS}


class ClipkError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_dtype", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("out_preact", C.c_void_p), ("ldp", C.c_int64),
        ("dact_aux", C.c_void_p), ("ldd", C.c_int64),
        ("dact", C.c_int),
        ("residual", C.c_void_p), ("ldr", C.c_int64), ("r_dtype", C.c_int),
        ("alpha", C.c_float),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("rope_L", C.c_int), ("rope_hd", C.c_int), ("rope_cols", C.c_int), ("rope_row0", C.c_int),
        ("aux_dtype", C.c_int),
        ("rope_interleaved", C.c_int),
    ]


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/clipk.h declares
SIGNATURES = {
    "clipk_version": (_i, []),
    "clipk_arch": (C.c_char_p, []),
    "clipk_status_string": (C.c_char_p, [_i]),
    "clipk_set_option": (_i, [C.c_char_p, _i]),
    "clipk_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "clipk_reset_options": (_i, []),
    "clipk_gemm_nt": (_i, [C.POINTER(GemmArgs), _vp]),
    "clipk_gemm_wgrad_workspace": (_sz, [_i, _i, _i]),
    "clipk_gemm_wgrad": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "clipk_simce_workspace": (_sz, [_i, _i, _i]),
    "clipk_simce_lse": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "clipk_simce_grad": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "clipk_simce_grad_scaled": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "clipk_ce_combine": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp]),
    "clipk_simce_pairs_workspace": (_sz, [_i, _i, _i]),
    "clipk_simce_lse_pairs": (_i, [_vp, _i, _i, _i, C.POINTER(_i), _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "clipk_simce_grad_pairs": (_i, [_vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _i, _vp, _vp, _f, _f, _f, _vp, _vp,
                                    _vp, _sz, _vp]),
    "clipk_sim_logits": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i64, _vp]),
    "clipk_ce_logits_lse": (_i, [_vp, _i64, _i, _i, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "clipk_ce_logits_bwd": (_i, [_vp, _i64, _i, _i, _vp, _i64, _i, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _i64, _vp, _i64, _vp]),
    "clipk_transpose_scale_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "clipk_gemm_f32_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "clipk_gemm_f32": (_i, [_vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _sz, _vp]),
    "clipk_gemm_f32_nt": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "clipk_layernorm_fwd": (_i, [_vp, _i, _i64, _vp, _vp, _f, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _vp]),
    "clipk_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "clipk_layernorm_bwd": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64,
                                 _vp, _vp, _i, _i, _i, _f, C.c_uint32, _vp, _sz, _vp]),
    "clipk_layernorm_meanpool_fwd": (_i, [_vp, _i, _i64, _vp, _vp, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "clipk_layernorm_meanpool_bwd": (_i, [_vp, _vp, _i, _i, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                          _i, _i, _vp, _sz, _vp]),
    "clipk_l2norm_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "clipk_l2norm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "clipk_cast_f32_to_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "clipk_cast_bf16_to_f32": (_i, [_vp, _vp, _i64, _vp]),
    "clipk_cast_transpose": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "clipk_cast_transpose_batched": (_i, [_vp, _i, _vp]),
    "clipk_act_fwd": (_i, [_vp, _vp, _i, _i64, _vp]),
    "clipk_act_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    "clipk_dact": (_i, [_vp, _i, _vp, _i, _vp, _i64, _vp]),
    "clipk_axpby_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "clipk_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "clipk_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, C.c_uint32, _vp]),
    "clipk_rope_qk": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "clipk_attn_fwd_rot": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "clipk_attn_varlen_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "clipk_attn_varlen_fwd_rot": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "clipk_attn_varlen_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _f, C.c_uint32,
                                   _vp]),
    "clipk_attn_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "clipk_attn_f32_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "clipk_dropout_f32": (_i, [_vp, _vp, _vp, _i64, _f, C.c_uint32, _vp]),
    "clipk_colsum_f32": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "clipk_colreduce_batched": (_i, [_vp, _i, _i, _vp]),
    "clipk_set_dropout_epoch": (_i, [_vp]),
    "clipk_gemm_wgrad_f32": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "clipk_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "clipk_embed_bwd_workspace": (_sz, [_i, _i, _i, _i]),
    "clipk_embed_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "clipk_pool_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "clipk_pool_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "clipk_pool_varlen_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipk_layernorm_bwd2": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "clipk_pool_varlen_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "clipk_sumsq_workspace": (_sz, [_i64]),
    "clipk_sumsq": (_i, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "clipk_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp]),
}

_lib = None
ABI_VERSION = 6          # CLIPK_ABI_VERSION of the library these signatures / the GemmArgs layout were written for


def load() -> C.CDLL:
    """Load libclipk.so once; raise loudly (no fallback) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClipkError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (there is no CPU fallback). "
            "Build it with `make` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    lib.clipk_version.restype = C.c_int
    got = lib.clipk_version()
    if got != ABI_VERSION:               # a stale in-tree build: argument structs would be misread
        raise ClipkError(f"{LIB_PATH} has ABI version {got}, this binding expects {ABI_VERSION}: rebuild it (`make`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().clipk_status_string(status).decode()
        raise ClipkError(f"{what} failed: {msg} (status {status})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream
