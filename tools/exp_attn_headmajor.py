#!/usr/bin/env python3
"""Would the whole-head attention kernels gain from head-major q / k / v / dO ([B][H][3][L][D]: a head's 256 rows of 48
bytes contiguous instead of 2880 / 960 bytes apart)?  TIMING ONLY: the probe build (make tools/probes/libattn_hm.so)
addresses the same buffers as if they were laid out that way - same byte count, garbage values - so the cost of a real
layout change (qkv / dO epilogues of the producing GEMMs) can be decided before it is built.

    make tools/probes/libattn_hm.so && python3 tools/exp_attn_headmajor.py [B]

Interleaved rounds after a warm-up (the clock ramp of a fresh process is worth 10-20 %)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_dplm_amd import ops, _ffi  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L, H, D = 256, 20, 24
vp, i32, f32, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32


def load(path):
    lib = C.CDLL(path)
    lib.clipk_attn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, u32, vp]
    lib.clipk_attn_bwd.restype = i32
    lib.clipk_attn_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp]
    lib.clipk_attn_fwd.restype = i32
    return lib


libs = {"token-major (product)": load(_ffi.LIB_PATH), "head-major (probe)": load(os.path.join(ROOT, "tools", "probes", "libattn_hm.so"))}
rnd = lambda s: torch.randn(s, device=dev).to(torch.bfloat16)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
qkv = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
out2, lse2 = torch.empty_like(out), torch.empty_like(lse)
delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)


def fwd(lib):
    assert lib.clipk_attn_fwd(qkv.data_ptr(), None, None, None, out2.data_ptr(), lse2.data_ptr(), B, L, H, D, D ** -0.5, 0.0, 0, None) == 0


def bwd(lib):
    assert lib.clipk_attn_bwd(qkv.data_ptr(), None, cos.data_ptr(), sin.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                              delta.data_ptr(), dqkv.data_ptr(), B, L, H, D, D ** -0.5, 2, 0.0, 0, None) == 0


def timeit(f, n=30):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for lib in libs.values():
    for _ in range(150):
        fwd(lib); bwd(lib)
torch.cuda.synchronize()
print(f"hd 24 whole-head attention, B = {B}, L = {L}, H = {H}: us per launch, 5 interleaved rounds")
for name, f in (("forward ", fwd), ("backward", bwd)):
    res = {k: [] for k in libs}
    for r in range(5):
        for k, lib in libs.items():
            res[k].append(timeit(lambda: f(lib)))
    for k, v in res.items():
        print(f"  {name} {k:24s}", " ".join(f"{x:7.1f}" for x in v), f"  median {sorted(v)[2]:7.1f}")
