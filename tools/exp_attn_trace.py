#!/usr/bin/env python3
"""Per-head timeline of the whole-head attention backward (attn_bwd_fused32<24>, the ESM-2-35M shape at the metric batch):
an experiment build with s_memtime stamps between the kernel's phases, summed over the heads of workgroup 0.

    make tools/probes/libattn_trace.so && python3 tools/exp_attn_trace.py [B] [24 | 96]

24 (default): attn_bwd_fused32<24>, ESM-2-35M heads with RoPE on pre-rotated rows; 96: attn_bwd_fused96w8_kernel, the RNA encoder's
heads (8 x 96, no rotation).

Prints cycles (and us at the in-kernel clock, s_memtime / s_memrealtime) per phase and head; the forward that produces
`out` / `lse` runs on the product library, only the traced backward on the experiment build."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
HD96 = len(sys.argv) > 2 and sys.argv[2] == "96"
L, H, D = (256, 8, 96) if HD96 else (256, 20, 24)
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libattn_trace.so"))
vp, i32, f32, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
lib.clipk_attn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, u32, vp]
lib.clipk_attn_bwd.restype = i32
lib.clipk_attn_set_trace.argtypes = [vp]
rnd = lambda s: torch.randn(s, device=dev).to(torch.bfloat16)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
qkv = rnd((B * L, 3 * H * D))
dout = rnd((B * L, H * D))
if HD96:
    out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=None, q_scale=D ** -0.5)
else:
    out, lse = ops.attn_fwd_rot_(qkv, B, L, H, D, (cos, sin), q_scale=D ** -0.5)
delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)
trace = torch.zeros(16, dtype=torch.int64, device=dev)


def launch():
    rc = lib.clipk_attn_bwd(qkv.data_ptr(), None, None if HD96 else cos.data_ptr(), None if HD96 else sin.data_ptr(),
                            out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, L, H, D,
                            D ** -0.5, 0 if HD96 else 1, 0.0, 0, None)
    assert rc == 0, rc


for _ in range(300):                                  # warm clocks
    launch()
torch.cuda.synchronize()
ref = (ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=None, q_scale=D ** -0.5) if HD96 else
       ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=(cos, sin), q_scale=D ** -0.5, prerotated=True))
ref = ref[0] if isinstance(ref, (tuple, list)) else ref
assert torch.equal(ref, dqkv), "the traced build must produce the product's gradients"
assert lib.clipk_attn_set_trace(trace.data_ptr()) == 0
N = 20


def traced(stagger):
    lib.clipk_attn_set_stagger(stagger)
    acc = torch.zeros(16, dtype=torch.float64)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    us = 0.0
    for _ in range(N):
        s.record()
        launch()
        e.record()
        torch.cuda.synchronize()
        us += s.elapsed_time(e) * 1e3 / N
        acc += trace.cpu().double()
    return (acc / N).tolist(), us


# experiment: the second workgroup of every CU (by hardware wave slot) starts k x ~3.9 us late
for k in [int(v) for v in os.environ.get("EXP_STAGGER", "").split()]:
    tt, us = traced(k)
    print(f"  start offset {k} x 3.9 us: {us:7.1f} us per launch; workgroup 0: {tt[9] / tt[8]:.0f} cycles per head, sweep {tt[2] / tt[8]:.0f}")
t, us0 = traced(0)
lib.clipk_attn_set_trace(None)
print(f"  no offset: {us0:7.1f} us per launch")
heads, cyc, ticks = t[8], t[9], t[10]
ghz = cyc / max(ticks, 1) * 0.1
names = ["wait for the prefetched rows + K/V staging + delta + dQ image zero (to the 1st barrier)",
         "K/V fragments + Q/dO staging (2 barriers)", "the sweep (8 steps)", "RoPE table row + issuing the next head's loads",
         "dK/dV images + barrier", "gradient rows: image reads, RoPE^T, stores issued", "closing barrier", "loop bookkeeping"]
if HD96:
    names = ["wait for the head's rows, K rows -> LDS, block 0, delta (2 barriers)",
             "sweep: S / dP, softmax, dS^T, dV / dK, staging of the next block (8 steps)", "sweep: dQ tiles, barrier, dQ rows, delta, barrier (8 steps)",
             "issuing the next head's loads", "sweep: the barrier after the first part (8 steps)", "dK / dV images, barriers, row stores", "-", "loop bookkeeping"]
print(f"{'attn_bwd_fused96w8' if HD96 else 'attn_bwd_fused32<24>'} B={B} L={L}: workgroup 0 walked {heads:.0f} heads in {cyc:.0f} cycles at {ghz:.2f} GHz in-kernel "
      f"= {cyc / heads:.0f} cycles = {cyc / heads / ghz / 1e3:.2f} us per head")
for i, n in enumerate(names):
    print(f"  {t[i] / heads:8.0f} cycles {t[i] / heads / ghz / 1e3:6.2f} us  {100 * t[i] / cyc:5.1f} %  {n}")
