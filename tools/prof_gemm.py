#!/usr/bin/env python3
"""Run the Linear kernels of the metric configuration (M = B x 256 token rows) a few times, ONE shape per epilogue mode
so that a kernel name in the profiler's output is one shape, plus one weight-gradient shape - for rocprofv3 --pmc /
--stats passes (tools/pmc_kernels.sh):

    python3 tools/prof_gemm.py [B] [launches]

gemm_nt_v3_kernel<0> plain bf16 (ESM qkv 1440 x 480), <1> f32 residual (ESM fc2 480 x 1920), <7> GELU + 8-bit GELU' codes
(ESM fc1 1920 x 480), <8> x GELU' from codes (ESM fc2 dgrad 1920 x 480), wgrad_v3_kernel (ESM fc1: dW[1920, 480])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_dplm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
T = B * 256
rnd = lambda s, sc=1.0: (torch.randn(s, device=dev) * sc).to(torch.bfloat16)
x480, x1920 = rnd((T, 480)), rnd((T, 1920))
w_qkv, w_fc1, w_fc2 = rnd((1440, 480), 0.05), rnd((1920, 480), 0.05), rnd((480, 1920), 0.05)
b1440, b1920, b480 = (torch.randn(n, device=dev) for n in (1440, 1920, 480))
res = torch.randn(T, 480, device=dev)
codes = torch.randint(0, 256, (T, 1920), device=dev, dtype=torch.uint8)
dy = rnd((T, 1920), 0.1)
for _ in range(N):
    ops.gemm_nt(x480, w_qkv, bias=b1440)
for _ in range(N):
    ops.gemm_nt(x1920, w_fc2, bias=b480, residual=res, out_dtype=torch.float32)
for _ in range(N):
    ops.gemm_nt(x480, w_fc1, bias=b1920, act="gelu", out_preact=True, aux_u8=True)
for _ in range(N):
    ops.gemm_nt(x480, w_fc1, dact_aux=codes, dact="gelu")
for _ in range(N):
    ops.gemm_wgrad(dy, x480, want_bias=True)
torch.cuda.synchronize()
print("prof_gemm done", B, N)
