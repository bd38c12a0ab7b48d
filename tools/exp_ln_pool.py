import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_kernels as bk
from clip_dplm_amd import ops
dev = torch.device("cuda:0")
for (B, L, cols, xdt) in ((1024, 256, 768, torch.bfloat16), (1024, 256, 768, torch.float32), (1024, 256, 480, torch.float32)):
    x = torch.randn(B * L, cols, device=dev).to(xdt)
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    pooled, mean, rstd, wrow = ops.layernorm_meanpool_fwd(x, g, b, 1e-5, B, L)
    dp = torch.randn(B, cols, device=dev)
    dy = torch.randn(B * L, cols, device=dev)
    for name, fn in (("pool fwd", lambda: ops.layernorm_meanpool_fwd(x, g, b, 1e-5, B, L)),
                     ("ln fwd f32 out", lambda: ops.layernorm_fwd(x, g, b, 1e-5)),
                     ("pool bwd f32 dx", lambda: ops.layernorm_meanpool_bwd(dp, wrow, x, g, mean, rstd, B, L, want_f32=True)),
                     ("pool bwd bf16 dx", lambda: ops.layernorm_meanpool_bwd(dp, wrow, x, g, mean, rstd, B, L, want_f32=False, want_bf16=True)),
                     ("ln bwd f32 dy f32 dx", lambda: ops.layernorm_bwd(dy, x, g, None, mean, rstd, want_f32=True)),
                     ("ln bwd f32 dy bf16 dx", lambda: ops.layernorm_bwd(dy, x, g, None, mean, rstd, want_f32=False, want_bf16=True))):
        med, mn = bk.timeit(fn, iters=5, rounds=5)
        print(f"B={B} L={L} cols={cols} x={str(xdt)[6:]:8s} {name:24s} {med*1e3:8.1f} us", flush=True)
