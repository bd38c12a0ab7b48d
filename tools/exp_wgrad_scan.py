import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_kernels as bk
from clip_dplm_amd import ops
ops.set_option("wgrad_kernel", 3)
for name, N, K in (("esm out", 480, 480), ("esm fc1", 1920, 480), ("rna fc1", 2048, 768)):
    for M in (32768, 65536, 131072, 262144):
        dy, x = bk.rnd((M, N), scale=0.1), bk.rnd((M, K))
        med, mn = bk.timeit(lambda: ops.gemm_wgrad(dy, x, want_bias=True), iters=10, rounds=7)
        print(f"{name:8s} M={M:7d} {med*1e3:8.1f} us  {2.0*M*N*K/med/1e9:6.0f} TF/s", flush=True)
