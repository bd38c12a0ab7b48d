#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the BASELINE-config-2 shapes (run on the GPU box).

    python tools/bench_kernels.py gemm        # gemm_nt v1 vs v2, interleaved rounds in one process
    python tools/bench_kernels.py wgrad | attn | ln | all

Random data (cdna_hip_programming.md §5.4 rule 25), HIP events on the launch stream, median of rounds.
"""
from __future__ import annotations

import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("BENCH_LIB"):          # an experiment build of the library (make tools/probes/libclipk_exp.so)
    from clip_dplm_amd import _ffi  # noqa: E402
    _ffi.LIB_PATH = os.path.join(ROOT, os.environ["BENCH_LIB"])
from clip_dplm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
T = int(os.environ.get("BENCH_BATCH", "512")) * 256      # tokens per step at B = 512 (BENCH_BATCH), L = 256


def timeit(fn, iters=5, rounds=5):
    fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / iters)
    return statistics.median(res), min(res)


def rnd(shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).to(dtype)


GEMM_SHAPES = [  # (name, M, N, K, epilogue)
    ("esm qkv", T, 1440, 480, "bias"), ("esm out", T, 480, 480, "res32"), ("esm fc1", T, 1920, 480, "gelu+pre"),
    ("esm fc2", T, 480, 1920, "res32"), ("rna qkv", T, 2304, 768, "bias"), ("rna out", T, 768, 768, "res32"),
    ("rna fc1", T, 2048, 768, "gelu+pre"), ("rna fc2", T, 768, 2048, "res32"),
    ("esm d_fc2", T, 1920, 480, "dact"), ("rna d_qkv", T, 768, 2304, "res32"), ("esm d_qkv", T, 480, 1440, "bias"),
    # round 3: GELU'(u) kept as 8-bit codes (the product path's FFN epilogues)
    ("esm fc1", T, 1920, 480, "gelu+d8"), ("esm d_fc2", T, 1920, 480, "dact8"),
    ("rna fc1", T, 2048, 768, "gelu+d8"), ("rna d_fc2", T, 2048, 768, "dact8"),
]


def bench_gemm():
    """arms interleaved in one process: 128^2 kernel with the run-time epilogue (round-1 baseline), 128^2 with the
    specialised straight-line epilogue, 256^2 phase-interleaved kernel (option gemm_kernel = 3).  Arms are dicts of
    libclipk options (ops.set_option); BENCH_NT / BENCH_NWG / BENCH_STAGGER / BENCH_ABL (the last needs a
    -DCLIPK_EXPERIMENTS build) add arms."""
    arms = [("v2gen", {"gemm_epi_generic": 1, "gemm_kernel": 2}), ("v2", {"gemm_kernel": 2}), ("v3", {"gemm_kernel": 3}),
            ("v4", {"gemm_kernel": 4})]
    if os.environ.get("BENCH_NT"):
        arms.append(("v3nt1", {"gemm_kernel": 3, "epi_nt": 1}))
    for ab in os.environ.get("BENCH_ABL", "").split():
        arms.append(("v3a" + ab, {"gemm_kernel": 3, "gemm_abl": int(ab)}))
    for nw in os.environ.get("BENCH_NWG", "").split():
        arms.append(("v3w" + nw, {"gemm_kernel": 3, "gemm_nwg": int(nw)}))
    for st in os.environ.get("BENCH_STAGGER", "").split():
        arms.append(("v3s" + st, {"gemm_kernel": 3, "gemm_stagger": int(st)}))
    for ab in os.environ.get("BENCH_ABL4", "").split():
        arms.append(("v4a" + ab, {"gemm_kernel": 4, "gemm_abl": int(ab)}))
    for nw in os.environ.get("BENCH_NWG4", "").split():          # "256:1" = grid 256 (one workgroup per CU), ablation 1
        g, _, ab = nw.partition(":")
        arms.append(("v4w" + nw, {"gemm_kernel": 4, "gemm_nwg": int(g), "gemm_abl": int(ab or 0)}))
    for st in os.environ.get("BENCH_STAGGER4", "").split():
        arms.append(("v4s" + st, {"gemm_kernel": 4, "gemm_stagger": int(st)}))
    if os.environ.get("BENCH_ARMS"):
        keep = os.environ["BENCH_ARMS"].split()
        arms = [a for a in arms if a[0] in keep or a[0].startswith(("v4s", "v4a", "v3a", "v4w"))]
    print(f"{'shape':12s} {'M':>7s} {'N':>5s} {'K':>5s} {'epi':9s} | " + " | ".join(f"{n:>5s} us  TF/s" for n, _ in arms)
          + " | v2/v2gen v3/v2 v3/v4")
    tot = {n: 0.0 for n, _ in arms}
    for name, M, N, K, epi in GEMM_SHAPES:
        a, b = rnd((M, K)), rnd((N, K), scale=0.05)
        bias = torch.randn(N, device=DEV)
        kw = {"bias": bias}
        if epi == "res32":
            kw.update(residual=torch.randn(M, N, device=DEV), out_dtype=torch.float32)
        elif epi == "gelu+pre":
            kw.update(act="gelu", out_preact=True)
        elif epi == "dact":
            kw = {"dact_aux": rnd((M, N)), "dact": "gelu"}
        elif epi == "gelu+d8":
            kw.update(act="gelu", out_preact=True, aux_u8=True)
        elif epi == "dact8":
            kw = {"dact_aux": torch.randint(0, 256, (M, N), device=DEV, dtype=torch.uint8), "dact": "gelu"}
        out = {}
        for n, opts in arms:
            ops.reset_options()
            for k, v in opts.items():
                ops.set_option(k, v)
            med, mn = timeit(lambda: ops.gemm_nt(a, b, **kw))
            out[n] = med
            tot[n] += med
        fl = 2.0 * M * N * K
        print(f"{name:12s} {M:7d} {N:5d} {K:5d} {epi:9s} | "
              + " | ".join(f"{out[n] * 1e3:7.1f} {fl / out[n] / 1e9:5.0f}" for n, _ in arms)
              + (f" | {out['v2gen'] / out['v2']:.2f}x {out['v2'] / out['v3']:.2f}x {out['v3'] / out['v4']:.2f}x"
                 if all(k in out for k in ("v2gen", "v2", "v3", "v4")) else ""), flush=True)
        del a, b, kw
    ops.reset_options()
    print("sum: " + ", ".join(f"{n} {tot[n]:.2f} ms" for n, _ in arms))


def bench_wgrad():
    print(f"{'shape':12s} {'M':>7s} {'N':>5s} {'K':>5s} |   v2 us  TF/s |   v3 us  TF/s | v3/v2")
    tot = {2: 0.0, 3: 0.0}
    for name, M, N, K, _ in GEMM_SHAPES[:8]:
        dy, x = rnd((M, N), scale=0.1), rnd((M, K))
        out = {}
        for ver in (2, 3):
            ops.set_option("wgrad_kernel", ver)
            med, mn = timeit(lambda: ops.gemm_wgrad(dy, x, want_bias=True))
            out[ver] = med
            tot[ver] += med
        fl = 2.0 * M * N * K
        print(f"{name:12s} {M:7d} {N:5d} {K:5d} | {out[2] * 1e3:7.1f} {fl / out[2] / 1e9:5.0f} | "
              f"{out[3] * 1e3:7.1f} {fl / out[3] / 1e9:5.0f} | {out[2] / out[3]:.2f}x", flush=True)
    ops.reset_options()
    print(f"sum: v2 {tot[2]:.2f} ms, v3 {tot[3]:.2f} ms")


def bench_wgrad_splits():
    """v3 weight-gradient kernel: M-split count sweep per shape (option wgrad_splits; 0 = the plan's own choice)."""
    ops.set_option("wgrad_kernel", 3)
    arms = [0] + [int(v) for v in os.environ.get("BENCH_SPLITS", "8 16 24 32").split()]
    print(f"{'shape':12s} " + " ".join(f"{('s=' + str(a)) if a else 'auto':>9s}" for a in arms) + "   (us)")
    for name, M, N, K, _ in GEMM_SHAPES[:8]:
        dy, x = rnd((M, N), scale=0.1), rnd((M, K))
        res = []
        for a in arms:
            ops.set_option("wgrad_splits", a)
            med, _ = timeit(lambda: ops.gemm_wgrad(dy, x, want_bias=True))
            res.append(med * 1e3)
        print(f"{name:12s} " + " ".join(f"{r:9.1f}" for r in res), flush=True)
    ops.reset_options()


def bench_attn():
    for name, B, L, H, D, rope in (("esm 35M", 512, 256, 20, 24, True), ("rna", 512, 256, 8, 96, False)):
        qkv = rnd((B * L, 3 * H * D))
        r = None
        if rope:
            inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
            fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
            r = (fr.cos().contiguous().to(DEV), fr.sin().contiguous().to(DEV))
        out, lse = ops.attn_fwd(qkv, B, L, H, D, rope=r, q_scale=D ** -0.5)
        dout = rnd((B * L, H * D))
        f, _ = timeit(lambda: ops.attn_fwd(qkv, B, L, H, D, rope=r, q_scale=D ** -0.5))
        bw, _ = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, rope=r, q_scale=D ** -0.5))
        fl = 4.0 * B * H * L * L * D
        print(f"attn {name:8s} fwd {f * 1e3:8.1f} us {fl / f / 1e9:6.0f} TF/s | bwd {bw * 1e3:8.1f} us "
              f"{2.5 * fl / bw / 1e9:6.0f} TF/s (algorithmic 10*B*H*L^2*D)")


def bench_ln():
    for cols in (480, 768):
        x = torch.randn(T, cols, device=DEV)
        g, b = torch.ones(cols, device=DEV), torch.zeros(cols, device=DEV)
        f, _ = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, want_f32=False, want_bf16=True))
        _, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, want_f32=False, want_bf16=True)
        dy = rnd((T, cols))
        bw, _ = timeit(lambda: ops.layernorm_bwd(dy, x, g, None, mean, rstd, dx_add=x, want_f32=True, want_bf16=True))
        fb = T * cols * (4 + 2)
        bb = T * cols * (2 + 4 + 4 + 4 + 2)
        print(f"layernorm d={cols}: fwd {f * 1e3:7.1f} us {fb / f / 1e9:6.2f} TB/s | bwd {bw * 1e3:7.1f} us {bb / bw / 1e9:6.2f} TB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("wgrad", "all"):
        bench_wgrad()
    if what == "wsplit":
        bench_wgrad_splits()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("ln", "all"):
        bench_ln()
