#!/usr/bin/env python3
"""bench.py — seq-pairs/sec of one full contrastive TRAINING step (forward + fused InfoNCE + backward + fused
AdamW with global-norm clip) of the BASELINE-config-2 dual encoder on MI355X:

    ESM-2-35M protein encoder (12 x 480, 20 heads, hd 24, ffn 1920, RoPE) + 6 x 768 RNA transformer
    (8 heads, ffn 2048, gelu, post-LN) + ProjectionHeads (P = 512), B = 512 pairs per GPU, L = 256, bf16 MFMA
    with f32 accumulate / residual stream / master weights, both encoders trained, synthetic data,
    random-init weights N(0, 0.02).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank/GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the bf16
MFMA Linear kernel gemm_nt, timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU
oracle timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
HBM_PEAK_GBS = 8000.0


def synth_batch(B, L, rna_dim, device, seed):
    """protein ids uniform over the 20 standard amino-acid ids [4, 24) with <cls>=0 first and <eos>=2
    last, no padding; RNA side N(0,1) features [B, L, rna_dim] (the reference feeds precomputed RNABERT vectors)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0] = 0
    ids[:, -1] = 2
    rna = torch.randn(B, L, rna_dim, generator=g)
    return rna.to(device), ids.to(device)


def cpu_baseline(model_sd, cfg, L, sample_b=8, steps=2):
    """Time the CPU oracle (kind 'port') on a bounded sample of the same workload: forward + backward + AdamW."""
    from oracle import model_ref
    try:
        ncores = len(os.sched_getaffinity(0))             # the cores this process may actually use (cgroup share)
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 64))
    torch.set_num_threads(ncores)
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model_sd.items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    rna, ids = synth_batch(sample_b, L, cfg["rna_dim"], "cpu", 1234)
    times = []
    budget_s, t_start = 30.0, time.perf_counter()
    for it in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _, _ = model_ref.protein_rna_clip_loss(sd, rna, ids, None, None, esm_layers=cfg["esm_layers"],
                                                     esm_heads=cfg["esm_heads"], rna_layers=cfg["rna_layers"],
                                                     rna_heads=cfg["rna_heads"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {it}: {times[-1]:.2f} s on {ncores} threads", file=sys.stderr, flush=True)
        if time.perf_counter() - t_start > budget_s:      # bounded sample: never hold the bench for minutes
            break
    timed = times[1:] if len(times) > 1 else times         # drop the warm-up step when there is more than one
    dt = sum(timed) / len(timed)
    return {"value": round(sample_b / dt, 3), "unit": "seq-pairs/s", "cores": ncores, "kind": "port",
            "sample": f"{len(timed)} training step(s) (fwd+bwd+clip+AdamW) of the CPU oracle at B={sample_b}, L={L}, "
                      f"f32, {'after 1 warm-up step' if len(times) > 1 else 'no warm-up (time budget)'}; {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="pairs per GPU")
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--esm", default="esm2_t12_35M_UR50D")
    ap.add_argument("--freeze-esm", action="store_true", help="reference behaviour (3_esm_integration.py:83-84)")
    ap.add_argument("--dual-stream", action="store_true",
                    help="enqueue the two towers on separate HIP streams (+1.5 %% pairs/s; a kernel's HIP-event time "
                         "then includes waiting for the other tower's kernels, so the per-kernel numbers are not "
                         "the kernels' own durations any more)")
    ap.add_argument("--single-stream", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--micro-batches", type=int, default=1, help="stream pairs per step (batch split over them)")
    ap.add_argument("--wgrad-stream", action="store_true", help="weight-gradient GEMMs on a side stream per tower")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--all-kernel-timers", action="store_true",
                    help="HIP events around EVERY kernel launch of the timed region (costs ~1 %% of the step); "
                         "default: only the dominant kernel, clipk_gemm_nt")
    args = ap.parse_args()

    import clip_dplm_amd as K
    from clip_dplm_amd import ops
    from clip_dplm_amd.distributed import init_distributed
    from clip_dplm_amd.encoders import ESM2_SHAPES

    rank, world, device = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    force_dist = bool(os.environ.get("CLIPK_FORCE_DIST"))     # rehearse the RCCL code path with a 1-rank group
    if force_dist and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    group = dist.group.WORLD if (world > 1 or force_dist) else None

    torch.manual_seed(0)                                  # identical weights on every rank
    model = K.ProteinRNACLIP(esm=args.esm, freeze_protein_encoder=args.freeze_esm).to(device).train()
    for m in model.modules():                             # training-step timing with dropout p = 0
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.dual_stream = bool(args.dual_stream) and not args.single_stream
    model.micro_batches = args.micro_batches
    import clip_dplm_amd.encoders as _enc
    _enc.WGRAD_SIDE_STREAM = args.wgrad_stream
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    opt = K.FusedAdamW(model, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, group=group)
    B, L = args.batch, args.seq_len
    rna, ids = synth_batch(B, L, 768, device, 1234 + rank)

    def step():
        opt.zero_grad()
        loss = model.loss(rna, ids, group=group)
        loss.backward()
        opt.step()
        return loss

    tw = time.perf_counter()
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warm-up step {i}: {time.perf_counter() - tw:.2f} s since start", file=sys.stderr, flush=True)
    timer = None
    if not args.no_kernel_timers:
        # timed region: only the dominant kernel (the roofline object); --all-kernel-timers times every launch
        timer = ops.KernelTimer(None if args.all_kernel_timers else ("gemm_nt",))
        ops.set_kernel_timer(timer)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    # Two more steps OUTSIDE the timed region, one HIP stream, HIP events around every launch: the per-class breakdown
    # (`kernels`) without taxing the timed region (~1 % when every launch is timed), and with --dual-stream also the
    # dominant kernel's own rate (`roofline.one_stream`): with two streams a kernel's event time includes waiting
    # for the other tower's kernels.
    timer_alone = None
    was_dual = model.dual_stream
    if timer is not None and rank == 0 and world == 1 and (was_dual or not args.all_kernel_timers):
        model.dual_stream = False
        step(); torch.cuda.synchronize()
        timer_alone = ops.KernelTimer()            # outside the timed region: every kernel class
        ops.set_kernel_timer(timer_alone)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        model.dual_stream = was_dual
    tmax = torch.tensor([dt], device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    if rank == 0:
        print(f"[bench] {args.steps} timed steps: {dt:.3f} s", file=sys.stderr, flush=True)
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    nl, d, h, f = ESM2_SHAPES[args.esm]
    out = {
        "metric": "seq-pairs/sec/node, contrastive training step, ESM-2-35M dual encoder",
        "value": round(B * world * args.steps / dt, 2),
        "unit": "seq-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: {args.esm} protein encoder + 6x768 RNA transformer, "
                               f"B={B} pairs/GPU, L={L}, full training step (fwd + fused InfoNCE + bwd + fused AdamW/clip), "
                               + ("ESM frozen" if args.freeze_esm else "both encoders trained"),
                   "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                   "projection_dim": 512, "hip_streams": 2 if model.dual_stream else 1},
        "loss": round(float(loss.item()), 5),
    }
    if timer is not None:
        summ = timer.summary()
        g = summ.get("gemm_nt")
        if g:
            achieved = g["work"] / (g["total_ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "clipk_gemm_nt = gemm_nt_v3_kernel / gemm_nt_v2_kernel (bf16 16x16x32 MFMA Linear fwd/dgrad)",
                               "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                               "avg_launch_us": round(g["avg_us"], 2), "launches": g["launches"],
                               "share_of_step": round(g["total_ms"] / (1e3 * dt), 4)}
            if timer_alone is not None and was_dual:
                ga = timer_alone.summary().get("gemm_nt")
                if ga:
                    ach = ga["work"] / (ga["total_ms"] * 1e-3) / 1e12
                    out["roofline"]["one_stream"] = {"achieved": round(ach, 2), "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                                     "avg_launch_us": round(ga["avg_us"], 2), "launches": ga["launches"],
                                                     "note": "same kernels, 2 extra steps outside the timed region "
                                                             "with both towers on one HIP stream"}
        # per-class breakdown: from the timed region with --all-kernel-timers, else from the two extra one-stream steps
        src, steps_src, tag = (summ, args.steps, "timed region") if (args.all_kernel_timers or timer_alone is None) \
            else (timer_alone.summary(), 2, "2 extra steps, one HIP stream, outside the timed region")
        out["kernels"] = {k: {"launches_per_step": v["launches"] // steps_src, "avg_us": round(v["avg_us"], 2),
                              "ms_per_step": round(v["total_ms"] / steps_src, 3),
                              "rate": round(v["work"] / (v["total_ms"] * 1e-3) / 1e12, 3),
                              "rate_unit": "TFLOP/s" if ("gemm" in k or "attn" in k) else "TB/s"}
                          for k, v in src.items()}
        out["kernels"]["_source"] = tag
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd_cpu, {"rna_dim": 768, "esm_layers": nl, "esm_heads": h, "rna_layers": 6,
                                                    "rna_heads": 8}, L)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
