"""Flat-buffer optimiser: AdamW + global-norm gradient clipping in two HIP kernels, sharded over ranks.

Reference step (current/rna_clip_codes.ipynb:2033-2034,2076-2077; old/clip_opt.py:168-171):
    clip_grad_norm_(model.parameters(), 1.0); AdamW(lr, weight_decay=0.01).step()   [+ CosineAnnealingLR(T_max=20)]

MI355X design:
  * every trainable parameter lives in ONE flat f32 buffer (params are views), gradients in a second one,
    Adam moments in two more: the update is a single grid-stride kernel at HBM speed (16 B/param/step)
    instead of ~300 small foreach launches;
  * multi-GPU (ZeRO-1 style, one process per GPU): reduce-scatter(sum) the flat gradient over RCCL, each rank
    updates its 1/W shard of the master weights and moments, all-gather the updated parameters.  On the
    fully-connected xGMI mesh both collectives drive all 7 links at once; a ring all-reduce would be
    per-link bound.  The global gradient norm is one scalar all-reduce of the shard sums of squares.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import os
import struct

import torch
import torch.distributed as dist

from . import functional as KF
from . import ops

_ALIGN = 64          # elements: every parameter starts on a 256-byte boundary (16-byte aligned kernel pointers)

# kernel namespace (see loss.py): gloo/CPU tests of the sharding bookkeeping substitute a torch restatement
_kernels = ops


class FlatParams:
    """Re-home the trainable parameters of `module` into one flat f32 buffer and give them flat .grad views."""

    def __init__(self, module: torch.nn.Module, world_size: int = 1):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        # Sub-modules may ask for some parameters to sit back to back (`flat_param_groups()`: ESM-2 keeps the HF
        # query / key / value Linears but runs ONE fused qkv GEMM; contiguous storage makes the fused [3d, d] weight a
        # zero-copy view instead of a torch.cat per layer and step).  A group is placed where its first member
        # would have been, members packed without padding (numel % 4 == 0 keeps them 16-byte aligned).
        group_of = {}
        for mod in module.modules():
            fn = getattr(mod, "flat_param_groups", None)
            if callable(fn):
                for grp in fn():
                    grp = [q for q in grp]
                    if all(q.requires_grad and q.numel() % 4 == 0 for q in grp):
                        for q in grp:
                            group_of[id(q)] = grp
        order, seen = [], set()
        for p in params:
            if id(p) in seen:
                continue
            grp = group_of.get(id(p), [p])
            for q in grp:
                if id(q) not in seen:
                    seen.add(id(q))
                    order.append((q, q is grp[-1]))               # pad to the alignment only after the last member
        # Gradient buckets (multi-GPU): the parameters of every sub-module that declares `grad_bucket = True` (the two
        # encoder stacks: their hand-written backward announces when the whole stack's gradients are final) are stored
        # together, each bucket padded to a multiple of world_size * _ALIGN so that it can be reduce-scattered /
        # all-gathered on its own while the rest of the backward still runs; everything else forms the last bucket.
        bucket_of, roots = {}, []
        for mod in module.modules():
            if getattr(mod, "grad_bucket", False):
                roots.append(mod)
                for q in mod.parameters():
                    bucket_of.setdefault(id(q), len(roots) - 1)
        nb = len(roots) + 1
        by_bucket = [[] for _ in range(nb)]
        for item in order:
            by_bucket[bucket_of.get(id(item[0]), nb - 1)].append(item)
        chunk = _ALIGN * max(world_size, 1)
        params, offs, total, self.buckets = [], [], 0, []
        for b, items in enumerate(by_bucket):
            if not items:
                continue
            start = total
            for p, pad in items:
                params.append(p)
                offs.append(total)
                total += ((p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN) if pad else p.numel()
            total = (total + chunk - 1) // chunk * chunk      # every bucket splits evenly into _ALIGN-aligned pieces
            self.buckets.append((start, total, roots[b] if b < len(roots) else None))
        self.params, self.offsets, self.numel = params, offs, total
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                self.data[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):          # keep .grad pointing into the flat buffer
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedAdamW:
    """AdamW (torch.optim.AdamW semantics) + clip_grad_norm_ folded in, on FlatParams; optional sharding.

    Sharded (group given, one process per GPU, RCCL): ZeRO-1 by BUCKET.  The flat buffer is a sequence of buckets (one
    per encoder stack + one for the rest, FlatParams); rank r owns the r-th 1/W piece of every bucket, so a bucket can be
    reduce-scattered as soon as its gradients are final: the encoder stacks' backward calls back when that is the case
    and the collective runs on a side stream under the rest of the backward (`overlap`, default on for NCCL/RCCL).
    step() reduces what is left, clips by the global norm (one scalar all-reduce), updates this rank's pieces and
    all-gathers the updated parameters bucket by bucket.  `grad_comm_dtype=torch.bfloat16` halves the gradient payload
    (sum in bf16 on the wire: off by default, the reference reduces f32 gradients)."""

    def __init__(self, module: torch.nn.Module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                 max_grad_norm: Optional[float] = 1.0, group=None, overlap: Optional[bool] = None,
                 grad_comm_dtype: torch.dtype = torch.float32):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        # parameter order of module.parameters() (what torch.optim.AdamW(model.parameters()) would index): the flat
        # buffer may store them in another order (fused qkv groups, buckets)
        # - frozen ones included, as torch.optim.AdamW(model.parameters()) does (triple_flow/5_training.py:128 over a
        # model whose ESM parameters are frozen, 3_esm_integration.py:83-84); they get an index and never any state
        self.param_order = list(module.parameters())
        self.flat = FlatParams(module, self.world)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        n = self.flat.numel
        self.shard = n // self.world
        dev = self.flat.data.device
        # this rank's pieces: (flat_lo, flat_hi, offset inside the shard-sized buffers m / v / gshard)
        self.pieces, off = [], 0
        for (s0, s1, _) in self.flat.buckets:
            piece = (s1 - s0) // self.world
            self.pieces.append((s0 + self.rank * piece, s0 + (self.rank + 1) * piece, off))
            off += piece
        assert off == self.shard
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.gshard = torch.empty(self.shard, dtype=torch.float32, device=dev) if group is not None else None
        self.norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_comm_dtype = grad_comm_dtype
        self._nccl = group is not None and dist.get_backend(group) != "gloo"
        self.overlap = (self._nccl if overlap is None else bool(overlap)) and group is not None
        self._comm = torch.cuda.Stream(device=dev) if (self._nccl and dev.type == "cuda") else None
        self._reduced = [False] * len(self.flat.buckets)
        # hipGraph replay (training.GraphedTrainStep): what changes from step to step - lr and the two bias corrections -
        # is read by the kernel from this device array instead of from launch arguments
        self.hyper = None
        self._hyper_host = None
        self._graph_body = False
        if self.overlap:
            for b, (_, _, root) in enumerate(self.flat.buckets):
                if root is not None:
                    root._grad_bucket_done = (lambda b=b: self._reduce_bucket(b))
                    root._grad_bucket_begin = (lambda b=b: self._retract_bucket(b))

    def zero_grad(self):
        self.flat.zero_grad()
        KF.discard_deferred_ln_param_grads()
        self._reduced = [False] * len(self.flat.buckets)
        # (the stacks' count of outstanding backwards is NOT reset here: with the common order forward -> zero_grad() ->
        # backward a stack applied twice would fire its reduce-scatter after the first of its two backwards, ADVICE r03;
        # step() clears it, so a forward that was never differentiated costs one step's overlap, never a gradient)

    def _retract_bucket(self, b: int) -> None:
        """A stack whose bucket was already sent in this step is about to produce more gradient (second backward()
        before step(): gradient accumulation): the sent result is dropped and the bucket is reduced again when it is
        final; the collective still reading flat.grad must finish before the new backward writes into it."""
        if not self._reduced[b]:
            return
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        self._reduced[b] = False

    # ---- gradient reduce-scatter of one bucket (called from the encoder stacks' backward, or from step())
    def _reduce_bucket(self, b: int) -> None:
        if self.group is None or self._reduced[b]:
            return
        s0, s1, _ = self.flat.buckets[b]
        lo, hi, off = self.pieces[b]
        dst = self.gshard[off:off + (hi - lo)]
        src = self.flat.grad[s0:s1]
        if not self._nccl:                                   # gloo (CPU tests) has no reduce_scatter
            tmp = src.clone()                                # flat.grad stays this rank's own sum (see _retract_bucket)
            dist.all_reduce(tmp, group=self.group)
            dst.copy_(tmp[lo - s0:hi - s0])
        else:
            cur = torch.cuda.current_stream()
            stream = self._comm if self._comm is not None else cur
            if stream is not cur:
                stream.wait_event(cur.record_event())        # the bucket's gradients are final at this point
            with torch.cuda.stream(stream):
                if self.grad_comm_dtype == torch.float32:
                    dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    lowp = src.to(self.grad_comm_dtype)
                    out = torch.empty(hi - lo, dtype=self.grad_comm_dtype, device=src.device)
                    dist.reduce_scatter_tensor(out, lowp, op=dist.ReduceOp.SUM, group=self.group)
                    dst.copy_(out)
        self._reduced[b] = True

    def enable_device_hyper(self) -> None:
        """Keep {lr, 1 - beta1^t, sqrt(1 - beta2^t)} in device memory (see prepare_step)."""
        if self.hyper is None:
            self.hyper = torch.zeros(3, dtype=torch.float32, device=self.flat.data.device)
            self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory() if self.flat.data.is_cuda else torch.zeros(3)

    def prepare_step(self, lr: Optional[float] = None) -> None:
        """Host side of one optimiser step when the step itself is a hipGraph replay: count the step and upload its
        hyper-parameters (bias corrections in double, as torch.optim.AdamW computes them)."""
        self.enable_device_hyper()
        self.step_count += 1
        lr = self.lr if lr is None else lr
        t = self.step_count
        # exactly what clipk_adamw_step computes from its (float) arguments: betas rounded to f32 first, pow / sqrt in double
        b1 = struct.unpack("f", struct.pack("f", self.betas[0]))[0]
        b2 = struct.unpack("f", struct.pack("f", self.betas[1]))[0]
        self._hyper_host[0] = lr
        self._hyper_host[1] = 1.0 - math.pow(b1, float(t))
        self._hyper_host[2] = math.sqrt(1.0 - math.pow(b2, float(t)))
        self.hyper.copy_(self._hyper_host, non_blocking=True)

    @torch.no_grad()
    def step(self, lr: Optional[float] = None) -> torch.Tensor:
        """Returns the (device) squared global gradient norm before clipping."""
        if not self._graph_body:                 # (inside a captured step the host bookkeeping is prepare_step()'s)
            self.step_count += 1
        lr = self.lr if lr is None else lr
        if self.group is not None:
            for b in range(len(self.flat.buckets)):
                self._reduce_bucket(b)                       # whatever the backward has not sent yet
            if self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
            g = self.gshard
        else:
            g = self.flat.grad
        _kernels.sumsq(g, out=self.norm_sq)
        if self.group is not None:
            dist.all_reduce(self.norm_sq, group=self.group)
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        for (lo, hi, off) in self.pieces:
            n = hi - lo
            _kernels.adamw_step(self.flat.data[lo:hi], g[off:off + n], self.m[off:off + n], self.v[off:off + n], lr,
                                self.betas[0], self.betas[1], self.eps, self.wd, max(self.step_count, 1),
                                grad_norm_sq=self.norm_sq if clip else None,
                                max_norm=self.max_grad_norm if clip else 0.0,
                                **({"hyper": self.hyper} if self._graph_body else {}))
        if self.group is not None:
            for (s0, s1, _), (lo, hi, _off) in zip(self.flat.buckets, self.pieces):
                w = self.flat.data[lo:hi]
                dist.all_gather_into_tensor(self.flat.data[s0:s1], w if self._nccl else w.clone(), group=self.group)
        for (_, _, root) in self.flat.buckets:
            if root is not None:
                root._bucket_pending = 0     # forwards that were never differentiated must not block the next step
        KF.mark_weights_dirty()              # bf16 W / W^T copies of the Linear weights are stale now ...
        if self.flat.data.is_cuda and os.environ.get("CLIPK_BATCH_REFRESH", "1") != "0":
            KF.refresh_weight_caches()       # ... rebuild them in one launch (what is left is refreshed lazily)
        return self.norm_sq

    # ---- checkpoint format: torch.optim.AdamW's, so optimizer_state of the reference's checkpoints
    # (triple_flow/5_training.py:335-358: torch.save({'model_state', 'optimizer_state', ...})) is interchangeable in
    # both directions.  Parameter indices follow module.parameters() order (what the reference hands its optimiser).
    def _gathered_moments(self):
        """Full-length (m, v) on every rank: the moments are sharded 1/W per rank (ZeRO-1)."""
        if self.group is None or self.world == 1:
            return self.m, self.v
        full = [torch.zeros(self.flat.numel, dtype=torch.float32, device=self.m.device) for _ in range(2)]
        for (s0, s1, _), (lo, hi, off) in zip(self.flat.buckets, self.pieces):
            for dst, src in ((full[0], self.m), (full[1], self.v)):
                if self._nccl:
                    dist.all_gather_into_tensor(dst[s0:s1], src[off:off + hi - lo].contiguous(), group=self.group)
                else:
                    dst[lo:hi].copy_(src[off:off + hi - lo])
        if not self._nccl:
            dist.all_reduce(full[0], group=self.group)
            dist.all_reduce(full[1], group=self.group)
        return full[0], full[1]

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [..]}
        with cloned, full (un-sharded) per-parameter tensors — loadable by torch.optim.AdamW and by FusedAdamW at any
        world size.  Collective when sharded: call on every rank."""
        m, v = self._gathered_moments()
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        state = {}
        for i, p in enumerate(self.param_order):
            o = off.get(id(p))
            st = {}
            if o is not None and self.step_count > 0:    # torch creates per-parameter state lazily at the first step,
                                                         # and never for a parameter without a gradient (frozen)
                st = {"step": torch.tensor(float(self.step_count)),
                      "exp_avg": m[o:o + p.numel()].view(p.shape).clone(),
                      "exp_avg_sq": v[o:o + p.numel()].view(p.shape).clone()}
            if st:
                state[i] = st
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(self.param_order)))}
        if self.max_grad_norm is not None:
            group["max_grad_norm"] = self.max_grad_norm  # extra key (the clip is folded into the step); torch ignores it
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW.state_dict() (and what state_dict() above returns); the legacy
        {'step', 'm', 'v'} flat format of round 1 is still read when world size matches."""
        if "param_groups" not in sd:                     # legacy private format (single process only)
            if self.world != 1:
                raise ValueError("the round-1 {'step','m','v'} optimiser state holds one rank's shard: load a "
                                 "torch.optim.AdamW-format state (FusedAdamW.state_dict()) instead")
            self.step_count = int(sd["step"])
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
            self.lr = sd.get("lr", self.lr)
            return
        g = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(g["params"]) != len(self.param_order):
            raise ValueError(f"optimizer state has {len(g['params'])} parameters in {len(sd['param_groups'])} group(s); "
                             f"this model has {len(self.param_order)} in one group")
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("amsgrad / maximize states are not supported by the fused AdamW step")
        self.lr = float(g["lr"])
        self.betas = tuple(g["betas"])
        self.eps = float(g["eps"])
        self.wd = float(g["weight_decay"])
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, p in enumerate(self.param_order):
            st = sd["state"].get(g["params"][i])
            if not st or id(p) not in off:               # no state yet, or a frozen parameter's entry: nothing to load
                continue
            steps.add(int(round(float(st["step"]))))
            o, n = off[id(p)], p.numel()
            for (lo, hi, soff) in self.pieces:           # the parts of this parameter that live in this rank's pieces
                a, b = max(o, lo), min(o + n, hi)
                if a >= b:
                    continue
                for dst, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq")):
                    src = st[key].reshape(-1).to(device=dst.device, dtype=torch.float32)
                    dst[soff + a - lo:soff + b - lo].copy_(src[a - o:b - o])
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused step keeps one counter")
        self.step_count = steps.pop() if steps else 0


def cosine_annealing_lr(base_lr: float, epoch: int, t_max: int = 20, eta_min: float = 0.0) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (rna_clip_codes.ipynb:2034, T_max=20)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
