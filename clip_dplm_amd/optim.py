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
        params = [q for q, _ in order]
        offs, total = [], 0
        for p, pad in order:
            offs.append(total)
            total += ((p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN) if pad else p.numel()
        chunk = _ALIGN * max(world_size, 1)
        total = (total + chunk - 1) // chunk * chunk          # shard size stays _ALIGN-aligned
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
    """AdamW (torch.optim.AdamW semantics) + clip_grad_norm_ folded in, on FlatParams; optional sharding."""

    def __init__(self, module: torch.nn.Module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                 max_grad_norm: Optional[float] = 1.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        # parameter order of module.parameters() (what torch.optim.AdamW(model.parameters()) would index): the flat
        # buffer may store them in another order (fused qkv groups)
        self.param_order = [p for p in module.parameters() if p.requires_grad]
        self.flat = FlatParams(module, self.world)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        n = self.flat.numel
        self.shard = n // self.world
        dev = self.flat.data.device
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.gshard = torch.empty(self.shard, dtype=torch.float32, device=dev) if group is not None else None
        self.norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)

    def zero_grad(self):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, lr: Optional[float] = None) -> torch.Tensor:
        """Returns the (device) squared global gradient norm before clipping."""
        self.step_count += 1
        lr = self.lr if lr is None else lr
        lo = self.rank * self.shard
        if self.gshard is not None:
            if dist.get_backend(self.group) == "gloo":
                dist.all_reduce(self.flat.grad, group=self.group)
                self.gshard.copy_(self.flat.grad[lo:lo + self.shard])
            else:
                dist.reduce_scatter_tensor(self.gshard, self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            g = self.gshard
        else:
            g = self.flat.grad
        w = self.flat.data[lo:lo + self.shard]
        _kernels.sumsq(g, out=self.norm_sq)
        if self.group is not None:
            dist.all_reduce(self.norm_sq, group=self.group)
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        _kernels.adamw_step(w, g, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count,
                            grad_norm_sq=self.norm_sq if clip else None,
                            max_norm=self.max_grad_norm if clip else 0.0)
        if self.group is not None:
            dist.all_gather_into_tensor(self.flat.data, w.clone() if dist.get_backend(self.group) == "gloo" else w,
                                        group=self.group)
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
        if dist.get_backend(self.group) == "gloo":
            full = [torch.zeros(self.flat.numel, dtype=torch.float32, device=self.m.device) for _ in range(2)]
            lo = self.rank * self.shard
            full[0][lo:lo + self.shard].copy_(self.m)
            full[1][lo:lo + self.shard].copy_(self.v)
            dist.all_reduce(full[0], group=self.group)
            dist.all_reduce(full[1], group=self.group)
            return full[0], full[1]
        m = torch.empty(self.flat.numel, dtype=torch.float32, device=self.m.device)
        v = torch.empty_like(m)
        dist.all_gather_into_tensor(m, self.m, group=self.group)
        dist.all_gather_into_tensor(v, self.v, group=self.group)
        return m, v

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [..]}
        with cloned, full (un-sharded) per-parameter tensors — loadable by torch.optim.AdamW and by FusedAdamW at any
        world size.  Collective when sharded: call on every rank."""
        m, v = self._gathered_moments()
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        state = {}
        for i, p in enumerate(self.param_order):
            o = off[id(p)]
            st = {}
            if self.step_count > 0:                      # torch creates per-parameter state lazily at the first step
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
        if "param_groups" not in sd:                     # legacy private format
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
        lo, hi = self.rank * self.shard, (self.rank + 1) * self.shard
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, p in enumerate(self.param_order):
            st = sd["state"].get(g["params"][i])
            if not st:
                continue
            steps.add(int(round(float(st["step"]))))
            o, n = off[id(p)], p.numel()
            a, b = max(o, lo), min(o + n, hi)            # the part of this parameter that lives in this rank's shard
            if a >= b:
                continue
            for dst, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq")):
                src = st[key].reshape(-1).to(device=dst.device, dtype=torch.float32)
                dst[a - lo:b - lo].copy_(src[a - o:b - o])
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused step keeps one counter")
        self.step_count = steps.pop() if steps else 0


def cosine_annealing_lr(base_lr: float, epoch: int, t_max: int = 20, eta_min: float = 0.0) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (rna_clip_codes.ipynb:2034, T_max=20)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
