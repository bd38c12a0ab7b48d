"""The step either side of the hot path: learning-rate schedule, early stopping, epoch loops and
checkpoints, mirroring the reference's training code so that its scripts run on FusedAdamW unchanged:

  * CosineAnnealingLR          torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=20)
                               (current/rna_clip_codes.ipynb:2034) for the flat fused optimiser, same state_dict keys;
  * EarlyStopping              current/rna_clip_codes.ipynb:2002-2027 (patience / min_delta, same return protocol);
  * train_epoch / evaluate_model   ibid. :2040-2089 (zero_grad, loss.backward, clip_grad_norm_(1.0), step) — the clip is
                               folded into FusedAdamW.step(), so there is no separate clip call;
  * save_checkpoint / load_checkpoint   triple_flow/5_training.py:335-358: the same dict keys ('model_state',
                               'optimizer_state', 'scheduler_state', 'training_state', 'config'); optimizer_state is in
                               torch.optim.AdamW's own format (FusedAdamW.state_dict), so checkpoints move between the
                               reference's torch optimiser and this one in both directions.
Host-side control flow only: every tensor operation happens inside the model / optimiser kernels.
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Iterable, Optional

import torch

from . import functional as KF
from . import ops


class CosineAnnealingLR:
    """eta_t = eta_min + (base_lr - eta_min) (1 + cos(pi t / T_max)) / 2 (the closed form torch documents; equal to its
    recursive update up to rounding).  `optimizer` needs an `lr` attribute (FusedAdamW) or torch-style param_groups."""

    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0, last_epoch: int = -1):
        self.optimizer, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lrs = [self._get_lr()]
        self.last_epoch = last_epoch
        self._step_count = 0
        self.step()

    def _get_lr(self) -> float:
        return self.optimizer.lr if hasattr(self.optimizer, "lr") else self.optimizer.param_groups[0]["lr"]

    def _set_lr(self, lr: float) -> None:
        if hasattr(self.optimizer, "lr"):
            self.optimizer.lr = lr
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = lr

    def step(self) -> None:
        self._step_count += 1
        self.last_epoch += 1
        lr = self.eta_min + (self.base_lrs[0] - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
        self._set_lr(lr)
        self._last_lr = [lr]

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self) -> Dict[str, Any]:
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": list(self.base_lrs),
                "last_epoch": self.last_epoch, "_step_count": self._step_count, "_last_lr": list(self._last_lr)}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.T_max, self.eta_min = sd["T_max"], sd["eta_min"]
        self.base_lrs = list(sd["base_lrs"])
        self.last_epoch, self._step_count = sd["last_epoch"], sd.get("_step_count", sd["last_epoch"] + 1)
        self._last_lr = list(sd.get("_last_lr", self.base_lrs))
        self._set_lr(self._last_lr[0])


class EarlyStopping:
    """rna_clip_codes.ipynb:2002-2027.  __call__(val_loss) returns True when the loss improved by more than min_delta
    (counter reset), False otherwise; `early_stop` is set once `patience` non-improving calls have accumulated."""

    def __init__(self, patience: int = 5, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self.counter = 0
        self.best_loss: Optional[float] = None
        self.early_stop = False

    def __call__(self, val_loss: float) -> bool:
        if self.best_loss is None:
            self.best_loss = val_loss
            return False
        if val_loss > self.best_loss - self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
            return False
        self.best_loss = val_loss
        self.counter = 0
        return True


def _loss_of(out):
    if isinstance(out, dict):
        return out["loss"]
    if isinstance(out, (tuple, list)):
        return out[-1]                                     # (rna_embed, rbp_embed, loss), rna_clip_codes.ipynb:1954
    return out


def train_epoch(model, loader: Iterable, optimizer, device=None, step_fn: Optional[Callable] = None) -> float:
    """One epoch of rna_clip_codes.ipynb:2061-2089.  `optimizer` = FusedAdamW(max_grad_norm=1.0): its step() clips by the
    global norm and applies AdamW in two kernels.  step_fn(model, batch) -> loss overrides `model(*batch)`."""
    model.train()
    total, n = 0.0, 0
    for batch in loader:
        batch = tuple(b.to(device) if (device is not None and torch.is_tensor(b)) else b for b in batch)
        optimizer.zero_grad()
        loss = step_fn(model, batch) if step_fn is not None else _loss_of(model(*batch))
        loss.backward()
        optimizer.step()
        total += float(loss.item())
        n += 1
    return total / max(n, 1)


@torch.no_grad()
def evaluate_model(model, loader: Iterable, device=None, step_fn: Optional[Callable] = None) -> float:
    """rna_clip_codes.ipynb:2040-2059."""
    model.eval()
    total, n = 0.0, 0
    for batch in loader:
        batch = tuple(b.to(device) if (device is not None and torch.is_tensor(b)) else b for b in batch)
        loss = step_fn(model, batch) if step_fn is not None else _loss_of(model(*batch))
        total += float(loss.item())
        n += 1
    return total / max(n, 1)


def save_checkpoint(path, model, optimizer, scheduler=None, training_state: Optional[dict] = None,
                    config: Optional[dict] = None) -> None:
    """triple_flow/5_training.py:335-347: same keys.  Collective when the optimiser is sharded (call on every rank; write
    from rank 0)."""
    ckpt = {"model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(),
            "scheduler_state": scheduler.state_dict() if scheduler is not None else None,
            "training_state": dict(training_state or {}), "config": dict(config or {})}
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
        torch.save(ckpt, path)


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location="cpu") -> dict:
    """triple_flow/5_training.py:349-358.  weights_only=True: tensors and plain containers only, nothing is executed."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ckpt["model_state"])
    if optimizer is not None and ckpt.get("optimizer_state") is not None:
        optimizer.load_state_dict(ckpt["optimizer_state"])
    if scheduler is not None and ckpt.get("scheduler_state") is not None:
        scheduler.load_state_dict(ckpt["scheduler_state"])
    return ckpt


class GraphedTrainStep:
    """One whole training step - zero_grad -> loss_fn(*inputs) -> backward -> clip + AdamW (the loop body of
    rna_clip_codes.ipynb:2061-2089) - captured ONCE in a hipGraph and replayed.

    For the models that pool one position (RNARBPCLIPModel, ContrastiveModel) the sliced step is ~350 launches of
    microsecond kernels over 32 rows: eager, the host cannot issue them as fast as the GPU retires them (7 ms per step
    on MI355X of which ~3 ms are kernel time).  Replayed from a graph the step costs its kernels.  What changes between
    steps lives in device memory: the inputs (static buffers, copied in before the replay) and AdamW's learning rate and
    bias corrections (`FusedAdamW.prepare_step`).

        step = GraphedTrainStep(model, opt, lambda rna, rbp: model(rna, rbp)[2], (rna, rbp))
        for rna, rbp in loader: loss = step(rna, rbp)           # same shapes as the example inputs

    Dropout (nn.TransformerEncoderLayer's 0.1 in train() mode, ipynb:1915): the kernels' masks are counter-based,
    keep(seed, element index), and a captured launch carries its host-drawn seed as a constant - so the step registers a
    device word (`ops.set_dropout_epoch`) that every dropout site adds to its seed when the kernel runs, and increments it
    at the end of the captured step: replay k drops with seed + k * 0x9E3779B9, its backward re-draws the same masks.
    torch's own `nn.Dropout` modules (the projection heads') are replay-safe by themselves: torch's device generator hands a
    captured kernel its Philox offset through device memory and advances it per replay.

    Inputs of another shape than the example (the short last batch of an epoch) run the same step eagerly (`eager_step`).
    Restrictions: single process (no collectives).  The returned loss of a replay is a static device tensor overwritten by
    the next replay (`.item()` / `.clone()` it to keep it)."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, warmup: int = 3):
        if getattr(optimizer, "group", None) is not None:
            raise ValueError("GraphedTrainStep captures single-process steps only")
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.static_in = [t.detach().clone().contiguous() for t in example_inputs]
        optimizer.enable_device_hyper()
        dev = optimizer.flat.data.device
        self.drop_epoch = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        # the warm-up steps are real optimiser steps on the example inputs: put weights and optimiser state back afterwards
        saved = (optimizer.flat.data.clone(), optimizer.m.clone(), optimizer.v.clone(), optimizer.step_count)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up steps: workspaces, kernel attributes, .grad views
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer._graph_body = True
        KF.set_capture_force(True)                         # bf16 weight copies: rebuilt at their point of use, in the graph
        ops.set_dropout_epoch(self.drop_epoch)             # launches captured from here on read their seed offset from it
        try:
            optimizer.prepare_step()                       # the captured step is a real one, too
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_loss = self._body()
                if self.drop_epoch is not None:
                    self.drop_epoch.add_(1)                # after the backward: the next replay draws new masks
        finally:
            optimizer._graph_body = False
            KF.set_capture_force(False)
            ops.set_dropout_epoch(None)                    # eager launches elsewhere keep their plain seeds
        if self.drop_epoch is not None:
            self.drop_epoch.zero_()
        optimizer.flat.data.copy_(saved[0])
        optimizer.m.copy_(saved[1])
        optimizer.v.copy_(saved[2])
        optimizer.step_count = saved[3]
        KF.mark_weights_dirty()                            # the bf16 copies the first replay reads must be those of the
        if optimizer.flat.data.is_cuda:                    # RESTORED weights (later replays refresh them in the graph)
            KF.refresh_weight_caches()

    def _body(self):
        self.opt.zero_grad()
        loss = self.loss_fn(*self.static_in)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _eager(self):
        self.opt.zero_grad()
        loss = self.loss_fn(*self.static_in)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def eager_step(self, *inputs, lr=None):
        """The same training step issued eagerly on `inputs` of any shape (what __call__ falls back to)."""
        KF.mark_weights_dirty()        # replays changed the weights behind the host's back: per-call bf16 copies are rebuilt
        self.opt.zero_grad()
        loss = self.loss_fn(*inputs)
        loss.backward()
        self.opt.step(lr=lr)
        return loss.detach()

    def __call__(self, *inputs, lr=None):
        if any(tuple(src.shape) != tuple(dst.shape) for dst, src in zip(self.static_in, inputs)):
            # another shape than the captured one - the short last batch of an epoch (rna_clip_codes.ipynb:2061-2089 iterates
            # a DataLoader without drop_last): the same step, issued eagerly; optimiser state and step count carry on
            return self.eager_step(*inputs, lr=lr)
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        self.opt.prepare_step(lr)
        self.graph.replay()
        return self.static_loss
