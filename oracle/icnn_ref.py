"""CPU restatement (PyTorch f32) of the ICNN transport map, eval mode.  Test infrastructure only.

triple_flow/2_icnn_core.py:88-127 (ConvexLayer.forward), :156-179 (SingleCellICNN.forward), :181-211 (gradient: the
transport map is autograd's d Psi / d x), triple_flow/4_transport_maps.py:113-145 (LN -> T -> LN) and :59-87 (cost).
Weights keyed by the reference's state_dict names.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .clip_ref import _linear, _ln

SD = Dict[str, torch.Tensor]


def icnn_potential(x, sd: SD, prefix: str, n_layers: int, eps_w: float = 1e-6, activation: str = "celu"):
    xh = _ln(x, sd, f"{prefix}.input_norm", 1e-5)
    z = None
    for k in range(n_layers):
        p = f"{prefix}.layers.{k}"
        y = _linear(xh, sd, f"{p}.linear")
        if z is not None:
            pos_w = F.softplus(sd[f"{p}.pos_weights"] + eps_w)
            y = y + (z @ pos_w.t()) * sd[f"{p}.scale"]
        y = _ln(y, sd, f"{p}.norm", 1e-5)
        z = F.softplus(y) if activation == "softplus" else F.celu(y)
    return _linear(z, sd, f"{prefix}.final")


def icnn_gradient(x, sd: SD, prefix: str, n_layers: int, **kw):
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        psi = icnn_potential(xr, sd, prefix, n_layers, **kw)
        g, = torch.autograd.grad(psi.sum(), xr)
    return g


def single_cell_transport(source, sd: SD, prefix: str, n_layers: int, **kw):
    s = _ln(source, sd, f"{prefix}.input_norm", 1e-5)
    t = icnn_gradient(s, sd, f"{prefix}.transport_net", n_layers, **kw)
    return _ln(t, sd, f"{prefix}.output_norm", 1e-5)


def transport_cost(transported, target_normed, regularization: float = 0.01):
    w2 = (transported - target_normed).norm(dim=-1).mean()
    sp = regularization * (transported.abs().sum(-1).mean() + target_normed.abs().sum(-1).mean())
    return w2 + sp, w2, sp
