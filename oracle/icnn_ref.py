"""CPU restatement (PyTorch f32) of the ICNN transport map, eval and train mode.  Test infrastructure only.

triple_flow/2_icnn_core.py:88-127 (ConvexLayer.forward), :156-179 (SingleCellICNN.forward), :181-211 (gradient: the
transport map is autograd's d Psi / d x), :213-241 (hessian), triple_flow/4_transport_maps.py:113-145 (LN -> T -> LN)
and :59-87 (cost).
Weights keyed by the reference's state_dict names.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .clip_ref import _linear, _ln

SD = Dict[str, torch.Tensor]


def icnn_potential(x, sd: SD, prefix: str, n_layers: int, eps_w: float = 1e-6, activation: str = "celu",
                   use_layer_norm: bool = True):
    xh = _ln(x, sd, f"{prefix}.input_norm", 1e-5)
    z = None
    for k in range(n_layers):
        p = f"{prefix}.layers.{k}"
        y = _linear(xh, sd, f"{p}.linear")
        if z is not None:
            pos_w = F.softplus(sd[f"{p}.pos_weights"] + eps_w)
            y = y + (z @ pos_w.t()) * sd[f"{p}.scale"]
        if use_layer_norm:                                # 2_icnn_core.py:72: nn.Identity otherwise
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


# ---- train mode: the reference's training branch, 2_icnn_core.py:104-117 (no_grad rescale of the z contribution when
# its mean |.| exceeds gradient_clip), :203-209 (per-row norm clip of T), 4_transport_maps.py:124-143 (cost on the
# normalised target).  `sd` tensors that require grad receive gradients through T (double backward).
def icnn_potential_train(x, sd: SD, prefix: str, n_layers: int, eps_w: float = 1e-6, activation: str = "celu",
                         gradient_clip: float = 1.0):
    xh = _ln(x, sd, f"{prefix}.input_norm", 1e-5)
    z = None
    for k in range(n_layers):
        p = f"{prefix}.layers.{k}"
        y = _linear(xh, sd, f"{p}.linear")
        if z is not None:
            pos_w = F.softplus(sd[f"{p}.pos_weights"] + eps_w)
            zc = (z @ pos_w.t()) * sd[f"{p}.scale"]
            # the reference rescales INSIDE its no_grad block (:113-117): when the branch fires the rescaled
            # contribution is a constant for autograd, i.e. no gradient (first or second order) flows through z here
            with torch.no_grad():
                zs = zc.abs().mean()
                if zs > gradient_clip:
                    zc = zc * (gradient_clip / zs)
            y = y + zc
        y = _ln(y, sd, f"{p}.norm", 1e-5)
        z = F.softplus(y) if activation == "softplus" else F.celu(y)
    return _linear(z, sd, f"{prefix}.final")


def single_cell_transport_train(source, target, sd: SD, prefix: str, n_layers: int, gradient_clip: float = 1.0,
                                regularization: float = 0.01, **kw):
    """Returns (transported, cost, w2, sparsity) with the autograd graph attached (call cost.backward())."""
    with torch.enable_grad():
        s = _ln(source, sd, f"{prefix}.input_norm", 1e-5)
        if not s.requires_grad:
            s = s.requires_grad_(True)
        psi = icnn_potential_train(s, sd, f"{prefix}.transport_net", n_layers, gradient_clip=gradient_clip, **kw)
        g, = torch.autograd.grad(psi.sum(), s, create_graph=True, retain_graph=True)
        gn = g.norm(dim=-1, keepdim=True)
        g = torch.where(gn > gradient_clip, g * gradient_clip / gn, g)
        t = _ln(g, sd, f"{prefix}.output_norm", 1e-5)
        tgt = _ln(target, sd, f"{prefix}.output_norm", 1e-5)
        cost, w2, sp = transport_cost(t, tgt, regularization)
    return t, cost, w2, sp


def icnn_hessian(x, sd: SD, prefix: str, n_layers: int, train: bool = False, gradient_clip: float = 1.0,
                 hessian_reg: float = 1e-4, **kw):
    """2_icnn_core.py:213-241: H[b, j, i] = d T_i(x_b) / d x_{b,j}, one autograd pass per output coordinate.  Train mode
    differentiates the norm-clipped T of the training branch and adds hessian_reg * I."""
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        if train:
            psi = icnn_potential_train(xr, sd, prefix, n_layers, gradient_clip=gradient_clip, **kw)
        else:
            psi = icnn_potential(xr, sd, prefix, n_layers, **kw)
        g, = torch.autograd.grad(psi.sum(), xr, create_graph=True, retain_graph=True)
        if train:
            gn = g.norm(dim=-1, keepdim=True)
            g = torch.where(gn > gradient_clip, g * gradient_clip / gn, g)
        cols = [torch.autograd.grad(g[..., i].sum(), xr, retain_graph=True)[0] for i in range(g.shape[-1])]
    h = torch.stack(cols, dim=-1)
    if train:
        h = h + hessian_reg * torch.eye(h.shape[-1]).expand_as(h)
    return h
