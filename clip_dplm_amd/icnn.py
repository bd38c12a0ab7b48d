"""ICNN transport maps (BASELINE config 5) on the libclipk kernels — inference path.

Mirror of triple_flow/2_icnn_core.py (ConvexLayer :42-127, SingleCellICNN :129-241) and
triple_flow/4_transport_maps.py (TransportCost :46-87, SingleCellTransport :89-145, TripleTransportMaps :147-224,
create_transport_system :248-281): same class names, constructor arguments, state_dict keys.

The transport map is T(x) = dPsi/dx.  The reference obtains it with autograd-of-autograd
(`torch.autograd.grad(y.sum(), x, create_graph=True)`, 2_icnn_core.py:197-201) in forced f32; here it is the
hand-derived input gradient, evaluated with exact-f32 MFMA Linear kernels (clipk_gemm_f32_nt), fused
LayerNorm(+CELU/softplus) forward kernels and their backward kernels (the LayerNorm-backward kernel applies
act'(.) and the normalisation Jacobian in one pass):

    x^ = LN(x);  a1 = W1 x^ + b1;  z1 = act(LN1(a1));  a_k = W_k x^ + b_k + c_k * z_{k-1} softplus(V_k + eps)^T;
    z_k = act(LN_k(a_k));  Psi = w z_K + b
    dz_K = w;  da_k = LNact_bwd(dz_k);  dx^ += da_k W_k;  dz_{k-1} = c_k * da_k softplus(V_k + eps);  T = LN_bwd(dx^)

Scope: eval-mode forward (`model.eval()`), i.e. the map itself.  The reference's train-time
extras — the data-dependent no_grad rescale of the z-contribution (2_icnn_core.py:113-117), the per-row norm clip of
T (:203-209) and back-propagation THROUGH T (double backward) — are not built; calling a module in training mode
raises instead of silently computing something else.  Only the working family of architectures is supported
(hidden_dims[:-1] == input_dim).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import ops


@dataclass
class ICNNConfig:                      # triple_flow/1_config.py:99-121 (fields read by the model code)
    input_dim: int
    hidden_dims: List[int]
    dropout: float = 0.1
    use_time: bool = True
    eps: float = 1e-6
    init_scale: float = 0.01
    gradient_clip: float = 1.0
    hessian_reg: float = 1e-4
    use_layer_norm: bool = True
    activation: str = "celu"
    weight_decay: float = 1e-5
    sparse_mode: bool = False
    biological_activation: bool = True
    stable_gradient: bool = True


@dataclass
class TransportOutput:                 # 4_transport_maps.py:39-44
    transported: torch.Tensor
    cost: torch.Tensor
    metrics: Optional[Dict[str, float]] = None


def _no_training(m: nn.Module):
    if m.training:
        raise NotImplementedError("clip_dplm_amd.icnn implements the eval-mode transport map; call .eval() "
                                  "(training through T needs double backward — not built, see DESIGN.md)")


class ConvexLayer(nn.Module):
    """2_icnn_core.py:42-127.  Parameters: linear.{weight,bias}, pos_weights, scale, norm.{weight,bias}."""

    def __init__(self, input_dim: int, output_dim: int, config: ICNNConfig):
        super().__init__()
        self.config = config
        self.linear = nn.Linear(input_dim, output_dim)
        self.pos_weights = nn.Parameter(torch.zeros(output_dim, input_dim))
        self.scale = nn.Parameter(torch.ones(1) * config.init_scale)
        self.norm = nn.LayerNorm(output_dim) if config.use_layer_norm else nn.Identity()
        nn.init.orthogonal_(self.linear.weight)
        bound = 1 / (input_dim ** 0.5)
        nn.init.uniform_(self.linear.bias, -bound, bound)

    def get_positive_weights(self) -> torch.Tensor:
        """softplus(W+ + eps) (:84-86); the eps shift is a [out,in] elementwise add (plumbing), softplus a kernel."""
        return ops.act_fwd((self.pos_weights.detach() + self.config.eps).contiguous(), "softplus")


class SingleCellICNN(nn.Module):
    """2_icnn_core.py:129-241."""

    def __init__(self, config: ICNNConfig):
        super().__init__()
        if not config.use_layer_norm:
            raise NotImplementedError("use_layer_norm=False is not built")
        if any(h != config.input_dim for h in config.hidden_dims[:-1]):
            raise ValueError("the reference only runs when hidden_dims[:-1] == input_dim (SURVEY App. A-11)")
        self.config = config
        self.input_norm = nn.LayerNorm(config.input_dim)
        self.layers = nn.ModuleList()
        prev = config.input_dim
        for h in config.hidden_dims:
            self.layers.append(ConvexLayer(prev, h, config))
            prev = h
        self.final = nn.Linear(prev, 1)

    # -- shared forward pass: returns Psi [B,1] and what the input gradient needs
    @torch.no_grad()
    def _forward(self, x: torch.Tensor):
        act = self.config.activation if self.config.activation == "softplus" else "celu"
        x = x.contiguous().float()
        xh, _, m0, r0 = ops.layernorm_fwd(x, self.input_norm.weight, self.input_norm.bias, self.input_norm.eps)
        saved = []
        z = None
        for layer in self.layers:
            if z is None:
                a = ops.gemm_f32_nt(xh, layer.linear.weight, bias=layer.linear.bias)
                pw = None
            else:
                pw = layer.get_positive_weights()
                zc = ops.gemm_f32_nt(z, pw)                                   # F.linear(z, pos_w)
                a = ops.gemm_f32_nt(xh, layer.linear.weight, bias=layer.linear.bias, addend=zc,
                                    addend_scale=layer.scale)
            z, _, m, r = ops.layernorm_fwd(a, layer.norm.weight, layer.norm.bias, layer.norm.eps, act=act)
            saved.append((a, m, r, pw))
        psi = z @ self.final.weight.t() + self.final.bias                     # [B,1] dot with one row: plumbing
        return psi, (x, xh, m0, r0, saved, act)

    def forward(self, x: torch.Tensor, return_intermediates: bool = False):
        _no_training(self)
        psi, _ = self._forward(x)
        return psi, None

    @torch.no_grad()
    def gradient(self, x: torch.Tensor, create_graph: bool = True) -> torch.Tensor:
        """T(x) = dPsi/dx (eval mode: no norm clip)."""
        _no_training(self)
        _, (x, xh, m0, r0, saved, act) = self._forward(x)
        B = x.shape[0]
        dz = self.final.weight.detach().expand(B, -1).contiguous()           # d Psi / d z_K = w
        dxh = None
        for layer, (a, m, r, pw) in zip(reversed(self.layers), reversed(saved)):
            da, _, _, _ = ops.layernorm_bwd(dz, a, layer.norm.weight, layer.norm.bias, m, r, act=act)
            wt = layer.linear.weight.detach().t().contiguous()                # [in, out]: da @ W = gemm_nt(da, W^T)
            dxh = ops.gemm_f32_nt(da, wt, addend=dxh)
            if pw is not None:
                dz = ops.gemm_f32_nt(da, pw.t().contiguous())
                dz = dz * layer.scale.detach()                                # scalar scale: plumbing
        t, _, _, _ = ops.layernorm_bwd(dxh, x, self.input_norm.weight, None, m0, r0)
        return t


class TransportCost(nn.Module):
    """4_transport_maps.py:46-87 (forward value only): mean ||s - t||_2 + reg * (mean ||s||_1 + mean ||t||_1)."""

    def __init__(self, regularization: float = 0.01):
        super().__init__()
        self.regularization = regularization

    @torch.no_grad()
    def forward(self, source: torch.Tensor, target: torch.Tensor):
        _, n = ops.l2norm_fwd((source - target).contiguous())                # row L2 norms from the normalise kernel
        w2 = n.mean()
        sparsity = self.regularization * (source.abs().sum(-1).mean() + target.abs().sum(-1).mean())
        return w2 + sparsity, {"w2_cost": w2.item(), "sparsity_cost": sparsity.item()}


class SingleCellTransport(nn.Module):
    """4_transport_maps.py:89-145: LN_in -> ICNN.gradient -> LN_out."""

    def __init__(self, input_dim: int, output_dim: int, config: ICNNConfig):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.transport_net = SingleCellICNN(config)
        self.cost_fn = TransportCost()
        self.input_norm = nn.LayerNorm(input_dim)
        self.output_norm = nn.LayerNorm(output_dim)

    @torch.no_grad()
    def forward(self, source: torch.Tensor, target: Optional[torch.Tensor] = None):
        _no_training(self)
        s, _, _, _ = ops.layernorm_fwd(source.contiguous().float(), self.input_norm.weight, self.input_norm.bias,
                                       self.input_norm.eps, want_stats=False)
        t = self.transport_net.gradient(s)
        out, _, _, _ = ops.layernorm_fwd(t, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps,
                                         want_stats=False)
        return out

    @torch.no_grad()
    def cost(self, source: torch.Tensor, target: torch.Tensor) -> TransportOutput:
        """The value the reference's training branch reports (:135-143), without the train-only ICNN tweaks."""
        transported = self.forward(source)
        tgt, _, _, _ = ops.layernorm_fwd(target.contiguous().float(), self.output_norm.weight, self.output_norm.bias,
                                         self.output_norm.eps, want_stats=False)
        c, metrics = self.cost_fn(transported, tgt)
        return TransportOutput(transported=transported, cost=c, metrics=metrics)


class TripleTransportMaps(nn.Module):
    """4_transport_maps.py:147-224 (eval): the three maps; the reference's ConsistencyChecker calls a tensor
    (App. A-12) and only runs in training, so it is not part of this path."""

    def __init__(self, cell_dim: int, pert_dim: int, protein_dim: int, config: ICNNConfig):
        super().__init__()
        self.cell_to_pert = SingleCellTransport(cell_dim, pert_dim, config)
        self.cell_to_protein = SingleCellTransport(cell_dim, protein_dim, config)
        self.pert_to_protein = SingleCellTransport(pert_dim, protein_dim, config)

    def forward(self, cell_states, pert_states=None, protein_states=None) -> Dict[str, Union[torch.Tensor, TransportOutput]]:
        _no_training(self)
        out = {}
        if pert_states is not None:
            out["cell_to_pert"] = self.cell_to_pert(cell_states, pert_states)
        if protein_states is not None:
            out["cell_to_protein"] = self.cell_to_protein(cell_states, protein_states)
        if pert_states is not None and protein_states is not None:
            out["pert_to_protein"] = self.pert_to_protein(pert_states, protein_states)
        return out


def create_transport_system(cell_dim: int, pert_dim: int, protein_dim: int, hidden_dims: Optional[List[int]] = None,
                            **kwargs) -> TripleTransportMaps:
    """4_transport_maps.py:248-281."""
    top = max(cell_dim, pert_dim, protein_dim)
    if hidden_dims is None:
        hidden_dims = [top, top // 2]
    return TripleTransportMaps(cell_dim, pert_dim, protein_dim, ICNNConfig(input_dim=top, hidden_dims=hidden_dims, **kwargs))
