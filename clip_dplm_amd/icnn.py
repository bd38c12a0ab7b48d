"""ICNN transport maps (BASELINE config 5) on the libclipk kernels.

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

Two paths, selected by `module.training`:
  * eval (the map itself): everything above on hand-written kernels, no autograd graph;
  * train: the reference's training branch — data-dependent no_grad rescale of the z contribution
    (2_icnn_core.py:113-117; the rescaled contribution is a constant for autograd there, reproduced here), per-row
    norm clip of T (:203-209), cost on the normalised target (4_transport_maps.py:124-143) — with back-propagation
    THROUGH T, i.e. second derivatives of Psi.  Every matrix product of that path (forward, first and second
    order) runs on the exact-f32 MFMA kernel through `_MatmulNT`, an autograd Function whose backward is built from
    itself and is therefore differentiable again; LayerNorm (+ CELU / softplus) runs on the fused row kernels through
    `_LNActFn` -> `_LNActBwdFn` -> clipk_layernorm_bwd2 (the hand-derived second-order backward, round 3), softplus(W+)
    on the activation kernels; what is left to ATen is scalar-level glue (the `scale` multiply and add, the no_grad
    rescale statistic, the per-row norm clip and the cost's row norms / means) and hessian()'s third derivatives.
Only the working family of architectures is supported (hidden_dims[:-1] == input_dim).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

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


class _MatMul(torch.autograd.Function):
    """C[M, N] = opA(a) @ opB(b) in exact f32 on clipk_gemm_f32 (opA: a or a stored [K, M]; opB: b [N, K] or b stored
    [K, N]).  backward() is written with _MatMul itself — every operand layout is a flag of the kernel, no transposed
    copies — so autograd can differentiate it again: that is what training through T(x) = dPsi/dx needs."""

    @staticmethod
    def forward(ctx, a, b, ta, tb):
        ctx.save_for_backward(a, b)
        ctx.flags = (ta, tb)
        return ops.gemm_f32(a.float(), b.float(), trans_a=ta, trans_b=tb)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb = ctx.flags
        da = db = None
        if ctx.needs_input_grad[0]:
            # d opA(a) = g @ opB(b)^T^T ... in the kernel's terms: contraction over N
            da = _MatMul.apply(b, g, not tb, False) if ta else _MatMul.apply(g, b, False, not tb)
        if ctx.needs_input_grad[1]:
            db = _MatMul.apply(a, g, not ta, True) if tb else _MatMul.apply(g, a, True, not ta)
        return da, db, None, None


class _MatmulNT:
    """a @ b^T (kept under its round-1 name for the call sites below)."""

    @staticmethod
    def apply(a, b):
        return _MatMul.apply(a, b, False, False)


def _linear_f32(x, weight, bias=None):
    y = _MatmulNT.apply(x, weight)
    return y if bias is None else y + bias


class _LNActBwdFn(torch.autograd.Function):
    """First backward of LayerNorm(+activation) as a differentiable op: (dy, a, gamma, beta) -> da on
    clipk_layernorm_bwd; its own backward is the second-order kernel clipk_layernorm_bwd2.  That is what back-propagation
    THROUGH the transport map T(x) = dPsi/dx needs (2_icnn_core.py:197-201 builds T with create_graph=True)."""

    @staticmethod
    def forward(ctx, dy, a, gamma, beta, mean, rstd, act):
        da, _, dg, db = ops.layernorm_bwd(dy.contiguous(), a, gamma, beta, mean, rstd, act=act)
        ctx.save_for_backward(dy, a, gamma, beta, mean, rstd)
        ctx.act = act
        ctx.mark_non_differentiable(dg, db)                  # first-order parameter gradients: leaves of the step
        return da, dg, db

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _gdg, _gdb):
        dy, a, gamma, beta, mean, rstd = ctx.saved_tensors
        d_dy, d_a, d_gamma, d_beta = ops.layernorm_bwd2(g.contiguous(), dy, a, gamma, beta, mean, rstd, act=ctx.act)
        return d_dy, d_a, d_gamma, d_beta, None, None, None


class _LNActFn(torch.autograd.Function):
    """y = act(LayerNorm(a)) on the fused kernel (2_icnn_core.py:121-127: norm then CELU / softplus; act None: the
    input / output LayerNorms of 4_transport_maps.py:103-104).  Differentiable twice: backward = _LNActBwdFn."""

    @staticmethod
    def forward(ctx, a, gamma, beta, eps, act):
        a = a.contiguous().float()
        y, _, mean, rstd = ops.layernorm_fwd(a, gamma, beta, eps, act=act)
        ctx.save_for_backward(a, gamma, beta, mean, rstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        a, gamma, beta, mean, rstd = ctx.saved_tensors
        da, dg, db = _LNActBwdFn.apply(dy, a, gamma, beta, mean, rstd, ctx.act)
        return da, dg, db, None, None


def _layer_norm(mod: nn.LayerNorm, x, act=None):
    return _LNActFn.apply(x, mod.weight, mod.bias, mod.eps, act)


class _ActFn(torch.autograd.Function):
    """Elementwise activation on clipk_act_fwd / clipk_act_bwd: softplus of the positive-weight matrix
    (2_icnn_core.py:84-86).  The weights enter Psi AND T only through the value softplus(W+), so one derivative is all
    the training step asks of it."""

    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.act_fwd(x, act)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), x, ctx.act), None


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

    def forward(self, x: torch.Tensor, z: Optional[torch.Tensor] = None, scale: Optional[float] = None,
                any_order: bool = False) -> torch.Tensor:
        """Differentiable (train-mode) layer, 2_icnn_core.py:88-127.  Matrix products on the exact-f32 MFMA kernel;
        LayerNorm + activation and softplus(W+) on the fused row / elementwise kernels through autograd Functions that
        can be differentiated twice (the training step goes through T = dPsi/dx).  any_order = True keeps the
        ATen composition of the element-wise part (hessian(): third derivatives)."""
        act = "softplus" if self.config.activation == "softplus" else "celu"
        kernels = not any_order and self.config.use_layer_norm
        y = _linear_f32(x, self.linear.weight, self.linear.bias)
        if z is not None:
            scale = scale if scale is not None else self.scale
            pw = _ActFn.apply(self.pos_weights + self.config.eps, "softplus") if kernels else \
                F.softplus(self.pos_weights + self.config.eps)
            zc = _linear_f32(z, pw) * scale
            if self.training:
                with torch.no_grad():                     # as in the reference: the rescaled tensor is a constant
                    zs = zc.abs().mean()
                    if zs > self.config.gradient_clip:
                        zc = zc * (self.config.gradient_clip / zs)
            y = y + zc
        if kernels:
            return _layer_norm(self.norm, y, act)
        y = self.norm(y)
        return F.softplus(y) if act == "softplus" else F.celu(y)


class SingleCellICNN(nn.Module):
    """2_icnn_core.py:129-241."""

    def __init__(self, config: ICNNConfig):
        super().__init__()
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
    def _forward(self, x: torch.Tensor, need_psi: bool = True):
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
        # [B,1] dot with one row: plumbing (not needed for the transport map itself)
        psi = z @ self.final.weight.t() + self.final.bias if need_psi else None
        return psi, (x, xh, m0, r0, saved, act)

    def _forward_diff(self, x: torch.Tensor, return_intermediates: bool = False, any_order: bool = False):
        """Differentiable forward, :156-179: matrix products on the exact-f32 MFMA kernel through the re-differentiable
        _MatmulNT; LayerNorm (+ CELU / softplus) on the fused kernels through _LNActFn (differentiable twice: the
        training step).  any_order = True: the element-wise part by ATen, for hessian()'s third derivatives."""
        x = self.input_norm(x.float()) if any_order else _layer_norm(self.input_norm, x.float())
        inter = [] if return_intermediates else None
        z = None
        for layer in self.layers:
            z = layer(x, z, any_order=any_order)
            if return_intermediates:
                inter.append(z)
        return _linear_f32(z, self.final.weight, self.final.bias), inter

    def forward(self, x: torch.Tensor, return_intermediates: bool = False):
        if self.training:
            return self._forward_diff(x, return_intermediates)
        if not self.config.use_layer_norm:                # the fused LN + activation kernels do not apply: same
            with torch.no_grad():                         # products, activation by ATen
                return self._forward_diff(x, any_order=True)[0], None
        psi, _ = self._forward(x)
        return psi, None

    def gradient(self, x: torch.Tensor, create_graph: bool = True) -> torch.Tensor:
        """T(x) = dPsi/dx.  Eval: hand-derived on kernels, no graph.  Train: autograd.grad with create_graph (:181-211),
        then the per-row norm clip."""
        if self.training:
            with torch.enable_grad():
                if not x.requires_grad:
                    x = x.detach().requires_grad_(True)
                y = self.forward(x)[0]
                grad, = torch.autograd.grad(y.sum(), x, create_graph=create_graph, retain_graph=True)
            gn = grad.norm(dim=-1, keepdim=True)
            return torch.where(gn > self.config.gradient_clip, grad * self.config.gradient_clip / gn, grad)
        if not self.config.use_layer_norm:                # no hand-derived chain without the fused LN kernels
            with torch.enable_grad():
                xr = x.detach().requires_grad_(True)
                grad, = torch.autograd.grad(self._forward_diff(xr, any_order=True)[0].sum(), xr)
            return grad.detach()
        return self._gradient_eval(x)

    @torch.enable_grad()
    def hessian(self, x: torch.Tensor) -> torch.Tensor:
        """H[b, j, i] = d T_i(x_b) / d x_{b,j} (2_icnn_core.py:213-241): one autograd pass over the (create_graph)
        transport map per output coordinate, the reference's convexity check.  Train mode differentiates the
        norm-clipped T of the training branch and adds hessian_reg * I.  Like the reference it keeps the graph
        (create_graph=True) so that a Hessian penalty can be trained through."""
        x = x if x.requires_grad else x.detach().requires_grad_(True)
        y = self._forward_diff(x, any_order=True)[0]
        grad, = torch.autograd.grad(y.sum(), x, create_graph=True, retain_graph=True)
        if self.training:
            gn = grad.norm(dim=-1, keepdim=True)
            grad = torch.where(gn > self.config.gradient_clip, grad * self.config.gradient_clip / gn, grad)
        cols = [torch.autograd.grad(grad[..., i].sum(), x, create_graph=True, retain_graph=True)[0]
                for i in range(grad.shape[-1])]
        hess = torch.stack(cols, dim=-1)
        if self.training:
            hess = hess + self.config.hessian_reg * torch.eye(hess.shape[-1], device=hess.device).expand_as(hess)
        return hess

    @torch.no_grad()
    def _gradient_eval(self, x: torch.Tensor) -> torch.Tensor:
        _, (x, xh, m0, r0, saved, act) = self._forward(x, need_psi=False)
        B = x.shape[0]
        dz = self.final.weight.detach().expand(B, -1).contiguous()           # d Psi / d z_K = w
        dxh = None
        for layer, (a, m, r, pw) in zip(reversed(self.layers), reversed(saved)):
            da, _, _, _ = ops.layernorm_bwd(dz, a, layer.norm.weight, layer.norm.bias, m, r, act=act,
                                            want_param_grads=False)
            # da @ W: the weight [out, in] is the kernel's "B stored [K, N]" layout (no transposed copy)
            dxh = ops.gemm_f32(da, layer.linear.weight.detach(), trans_b=True, addend=dxh)
            if pw is not None:
                dz = ops.gemm_f32(da, pw, trans_b=True, alpha=layer.scale.detach())    # scale_k * da @ softplus(W+)
        t, _, _, _ = ops.layernorm_bwd(dxh, x, self.input_norm.weight, None, m0, r0, want_param_grads=False)
        return t


class TransportCost(nn.Module):
    """4_transport_maps.py:46-87 (forward value only): mean ||s - t||_2 + reg * (mean ||s||_1 + mean ||t||_1)."""

    def __init__(self, regularization: float = 0.01):
        super().__init__()
        self.regularization = regularization

    def forward(self, source: torch.Tensor, target: torch.Tensor):
        if torch.is_grad_enabled() and (source.requires_grad or target.requires_grad):
            w2 = torch.norm(source - target, dim=-1).mean()                   # training: autograd needs the graph
            sparsity = self.regularization * (torch.norm(source, p=1, dim=-1).mean() + torch.norm(target, p=1, dim=-1).mean())
            return w2 + sparsity, {"w2_cost": w2.item(), "sparsity_cost": sparsity.item()}
        return self._forward_value(source, target)

    @torch.no_grad()
    def _forward_value(self, source: torch.Tensor, target: torch.Tensor):
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

    def forward(self, source: torch.Tensor, target: Optional[torch.Tensor] = None):
        if self.training:                                 # 4_transport_maps.py:113-145, differentiable
            transported = _layer_norm(self.output_norm,
                                      self.transport_net.gradient(_layer_norm(self.input_norm, source.float())))
            if target is not None:
                cost, metrics = self.cost_fn(transported, _layer_norm(self.output_norm, target.float()))
                return TransportOutput(transported=transported, cost=cost, metrics=metrics)
            return transported
        return self._forward_eval(source)

    @torch.no_grad()
    def _forward_eval(self, source: torch.Tensor):
        s, _, _, _ = ops.layernorm_fwd(source.contiguous().float(), self.input_norm.weight, self.input_norm.bias,
                                       self.input_norm.eps, want_stats=False)
        t = self.transport_net.gradient(s)
        out, _, _, _ = ops.layernorm_fwd(t, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps,
                                         want_stats=False)
        return out

    @torch.no_grad()
    def cost(self, source: torch.Tensor, target: torch.Tensor) -> TransportOutput:
        """The value the reference's training branch reports (:135-143), without the train-only ICNN tweaks."""
        transported = self._forward_eval(source)
        tgt, _, _, _ = ops.layernorm_fwd(target.contiguous().float(), self.output_norm.weight, self.output_norm.bias,
                                         self.output_norm.eps, want_stats=False)
        c, metrics = self.cost_fn(transported, tgt)
        return TransportOutput(transported=transported, cost=c, metrics=metrics)


class TripleTransportMaps(nn.Module):
    """4_transport_maps.py:147-224: the three maps.  The reference's ConsistencyChecker calls a tensor (App. A-12), so
    its training call with all three modalities raises; the same error is raised here."""

    def __init__(self, cell_dim: int, pert_dim: int, protein_dim: int, config: ICNNConfig):
        super().__init__()
        self.cell_to_pert = SingleCellTransport(cell_dim, pert_dim, config)
        self.cell_to_protein = SingleCellTransport(cell_dim, protein_dim, config)
        self.pert_to_protein = SingleCellTransport(pert_dim, protein_dim, config)
        self.multi_stream = False          # opt-in (eval): the maps on HIP streams of their own (KF.parallel_branches)
        self._streams = None

    def forward(self, cell_states, pert_states=None, protein_states=None) -> Dict[str, Union[torch.Tensor, TransportOutput]]:
        if self.training and pert_states is not None and protein_states is not None:
            # the reference's ConsistencyChecker calls a tensor (`pert_protein(cell_pert)`, :242) and raises TypeError
            raise TypeError("'Tensor' object is not callable  (reference 4_transport_maps.py:242: training with all three "
                            "modalities is broken upstream, SURVEY App. A-12; train two modalities per call)")
        if (self.multi_stream and not self.training and cell_states.is_cuda
                and pert_states is not None and protein_states is not None):
            # eval: three independent chains of ~15 launches each (T(x) of one map only reads its source) - side by side
            # on three streams; replayed from a hipGraph (GraphedTransport) they are three branches of it
            from . import functional as KF
            if self._streams is None:
                self._streams = KF.branch_streams(3)
            a, b, c = KF.parallel_branches(
                self._streams,
                (lambda: self.cell_to_pert(cell_states, pert_states), lambda: self.cell_to_protein(cell_states, protein_states),
                 lambda: self.pert_to_protein(pert_states, protein_states)),
                ((cell_states, pert_states), (cell_states, protein_states), (pert_states, protein_states)))
            return {"cell_to_pert": a, "cell_to_protein": b, "pert_to_protein": c}
        out = {}
        if pert_states is not None:
            out["cell_to_pert"] = self.cell_to_pert(cell_states, pert_states)
        if protein_states is not None:
            out["cell_to_protein"] = self.cell_to_protein(cell_states, protein_states)
        if pert_states is not None and protein_states is not None:
            out["pert_to_protein"] = self.pert_to_protein(pert_states, protein_states)
        return out


class GraphedTransport:
    """Eval-mode transport maps replayed from ONE hipGraph: the op is ~15 small launches per map (two LayerNorms, six
    exact-f32 products, the fused LN + CELU passes), i.e. launch-bound in eager mode.  Static input / output buffers;
    call with tensors of the captured shapes.

        g = GraphedTransport(maps, cell, pert, protein); out = g(cell, pert, protein)     # dict like maps(...)"""

    def __init__(self, maps: "TripleTransportMaps", cell, pert=None, protein=None, warmup: int = 2):
        if maps.training:
            raise ValueError("GraphedTransport captures the eval-mode path (no autograd graph)")
        self.maps = maps
        self.static_in = [None if t is None else t.detach().clone().contiguous() for t in (cell, pert, protein)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream: workspaces, lazy init
            for _ in range(warmup):
                maps(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = maps(*self.static_in)

    @torch.no_grad()
    def __call__(self, cell, pert=None, protein=None):
        for dst, src in zip(self.static_in, (cell, pert, protein)):
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        return self.static_out


def create_transport_system(cell_dim: int, pert_dim: int, protein_dim: int, hidden_dims: Optional[List[int]] = None,
                            **kwargs) -> TripleTransportMaps:
    """4_transport_maps.py:248-281."""
    top = max(cell_dim, pert_dim, protein_dim)
    if hidden_dims is None:
        hidden_dims = [top, top // 2]
    return TripleTransportMaps(cell_dim, pert_dim, protein_dim, ICNNConfig(input_dim=top, hidden_dims=hidden_dims, **kwargs))
