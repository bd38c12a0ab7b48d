"""TEST-ONLY torch/CPU restatement of clip_dplm_amd.ops (same signatures), used to exercise the HOST logic —
hand-written layer backward wiring, flat-buffer optimiser, rank bookkeeping over gloo — in the CPU-only
container.  It is installed by monkeypatching inside tests (see `install`); the product never imports it and has
no CPU path of its own.  Rounds to bf16 where the kernels round, so CPU results track the GPU path closely.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BF = torch.bfloat16


def _act(x, act):
    if act == "relu":
        return torch.relu(x)
    if act == "gelu":
        return F.gelu(x)
    if act == "celu":
        return F.celu(x)
    if act == "softplus":
        return F.softplus(x)
    return x


def _act_grad(x, act):
    if act == "relu":
        return (x > 0).to(x.dtype)
    if act == "gelu":
        cdf = 0.5 * (1 + torch.erf(x / math.sqrt(2)))
        pdf = torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        return cdf + x * pdf
    if act == "celu":
        return torch.where(x > 0, torch.ones_like(x), torch.exp(x))
    if act == "softplus":
        return torch.sigmoid(x)
    return torch.ones_like(x)


def drop_hash32(seed, idx):
    """The kernels' dropout hash (csrc/common.h drop_keep) in torch integer arithmetic: idx int64 >= 0 -> uint32."""
    M = 0xFFFFFFFF
    lo, hi = idx & M, (idx >> 32) & M
    h = (lo * 0x9E3779B1 + hi * 0x85EBCA77 + (int(seed) & M)) & M
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    return h ^ (h >> 16)


_DROP_EPOCH = None


def set_dropout_epoch(epoch):
    """csrc/common.h drop_seed_eff: while a one-element tensor is registered, every site uses seed + epoch * 0x9E3779B9."""
    global _DROP_EPOCH
    _DROP_EPOCH = epoch


def drop_mult(p, seed, idx):
    """keep / (1 - p) multiplier for element indices idx (any shape, int64)."""
    if _DROP_EPOCH is not None:
        seed = (int(seed) + (int(_DROP_EPOCH.item()) & 0xFFFFFFFF) * 0x9E3779B9) & 0xFFFFFFFF
    t = p * 4294967296.0
    thr = 1 if t < 1.0 else (4294967295 if t >= 4294967295.0 else int(t))
    return (drop_hash32(seed, idx) >= thr).float() / (1.0 - p)


GD8_MIN, GD8_STEP = -0.13, 0.005          # code 26 <-> GELU' = 0, code 226 <-> GELU' = 1 (csrc/common.h)


def gelu_grad_code(x):
    """clipk.h aux_dtype = U8: GELU'(x) -> nearest of 256 levels -0.13 + 0.005 k."""
    return torch.clamp(torch.floor(_act_grad(x, "gelu") * 200.0 + 26.0 + 0.5), 0, 255).to(torch.uint8)


def gemm_nt(a, b, bias=None, act=None, out_dtype=BF, residual=None, out_preact=False, dact_aux=None, dact=None,
            alpha=1.0, out=None, dropout=None, rope=None, aux_u8=False, rope_interleaved=False):
    aux8 = bool(aux_u8 and out_preact) or (dact_aux is not None and dact_aux.dtype == torch.uint8)
    if aux8 and (a.shape[1] % 32 or b.shape[0] % 8):      # as the kernels: generic-K GEMM has no 8-bit aux (gemm_nt.hip)
        raise RuntimeError("clipk_gemm_nt failed: unsupported shape for the gfx950 kernels (emulated: u8 aux needs "
                           "K % 32 == 0 and N % 8 == 0)")
    v = (a.float() @ b.float().t()) * alpha
    if bias is not None:
        v = v + bias
    if rope is not None and rope_interleaved:              # neighbouring pairs (2 j, 2 j + 1) share angle j
        cos, sin, L, hd, cols = rope
        M_ = v.shape[0]
        pos = torch.arange(M_) % L
        x = v[:, :cols].reshape(M_, cols // hd, hd // 2, 2)
        c, s_ = cos[pos][:, None, :], sin[pos][:, None, :]
        y = torch.stack([x[..., 0] * c - x[..., 1] * s_, x[..., 1] * c + x[..., 0] * s_], -1)
        v = torch.cat([y.reshape(M_, cols), v[:, cols:]], 1)
    elif rope is not None:                                # rotate-half on the first `cols` columns, heads of hd columns
        cos, sin, L, hd, cols = rope
        M_ = v.shape[0]
        pos = torch.arange(M_) % L
        c = torch.cat([cos[pos], cos[pos]], -1)[:, None, :]            # [M, 1, hd]
        s_ = torch.cat([sin[pos], sin[pos]], -1)[:, None, :]
        x = v[:, :cols].reshape(M_, cols // hd, hd)
        rot = torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
        v = torch.cat([(x * c + rot * s_).reshape(M_, cols), v[:, cols:]], 1)
    pre = (gelu_grad_code(v) if aux_u8 else v.to(BF)) if out_preact else None
    v = _act(v, act)
    if dropout:
        M_, N_ = v.shape
        v = v * drop_mult(dropout[0], dropout[1], torch.arange(M_ * N_, dtype=torch.int64).view(M_, N_))
    if dact_aux is not None and dact_aux.dtype == torch.uint8:
        v = v * (dact_aux.float() * GD8_STEP + GD8_MIN)
    elif dact_aux is not None:
        v = v * _act_grad(dact_aux.float(), dact)
    if residual is not None:
        v = v + residual.float()
    c = v.to(out.dtype if out is not None else out_dtype)
    if out is not None:
        out.copy_(c)
        c = out
    return (c, pre) if out_preact else c


def gelu_aux_u8_applies(k_in, n_ffn):
    return k_in % 32 == 0 and n_ffn % 8 == 0


def gemm_wgrad(dy, x, dw=None, dbias=None, accumulate=False, want_bias=False, il=(0, 0)):
    g = dy.float().t() @ x.float()
    b = dy.float().sum(0)
    if il[1]:                                              # rows of the permuted order -> rows of the original order
        src = il_source_rows(g.shape[0], il[0], il[1])
        g2, b2 = torch.empty_like(g), torch.empty_like(b)
        g2[src], b2[src] = g, b
        g, b = g2, b2
    if dw is None:
        dw = g
    else:
        dw.copy_(dw + g if accumulate else g)
    if want_bias or dbias is not None:
        if dbias is None:
            dbias = b
        else:
            dbias.copy_(dbias + b if accumulate else b)
    return dw, dbias


def _scores(x, y, scale, cache):
    keys = y if cache is None else torch.cat([y, cache], 0)
    return (x.double() @ keys.double().t()) * scale.double()


def simce_lse(x, y, scale, label_offset=0, cache=None):
    s = _scores(x, y, scale, cache)
    idx = torch.arange(x.shape[0]) + label_offset
    return torch.logsumexp(s, 1).float(), s[torch.arange(x.shape[0]), idx].float()


def ce_combine(lse_r, pos_r, lse_c, pos_c, w_row, w_col, bg):
    local = w_row * (lse_r - pos_r).sum()
    if lse_c is not None:
        local = local + w_col * (lse_c - pos_c).sum()
    return local / bg


def simce_grad(x, y, scale, lse_x, lse_y, w_row, w_col, inv_bg, label_offset=0, cache=None, upstream=None):
    if upstream is not None:
        inv_bg = inv_bg * float(upstream.reshape(-1)[0])
    s = _scores(x, y, scale, cache)
    ny = y.shape[0]
    g = w_row * torch.exp(s - lse_x.double()[:, None])
    g[:, :ny] += w_col * torch.exp(s[:, :ny] - lse_y.double()[None, :])
    idx = torch.arange(x.shape[0]) + label_offset
    g[torch.arange(x.shape[0]), idx] -= (w_row + w_col)
    g = g * inv_bg
    keys = y if cache is None else torch.cat([y, cache], 0)
    dx = (g @ keys.double()) * scale.double()
    dsc = (g * (s / scale.double())).sum(1)
    return dx.float(), dsc.float()


def simce_lse_pairs(E, pairs, scale):
    out = [simce_lse(E[a], E[b], scale) for a, b in pairs]
    return torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out])


def simce_grad_pairs(E, pairs, scale, lse, w_row, w_col, inv_bg):
    pairs = list(pairs)
    out = []
    for i, (a, b) in enumerate(pairs):
        r = pairs.index((b, a))
        out.append(simce_grad(E[a], E[b], scale, lse[i], lse[r], w_row, w_col, inv_bg))
    return torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out])


def gemm_f32_nt(x, w, bias=None, addend=None, addend_scale=None):
    out = x @ w.t()
    if bias is not None:
        out = out + bias
    if addend is not None:
        out = out + (addend_scale if addend_scale is not None else 1.0) * addend
    return out


def sim_logits(x, y, scale):
    return (x @ y.t()) * scale


def matmul_f32_nt(a, b):
    return a @ b.t()


def colsum_f32(x, out=None, accumulate=False):
    s_ = x.sum(0)
    if out is None:
        return s_
    out.copy_(out + s_ if accumulate else s_)
    return out


def gemm_wgrad_f32(dy, x, dw=None, dbias=None, accumulate=False, want_bias=True):
    w_ = dy.t() @ x
    b_ = dy.sum(0)
    if dw is None:
        dw = w_
    else:
        dw.copy_(dw + w_ if accumulate else w_)
    if dbias is None:
        dbias = b_ if want_bias else None
    else:
        dbias.copy_(dbias + b_ if accumulate else b_)
    return dw, dbias


def gemm_f32(a, b, trans_a=False, trans_b=False, bias=None, addend=None, addend_scale=None, alpha=None, out=None):
    r = (a.t() if trans_a else a) @ (b if trans_b else b.t())
    if alpha is not None:
        r = r * alpha
    if bias is not None:
        r = r + bias
    if addend is not None:
        r = r + (addend_scale if addend_scale is not None else 1.0) * addend
    if out is not None:
        out.copy_(r)
        return out
    return r


def transpose_scale_f32(x, scale=None):
    return (x.t() * (scale if scale is not None else 1.0)).contiguous()


def ce_logits_lse(S, S2=None, columns=False, label_offset=0):
    if columns:
        return torch.logsumexp(S, 0), torch.diagonal(S, -label_offset).clone()
    full = S if S2 is None else torch.cat([S, S2], 1)
    return torch.logsumexp(full, 1), torch.diagonal(S, label_offset).clone()


def ce_logits_bwd(S, S2, lse_row, lse_col, w_row, w_col, g, off_row=0, off_col=0):
    eye = torch.eye(S.shape[0], S.shape[1], dtype=S.dtype)
    d = torch.zeros_like(S)
    d2 = None
    if lse_row is not None:
        d = d + w_row * (torch.exp(S - lse_row[:, None]) - eye)
        if S2 is not None:
            d2 = g * w_row * torch.exp(S2 - lse_row[:, None])
    if lse_col is not None:
        d = d + w_col * (torch.exp(S - lse_col[None, :]) - eye)
    return g * d, d2


def layernorm_fwd(x, gamma, beta, eps, act=None, want_f32=True, want_bf16=False, want_stats=True):
    xf = x.float()
    mean = xf.mean(-1)
    var = ((xf - mean[:, None]) ** 2).mean(-1)
    rstd = torch.rsqrt(var + eps)
    y = _act((xf - mean[:, None]) * rstd[:, None] * gamma + beta, act)
    return (y if want_f32 else None), (y.to(BF) if want_bf16 else None), mean, rstd


def layernorm_bwd_partial_shape(rows, cols):
    return 1, 2 * cols


def colreduce_entries(entries):
    for part, blocks, cols, dg, db in entries:
        p = part.view(-1)[:blocks * 2 * cols].view(blocks, 2 * cols).sum(0)
        dg.add_(p[:cols])
        db.add_(p[cols:])


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, act=None, dx_add=None, want_f32=True, want_bf16=False, want_param_grads=True,
                  dgamma=None, dbeta=None, accumulate=False, dropout_bf16=None, part_out=None):
    xf, dyf = x.float(), dy.float()
    xh = (xf - mean[:, None]) * rstd[:, None]
    if act is not None:
        dyf = dyf * _act_grad(xh * gamma + beta, act)
    g = dyf * gamma
    c1 = g.mean(-1, keepdim=True)
    c2 = (g * xh).mean(-1, keepdim=True)
    dx = rstd[:, None] * (g - c1 - xh * c2)
    if dx_add is not None:
        dx = dx + dx_add.float()
    dg, db = (dyf * xh).sum(0), dyf.sum(0)
    if part_out is not None:                               # one partial row [dgamma | dbeta], reduced later
        part_out.view(-1)[:2 * dg.numel()].copy_(torch.cat([dg, db]))
        dg = db = None
    if dgamma is not None:
        dgamma.copy_(dgamma + dg if accumulate else dg)
        dbeta.copy_(dbeta + db if accumulate else db)
        dg, db = dgamma, dbeta
    dxb = None
    if want_bf16:
        dxb = dx
        if dropout_bf16:
            R, Cc = dx.shape
            dxb = dx * drop_mult(dropout_bf16[0], dropout_bf16[1], torch.arange(R * Cc, dtype=torch.int64).view(R, Cc))
        dxb = dxb.to(BF)
    return (dx if want_f32 else None), dxb, dg, db


def layernorm_bwd2(g, dy, a, gamma, beta, mean, rstd, act=None, dgamma=None, dbeta=None, accumulate=False):
    """Backward of the LayerNorm(+act) backward by torch autograd over a torch restatement of the first backward
    (statistics recomputed from `a`, as the kernel's derivation assumes)."""
    eps_free = True                                        # mean / rstd are functions of a: recompute rstd from var
    with torch.enable_grad():
        dyr, ar = dy.detach().clone().requires_grad_(True), a.detach().clone().requires_grad_(True)
        gr = gamma.detach().clone().requires_grad_(True)
        br = (beta.detach().clone().requires_grad_(True)) if beta is not None else None
        mu = ar.mean(-1, keepdim=True)
        var = ((ar - mu) ** 2).mean(-1, keepdim=True)
        # rstd = (var + eps)^-1/2 with the eps the forward used: recover it from the saved rstd of the same rows
        eps = (1.0 / rstd.double() ** 2 - var.detach().double().squeeze(-1)).mean().clamp_min(0).float()
        r = torch.rsqrt(var + eps)
        xh = (ar - mu) * r
        u = dyr * gr
        if act is not None:
            n = xh * gr + br
            u = u * torch.autograd.grad(_act(n, act).sum(), n, create_graph=True)[0]
        da = r * (u - u.mean(-1, keepdim=True) - xh * (u * xh).mean(-1, keepdim=True))
        ins = [dyr, ar, gr] + ([br] if (br is not None and act is not None) else [])
        outs = torch.autograd.grad(da, ins, g, allow_unused=True)
    d_dy, d_a, dg = outs[0], outs[1], outs[2]
    db = outs[3] if len(outs) > 3 and outs[3] is not None else torch.zeros_like(gamma)
    if dgamma is not None:
        dgamma.copy_(dgamma + dg if accumulate else dg)
        dbeta.copy_(dbeta + db if accumulate else db)
        dg, db = dgamma, dbeta
    return d_dy, d_a, dg, db


def meanpool_fused_supported(cols):
    return cols % 4 == 0 and (4 * cols + 4) * 4 <= 65536


def layernorm_meanpool_fwd(x, gamma, beta, eps, B, L, mask=None):
    y, _, mean, rstd = layernorm_fwd(x, gamma, beta, eps)
    m = torch.ones(B * L) if mask is None else mask.view(-1).float()
    n = m.view(B, L).sum(1)
    inv = torch.where(n > 0, 1.0 / n.clamp_min(1.0), torch.zeros_like(n))
    pooled = (y * m[:, None]).view(B, L, -1).sum(1) * inv[:, None]
    return pooled, mean, rstd, m * inv.repeat_interleave(L)


def layernorm_meanpool_bwd(dpooled, row_weight, x, gamma, mean, rstd, B, L, want_f32=True, want_bf16=False,
                           dgamma=None, dbeta=None, accumulate=False):
    dy = dpooled.repeat_interleave(L, 0) * row_weight[:, None]
    return layernorm_bwd(dy, x, gamma, None, mean, rstd, want_f32=want_f32, want_bf16=want_bf16, dgamma=dgamma,
                         dbeta=dbeta, accumulate=accumulate)


def l2norm_fwd(x, eps=1e-12):
    n = x.norm(dim=-1)
    return x / n.clamp_min(eps)[:, None], n


def l2norm_bwd(dy, y, n, eps=1e-12):
    dot = (dy * y).sum(-1, keepdim=True)
    return (dy - y * dot) / n.clamp_min(eps)[:, None]


def to_bf16(x):
    return x.to(BF)


def to_f32(x):
    return x.float()


def il_source_rows(n_rows, hd, il_rows):
    r = torch.arange(n_rows, dtype=torch.int64)
    d = r % hd
    return torch.where(r < il_rows, (r - d) + (d // 2) + (d % 2) * (hd // 2), r)


def cast_transpose(w, want_w=True, want_wt=True, w_out=None, wt_out=None, il=(0, 0)):
    if il[1]:
        w = w[il_source_rows(w.shape[0], il[0], il[1])]
    wb = w.to(BF)
    return wb, wb.t().contiguous()


def act_fwd(x, act):
    return _act(x, act)


def act_bwd(dy, x, act):
    return dy * _act_grad(x, act)


def dact(dy, aux_bf16, act):
    return (dy.float() * _act_grad(aux_bf16.float(), act)).to(BF)


def axpby_dev(a, b, s):
    return s * b if a is None else a + s * b


def _attn_math(qkv, B, L, H, D, key_mask, rope, q_scale, dropout=None, row0=0, Lstride=None):
    x = qkv.view(B, L, 3, H, D).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    if rope is not None:
        cos, sin = rope
        cosf, sinf = torch.cat([cos, cos], -1)[None, None], torch.cat([sin, sin], -1)[None, None]

        def rot(t):
            return torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)
        q = (q * cosf + rot(q) * sinf).to(BF).float()        # kernels keep rotated q/k in bf16
        k = (k * cosf + rot(k) * sinf).to(BF).float()
    s = (q @ k.transpose(-1, -2)) * q_scale
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    if dropout:
        # index = ((token row of the query) * H + h) * Lstride + key, as the kernels (attention.hip attn_drop)
        Ls = L if Lstride is None else Lstride
        qrow = row0 + (torch.arange(B)[:, None] * L + torch.arange(L)[None, :])                 # [B, L]
        idx = ((qrow[:, None, :, None] * H + torch.arange(H)[None, :, None, None]) * Ls
               + torch.arange(L)[None, None, None, :]).to(torch.int64)
        p = p * drop_mult(dropout[0], dropout[1], idx)
    o = p @ v
    return o.permute(0, 2, 1, 3).reshape(B * L, H * D), torch.logsumexp(s, -1)


def attn_fwd(qkv, B, L, H, D, key_mask=None, rope=None, q_scale=1.0, dropout=None, row0=0, Lstride=None):
    o, lse = _attn_math(qkv.float(), B, L, H, D, key_mask, rope, q_scale, dropout, row0, Lstride)
    return o.to(BF), lse


def _rope_qk_il(x, B, L, H, D, rope):
    """as _rope_qk for heads in pair-interleaved column order: columns 2 j, 2 j + 1 are a pair with angle j"""
    cos, sin = rope
    v = x.view(B, L, 3, H, D // 2, 2)
    c, s_ = cos[None, :, None, None, :], sin[None, :, None, None, :]
    qk = v[:, :, :2]
    y = torch.stack([qk[..., 0] * c - qk[..., 1] * s_, qk[..., 1] * c + qk[..., 0] * s_], -1)
    return torch.cat([y, v[:, :, 2:]], 2).reshape(B * L, 3 * H * D)


def _rope_qk(x, B, L, H, D, rope):
    """rotate the q and k sections of x [B*L, 3*H*D] (f32 math)"""
    cos, sin = rope
    cosf, sinf = torch.cat([cos, cos], -1)[None, :, None], torch.cat([sin, sin], -1)[None, :, None]
    v = x.view(B, L, 3, H, D)

    def rot(t):
        return torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)
    qk = v[:, :, :2]
    qk = qk * cosf[:, :, None] + rot(qk) * sinf[:, :, None]
    return torch.cat([qk, v[:, :, 2:]], 2).reshape(B * L, 3 * H * D)


def rope_qk_(qkv, B, L, H, D, rope):
    qkv.copy_(_rope_qk(qkv.float(), B, L, H, D, rope).to(BF))
    return qkv


def attn_fwd_rot_(qkv, B, L, H, D, rope, key_mask=None, q_scale=1.0):
    rope_qk_(qkv, B, L, H, D, rope)
    return attn_fwd(qkv, B, L, H, D, key_mask=key_mask, rope=None, q_scale=q_scale)


def attn_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=None, rope=None, q_scale=1.0, prerotated=False, dropout=None,
             row0=0, Lstride=None):
    with torch.enable_grad():
        q = qkv.float().detach().requires_grad_(True)
        if prerotated and rope is not None:
            # q, k are rotated already: gradient w.r.t. the rotated rows, then through RoPE^T (= vjp of the rotation)
            o, _ = _attn_math(q, B, L, H, D, key_mask, None, q_scale)
            g_rot, = torch.autograd.grad(o, q, dout.float())
            z = torch.zeros_like(q).requires_grad_(True)
            rot = _rope_qk_il if int(prerotated) == 2 else _rope_qk       # 2: pair-interleaved head order (clipk.h)
            g, = torch.autograd.grad(rot(z, B, L, H, D, rope), z, g_rot)
        else:
            o, _ = _attn_math(q, B, L, H, D, key_mask, rope, q_scale, dropout, row0, Lstride)
            g, = torch.autograd.grad(o, q, dout.float())
    return g.to(BF)


def attn_f32_fwd(qkv, B, L, H, D, key_mask=None, q_scale=1.0, dropout=None):
    o, lse = _attn_math(qkv, B, L, H, D, key_mask, None, q_scale, dropout)
    return torch.nan_to_num(o, nan=0.0), lse               # kernels: a fully masked row gives zeros


def attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=None, q_scale=1.0, dropout=None):
    with torch.enable_grad():
        q = qkv.detach().requires_grad_(True)
        o, _ = _attn_math(q, B, L, H, D, key_mask, None, q_scale, dropout)
        g, = torch.autograd.grad(o, q, dout)
    return g


def dropout_f32(x, dropout, addend=None):
    y = x * drop_mult(dropout[0], dropout[1], torch.arange(x.numel(), dtype=torch.int64).view(x.shape))
    return y if addend is None else y + addend


def _segments(cu):
    cu = [int(v) for v in cu.tolist()]
    return [(cu[i], cu[i + 1]) for i in range(len(cu) - 1)]


def attn_varlen_fwd(qkv, cu_seqlens, max_len, H, D, rope=None, q_scale=1.0, dropout=None):
    outs, lses = [], []
    for a, b in _segments(cu_seqlens):
        o, l = attn_fwd(qkv[a:b].contiguous(), 1, b - a, H, D, rope=None if rope is None else (rope[0][: b - a], rope[1][: b - a]),
                        q_scale=q_scale, dropout=dropout, row0=a, Lstride=max_len)
        outs.append(o)
        lses.append(l.reshape(H, b - a))
    return torch.cat(outs, 0), torch.cat(lses, 1)


def varlen_whole_head_applies(max_len, D):
    return D in (16, 24, 32) and 128 < int(max_len) <= 256


def attn_varlen_fwd_rot_(qkv, cu_seqlens, max_len, H, D, rope, q_scale=1.0):
    for a, b in _segments(cu_seqlens):                     # rotate every sequence's q / k in place, positions from 0
        qkv[a:b] = _rope_qk(qkv[a:b].float(), 1, b - a, H, D, (rope[0][: b - a], rope[1][: b - a])).to(BF)
    return attn_varlen_fwd(qkv, cu_seqlens, max_len, H, D, rope=None, q_scale=q_scale)


def attn_varlen_bwd(qkv, out, dout, lse, cu_seqlens, max_len, H, D, rope=None, q_scale=1.0, dropout=None,
                    prerotated=False):
    gs = []
    for a, b in _segments(cu_seqlens):
        r = None if rope is None else (rope[0][: b - a], rope[1][: b - a])
        gs.append(attn_bwd(qkv[a:b].contiguous(), out[a:b].contiguous(), dout[a:b].contiguous(),
                           lse[:, a:b].reshape(1, H, b - a).contiguous(), 1, b - a, H, D, rope=r, q_scale=q_scale,
                           prerotated=prerotated, dropout=dropout, row0=a, Lstride=max_len))
    return torch.cat(gs, 0)


def pool_varlen_fwd(x, cu_seqlens, mode=1):
    return torch.stack([x[a] if mode == 0 else x[a:b].mean(0) for a, b in _segments(cu_seqlens)])


def pool_varlen_bwd(dy, cu_seqlens, T, mode=1):
    dx = torch.zeros(T, dy.shape[1], dtype=dy.dtype)
    for i, (a, b) in enumerate(_segments(cu_seqlens)):
        if mode == 0:
            dx[a] = dy[i]
        else:
            dx[a:b] = dy[i] / (b - a)
    return dx


def embed_fwd(ids, table, row_scale=None, mask=None, mask_token_id=-1):
    B, L = ids.shape
    x = table[ids]
    sc = torch.ones(B, L)
    if row_scale is not None:
        sc = sc * row_scale[:, None]
    if mask is not None:
        sc = sc * mask.view(B, L).float()
    sc = sc * (ids != mask_token_id).float()
    return (x * sc[..., None]).reshape(B * L, -1)


def embed_bwd(ids, dx, dtable, row_scale=None, mask=None, mask_token_id=-1):
    B, L = ids.shape
    sc = torch.ones(B, L)
    if row_scale is not None:
        sc = sc * row_scale[:, None]
    if mask is not None:
        sc = sc * mask.view(B, L).float()
    sc = sc * (ids != mask_token_id).float()
    dtable.index_add_(0, ids.view(-1), dx * sc.view(-1, 1))
    return dtable


def pool_fwd(x, B, L, mask=None, mode=1):
    x3 = x.view(B, L, -1)
    if mode == 0:
        return x3[:, 0].clone()
    if mask is None:
        return x3.mean(1)
    m = mask.view(B, L).float()
    return (x3 * m[..., None]).sum(1) / m.sum(1, keepdim=True)


def pool_bwd(dy, B, L, mask=None, mode=1):
    d = dy.shape[-1]
    if mode == 0:
        dx = torch.zeros(B, L, d)
        dx[:, 0] = dy
        return dx.view(B * L, d)
    m = mask.view(B, L).float() if mask is not None else torch.ones(B, L)
    return (dy[:, None] * (m / m.sum(1, keepdim=True))[..., None]).reshape(B * L, d)


def sumsq(g, out=None):
    v = (g.double() ** 2).sum().float().reshape(1)
    if out is not None:
        out.copy_(v)
        return out
    return v


def adamw_step(w, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_norm=0.0, grad_scale=1.0,
               w_bf16=None, hyper=None):
    clip = grad_scale
    if grad_norm_sq is not None:
        norm = grad_norm_sq.sqrt().item() * grad_scale
        clip *= min(1.0, max_norm / (norm + 1e-6))
    gi = g * clip
    w.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(gi, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    if hyper is not None:
        lr, bc1, bc2 = float(hyper[0]), float(hyper[1]), float(hyper[2]) ** 2
    w.addcdiv_(m, v.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)


_NAMES = [n for n, f in list(globals().items()) if callable(f) and not n.startswith("_") and n not in ("install",)]


def install(monkeypatch):
    """Route clip_dplm_amd.ops.* to this restatement for the duration of one test."""
    from clip_dplm_amd import ops
    for n in _NAMES:
        if hasattr(ops, n) and n not in ("KernelTimer",):
            monkeypatch.setattr(ops, n, globals()[n])
