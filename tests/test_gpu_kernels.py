"""GPU parity tests of every libclipk kernel, called through the C ABI (ctypes).

Each check compares the HIP kernel with a plain PyTorch f32 computation of the same op on the same
(bf16-rounded where the kernel takes bf16) inputs.  Tolerances are stated per test.
"""
import math
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from clip_dplm_amd import ops
    return ops


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def test_probe_layouts():
    """Hardware lane maps the kernels rely on (MFMA A/B/C, ds_read_b64_tr_b16)."""
    exe = os.path.join(ROOT, "tools", "probes", "probe_layouts")
    if not os.path.exists(exe):
        pytest.skip("probe binary not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (512, 1024, 480), (300, 360, 120), (1000, 1440, 480),
                                   (128, 480, 1920), (77, 8, 8), (4096, 2304, 768)])
def test_gemm_nt_plain(dev, M, N, K):
    ops = _ops()
    a = _rand((M, K), dev, 1, dtype=torch.bfloat16)
    b = _rand((N, K), dev, 2, 0.05, dtype=torch.bfloat16)
    ref = a.float() @ b.float().t()
    c32 = ops.gemm_nt(a, b, out_dtype=torch.float32)
    # f32 accumulate of exact bf16 products: only summation order differs
    assert torch.allclose(c32, ref, rtol=1e-4, atol=1e-4 * math.sqrt(K)), (c32 - ref).abs().max()
    c16 = ops.gemm_nt(a, b)
    assert torch.allclose(c16.float(), ref, rtol=1e-2, atol=1e-2)


_GEMM_KERNEL = {"v2": 2, "v3": 3, "v4": 4}      # option gemm_kernel: 128x128 | persistent 256x256 | persistent 128x256 x 2 per CU


@pytest.mark.parametrize("kernel,M", [("v2", 384), ("v3", 2304), ("v4", 2304)])
@pytest.mark.parametrize("act", [None, "relu", "gelu"])
def test_gemm_nt_epilogue(dev, kopt, act, kernel, M):
    """run-time (generic) epilogue combinations — f32 out + residual + pre-activation, bf16 residual, act' — through
    all three tile structures"""
    kopt("gemm_kernel", _GEMM_KERNEL[kernel])
    ops = _ops()
    N, K = 256, 192
    a = _rand((M, K), dev, 3, dtype=torch.bfloat16)
    b = _rand((N, K), dev, 4, 0.1, dtype=torch.bfloat16)
    bias = _rand((N,), dev, 5)
    res = _rand((M, N), dev, 6)
    pre_ref = a.float() @ b.float().t() + bias
    if act == "relu":
        act_ref = F.relu(pre_ref)
    elif act == "gelu":
        act_ref = F.gelu(pre_ref)
    else:
        act_ref = pre_ref
    out, pre = ops.gemm_nt(a, b, bias=bias, act=act, out_dtype=torch.float32, residual=res, out_preact=True)
    assert torch.allclose(out, act_ref + res, rtol=1e-4, atol=2e-4)
    assert torch.allclose(pre.float(), pre_ref, rtol=1e-2, atol=1e-2)
    # bf16 residual + bf16 out
    res16 = res.to(torch.bfloat16)
    out16 = ops.gemm_nt(a, b, bias=bias, act=act, residual=res16)
    assert torch.allclose(out16.float(), act_ref + res16.float(), rtol=2e-2, atol=2e-2)
    # activation-derivative epilogue (dgrad of Linear -> act): v * act'(aux)
    aux = _rand((M, N), dev, 7, dtype=torch.bfloat16)
    auxf = aux.float().requires_grad_(True)
    y = {"relu": F.relu, "gelu": F.gelu, None: (lambda t: t)}[act](auxf)
    gref, = torch.autograd.grad(y, auxf, torch.ones_like(y))
    outd = ops.gemm_nt(a, b, out_dtype=torch.float32, dact_aux=aux, dact=act)
    assert torch.allclose(outd, (a.float() @ b.float().t()) * gref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("kernel", ["v2", "v3", "v4", "v2generic", "v3generic", "v4generic"])
@pytest.mark.parametrize("M,N,K", [(2048, 256, 128), (2500, 360, 160), (4096, 1440, 480), (2304, 480, 1920),
                                   (3000, 776, 192), (66000, 520, 480), (40000, 1000, 224)])
def test_gemm_nt_specialised_epilogues(dev, kopt, kernel, M, N, K):
    """The four compile-time epilogue modes (gemm_epilogue.h) of the 128x128 kernel and of the persistent 256x256
    phase-interleaved kernel (option gemm_kernel = 3), ragged M / N edges and the K % 64 == 32 tail, against torch f32.  The
    two largest shapes have 774 / 628 output tiles, i.e. every persistent workgroup walks 2-4 tiles: that covers the
    next-tile prefetch under the epilogue and the store-tolerant vmcnt bookkeeping."""
    ops = _ops()
    kopt("gemm_kernel", _GEMM_KERNEL[kernel[:2]])
    if kernel.endswith("generic"):
        kopt("gemm_epi_generic", 1)
    a = _rand((M, K), dev, 11, dtype=torch.bfloat16)
    b = _rand((N, K), dev, 12, 0.05, dtype=torch.bfloat16)
    bias = _rand((N,), dev, 13)
    res = _rand((M, N), dev, 14)
    aux = _rand((M, N), dev, 15, dtype=torch.bfloat16)
    ref = a.float() @ b.float().t()
    tol = dict(rtol=1e-2, atol=1e-2)
    # PLAIN (with and without bias)
    assert torch.allclose(ops.gemm_nt(a, b).float(), ref, **tol)
    assert torch.allclose(ops.gemm_nt(a, b, bias=bias).float(), ref + bias, **tol)
    # RES32
    out = ops.gemm_nt(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    assert torch.allclose(out, ref + bias + res, rtol=1e-4, atol=1e-4 * math.sqrt(K)), (out - ref - bias - res).abs().max()
    # GELU_PRE
    g, u = ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True)
    assert torch.allclose(u.float(), ref + bias, **tol)
    assert torch.allclose(g.float(), F.gelu(ref + bias), **tol)
    # RES16 / PRES16: bf16 residual, f32 or bf16 out
    res16 = res.to(torch.bfloat16)
    out = ops.gemm_nt(a, b, bias=bias, residual=res16, out_dtype=torch.float32)
    assert torch.allclose(out, ref + bias + res16.float(), rtol=1e-4, atol=1e-4 * math.sqrt(K))
    out = ops.gemm_nt(a, b, residual=res16)
    assert torch.allclose(out.float(), ref + res16.float(), **tol)
    # DGELU
    auxf = aux.float().requires_grad_(True)
    gref, = torch.autograd.grad(F.gelu(auxf), auxf, torch.ones_like(auxf))
    d = ops.gemm_nt(a, b, dact_aux=aux, dact="gelu")
    assert torch.allclose(d.float(), ref * gref, **tol)
    # GELU_D8 / DGELU8: the FFN pair's auxiliary as 8-bit codes of GELU'(pre-activation) (clipk.h aux_dtype)
    g8, q = ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True, aux_u8=True)
    assert q.dtype == torch.uint8 and q.shape == (M, N)
    assert torch.equal(g8, g)                                   # the forward value does not depend on the aux format
    pre = (ref + bias).requires_grad_(True)
    dref, = torch.autograd.grad(F.gelu(pre), pre, torch.ones_like(pre))
    step, lo = 0.005, -0.13                                     # code 26 <-> 0, code 226 <-> 1 (csrc/common.h)
    dec = q.float() * step + lo
    # the kernel's pre-activation differs from torch's f32 product by summation order only: codes within one level
    assert (dec - dref).abs().max().item() <= 1.5 * step, (dec - dref).abs().max().item()
    assert (dec - dref).abs().mean().item() <= 0.3 * step
    d8 = ops.gemm_nt(a, b, dact_aux=q, dact="gelu")
    assert torch.allclose(d8.float(), ref * dec, **tol)
    # nothing may be written outside [M, N]: run into a padded buffer and check the guard band
    big = torch.full((M + 8, N + 8), 7.0, dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(a, b, bias=bias, out=big[:M, :N])
    assert (big[M:] == 7.0).all() and (big[:, N:] == 7.0).all()
    assert torch.allclose(big[:M, :N].float(), ref + bias, **tol)


@pytest.mark.parametrize("kernel", ["v2", "v3", "v3p"])
@pytest.mark.parametrize("M,N,K", [(512, 128, 128), (4096, 1440, 480), (1000, 360, 120), (8192, 480, 1920),
                                   (300, 8, 16), (16384, 768, 768), (20000, 1440, 480), (1100, 264, 520),
                                   (1024, 2048, 768), (66000, 480, 480)])
def test_gemm_wgrad(dev, kopt, kernel, M, N, K):
    """128x128 kernel and the 256x256 kernel in both of its schedules (option wgrad_kernel = 3: 8-phase, 4: software
    pipelined; both take every M >= 1024): ragged N / K edges, a ragged last 64-row step (M = 20000, 1100, 66000), one-,
    two- and three-step splits, bias turns over 1 / 2 / 3 k-tile workgroups."""
    kopt("wgrad_kernel", {"v2": 2, "v3": 3, "v3p": 4}[kernel])
    ops = _ops()
    dy = _rand((M, N), dev, 8, 0.1, dtype=torch.bfloat16)
    x = _rand((M, K), dev, 9, dtype=torch.bfloat16)
    ref = dy.float().t() @ x.float()
    bref = dy.float().sum(0)
    dw, db = ops.gemm_wgrad(dy, x, want_bias=True)
    tol = 2e-4 * math.sqrt(M)
    assert torch.allclose(dw, ref, rtol=1e-4, atol=tol), (dw - ref).abs().max()
    assert torch.allclose(db, bref, rtol=1e-4, atol=tol), (db - bref).abs().max()
    # accumulate into existing buffers; second call must be bitwise reproducible
    dw2, db2 = ops.gemm_wgrad(dy, x, dw=dw.clone(), dbias=db.clone(), accumulate=True)
    assert torch.allclose(dw2, 2 * ref, rtol=1e-4, atol=2 * tol)
    assert torch.allclose(db2, 2 * bref, rtol=1e-4, atol=2 * tol)
    dw3, _ = ops.gemm_wgrad(dy, x, want_bias=True)
    assert torch.equal(dw3, dw)


def test_cast_transpose_batched(dev):
    """one launch over several weights == the per-weight kernel (ragged 64x64 tile edges included)"""
    ops = _ops()
    shapes = [(480, 480), (1440, 480), (100, 72), (8, 1920), (513, 65)]
    ws = [_rand(sh, dev, 40 + i) for i, sh in enumerate(shapes)]
    ils = [(0, 0), (24, 960), (0, 0), (0, 0), (0, 0)]             # the fused qkv weight: q and k sections pair-interleaved
    ref = [ops.cast_transpose(w, il=il) for w, il in zip(ws, ils)]
    outs = [(torch.zeros_like(a), torch.zeros_like(b)) for a, b in ref]
    desc = torch.tensor([(w.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), w.shape[0], w.shape[1], il[0] or 2, il[1])
                         for w, o, il in zip(ws, outs, ils)], dtype=torch.int64).to(dev)
    ops.cast_transpose_batched(desc)
    for (a, b), (oa, ob) in zip(ref, outs):
        assert torch.equal(a, oa) and torch.equal(b, ob)
    # pair-interleaved head order: copy row 2 j / 2 j + 1 of a head = original row j / j + 12
    src = ops.il_source_rows(1440, 24, 960).to(dev)
    assert src[:6].tolist() == [0, 12, 1, 13, 2, 14] and src[24:28].tolist() == [24, 36, 25, 37] and src[960] == 960
    assert torch.equal(ref[1][0], ws[1][src].to(torch.bfloat16))
    assert torch.equal(ref[1][1], ws[1][src].t().contiguous().to(torch.bfloat16))


def test_weight_cache_batched_refresh(dev):
    """FusedAdamW.step() refreshes all bf16 weight copies in one launch; the next forward must see the new weights"""
    import clip_dplm_amd as K
    from clip_dplm_amd import functional as KF
    from types import SimpleNamespace as NS
    torch.manual_seed(0)
    sub = lambda h: NS(hidden_size=h, num_hidden_layers=2, layer_norm_eps=1e-12)
    cfg = NS(rna_config=sub(64), protein_config=sub(64), diffmap_config=sub(64), projection_dim=32,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg).to(dev).eval()
    opt = K.FusedAdamW(m, lr=1e-2)
    a, b = _rand((64, 64), dev, 50), _rand((64, 64), dev, 51)
    opt.zero_grad(); m.loss(a, b, symmetric=True).backward(); opt.step()
    stale = [c for c in KF._CACHES if c.src is not None and c.key != (c.src.data_ptr(), c.src._version, KF._WEIGHT_EPOCH,
                                                                    tuple(c.src.shape))]
    assert not stale                                     # everything was rebuilt by the step itself
    for c in KF._CACHES:
        if c.src is not None and c.wb is not None:
            assert torch.equal(c.wb, c.src.detach().to(torch.bfloat16))
            assert torch.equal(c.wtb, c.src.detach().t().contiguous().to(torch.bfloat16))


@pytest.mark.parametrize("B,L,H,D", [(3, 100, 4, 24), (2, 256, 20, 24), (2, 130, 2, 64)])
def test_rope_prerotation_equals_rotation_at_staging(dev, B, L, H, D, kopt):
    """clipk_rope_qk (in place, once) + attention without forward rotation == attention that rotates q / k while
    staging them: identical bf16 values reach the MFMAs, so outputs, LSE and dqkv (same kernels) are bit-identical."""
    ops = _ops()
    kopt("attn_fused_bwd", 0)
    qkv = _rand((B * L, 3 * H * D), dev, 60, dtype=torch.bfloat16)
    dout = _rand((B * L, H * D), dev, 61, dtype=torch.bfloat16)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    rope = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
    lens = torch.tensor([L] + [max(1, L - 17 * (i + 1)) for i in range(B - 1)])
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).contiguous().to(dev)
    o1, lse1 = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5)
    g1 = ops.attn_bwd(qkv, o1, dout, lse1, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5)
    rot = ops.rope_qk_(qkv.clone(), B, L, H, D, rope)
    assert torch.equal(rot.view(B * L, 3, H * D)[:, 2], qkv.view(B * L, 3, H * D)[:, 2])        # v untouched
    o2, lse2 = ops.attn_fwd(rot, B, L, H, D, key_mask=mask, rope=None, q_scale=D ** -0.5)
    g2 = ops.attn_bwd(rot, o2, dout, lse2, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5, prerotated=True)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    assert torch.equal(g1, g2)
    # default dispatch: short heads at 128 < L <= 256 take the whole-head backward when q / k arrive rotated; it sums
    # in another order, so agreement there is to bf16 rounding
    kopt("attn_fused_bwd", -1)
    g3 = ops.attn_bwd(rot, o2, dout, lse2, B, L, H, D, key_mask=mask, rope=rope, q_scale=D ** -0.5, prerotated=True)
    m3 = mask.view(B * L, 1).float()
    assert ((g3.float() - g1.float()) * m3).abs().max().item() < 8e-3 * (g1.float() * m3).abs().max().item()


# ------------------------------------------------------------------------------------------------ simce
def _unit(shape, dev, seed):
    return F.normalize(_rand(shape, dev, seed), dim=-1).contiguous()


@pytest.mark.parametrize("B,P,scale", [(256, 128, 14.2849), (512, 512, 14.2849), (100, 64, 100.0), (1024, 512, 30.0),
                                       (40, 768, 5.0)])
def test_simce_symmetric(dev, B, P, scale):
    ops = _ops()
    a, b = _unit((B, P), dev, 10), _unit((B, P), dev, 11)
    sc = torch.tensor([scale], device=dev)
    S = (a.double() @ b.double().t()) * scale
    lab = torch.arange(B, device=dev)
    lse_r, pos_r = ops.simce_lse(a, b, sc)
    lse_c, pos_c = ops.simce_lse(b, a, sc)
    assert torch.allclose(lse_r.double(), torch.logsumexp(S, 1), rtol=0, atol=2e-5)
    assert torch.allclose(lse_c.double(), torch.logsumexp(S, 0), rtol=0, atol=2e-5)
    assert torch.allclose(pos_r.double(), S.diag(), rtol=0, atol=2e-5)
    assert torch.allclose(pos_c.double(), S.diag(), rtol=0, atol=2e-5)
    loss = 0.5 * ((lse_r - pos_r).mean() + (lse_c - pos_c).mean())
    ref = 0.5 * (F.cross_entropy(S, lab) + F.cross_entropy(S.t(), lab))
    assert abs(loss.item() - ref.item()) < 1e-5          # loss parity bar is 1e-3; f32 MFMA gives ~1e-6
    # logits for the drop-in API
    Sk = ops.sim_logits(a, b, sc)
    assert torch.allclose(Sk.double(), S, rtol=0, atol=2e-5)
    # gradients of the symmetric loss w.r.t. a, b and the scale
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    sd = torch.tensor(scale, dtype=torch.float64, device=dev, requires_grad=True)
    Sd = (ad @ bd.t()) * sd
    Lr = 0.5 * (F.cross_entropy(Sd, lab) + F.cross_entropy(Sd.t(), lab))
    ga, gb, gs = torch.autograd.grad(Lr, (ad, bd, sd))
    da, dsa = ops.simce_grad(a, b, sc, lse_r, lse_c, 0.5, 0.5, 1.0 / B)
    db, dsb = ops.simce_grad(b, a, sc, lse_c, lse_r, 0.5, 0.5, 1.0 / B)
    assert torch.allclose(da.double(), ga, rtol=1e-4, atol=1e-6), (da.double() - ga).abs().max()
    assert torch.allclose(db.double(), gb, rtol=1e-4, atol=1e-6)
    assert abs(dsa.sum().item() - gs.item()) < 1e-5 * max(1.0, abs(gs.item()))
    assert abs(dsb.sum().item() - gs.item()) < 1e-5 * max(1.0, abs(gs.item()))


def test_simce_one_sided_and_cache(dev):
    """One-sided CE (old/ablation.py:16) and the cache-column variant of old/clip_opt.py:130-151."""
    ops = _ops()
    B, P, Nc, scale = 128, 128, 200, 14.2849
    a, b, cache = _unit((B, P), dev, 12), _unit((B, P), dev, 13), _unit((Nc, P), dev, 14)
    sc = torch.tensor([scale], device=dev)
    lab = torch.arange(B, device=dev)
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    S = (ad @ bd.t()) * scale
    Sc = (ad @ cache.double().t()) * scale
    ref = 0.5 * (F.cross_entropy(torch.cat([S, Sc], 1), lab) + F.cross_entropy(S.t(), lab))
    ga, gb = torch.autograd.grad(ref, (ad, bd))
    lse_r, pos_r = ops.simce_lse(a, b, sc, cache=cache)
    lse_c, pos_c = ops.simce_lse(b, a, sc)
    loss = 0.5 * ((lse_r - pos_r).mean() + (lse_c - pos_c).mean())
    assert abs(loss.item() - ref.item()) < 1e-5
    # d/da: rows of a see keys b (+cache) in the row direction and b in the column direction
    da, _ = ops.simce_grad(a, b, sc, lse_r, lse_c, 0.5, 0.5, 1.0 / B, cache=cache)
    # d/db: rows of b are "queries" of the column direction (w_row=0.5 -> lse_c) and keys of the row direction
    db, _ = ops.simce_grad(b, a, sc, lse_c, lse_r, 0.5, 0.5, 1.0 / B)
    assert torch.allclose(da.double(), ga, rtol=1e-4, atol=1e-6)
    assert torch.allclose(db.double(), gb, rtol=1e-4, atol=1e-6)
    # one-sided
    ref1 = F.cross_entropy((a.double() @ b.double().t()) * scale, lab)
    lse1, pos1 = ops.simce_lse(a, b, sc)
    assert abs((lse1 - pos1).mean().item() - ref1.item()) < 1e-5


def test_simce_sharded_equals_global(dev):
    """Virtual ranks: W shards of the batch with label offsets reproduce the unsharded loss/gradients."""
    ops = _ops()
    W, Bl, P, scale = 4, 96, 256, 20.0
    Bg = W * Bl
    a, b = _unit((Bg, P), dev, 15), _unit((Bg, P), dev, 16)
    sc = torch.tensor([scale], device=dev)
    lse_r_g, pos_g = ops.simce_lse(a, b, sc)
    lse_c_g, _ = ops.simce_lse(b, a, sc)
    da_g, _ = ops.simce_grad(a, b, sc, lse_r_g, lse_c_g, 0.5, 0.5, 1.0 / Bg)
    lse_r_parts, da_parts = [], []
    for r in range(W):
        sl = slice(r * Bl, (r + 1) * Bl)
        lr, pr = ops.simce_lse(a[sl].contiguous(), b, sc, label_offset=r * Bl)
        lse_r_parts.append(lr)
        assert torch.allclose(pr, pos_g[sl], atol=1e-6)
    lse_r_cat = torch.cat(lse_r_parts)
    assert torch.allclose(lse_r_cat, lse_r_g, atol=1e-5)
    for r in range(W):
        sl = slice(r * Bl, (r + 1) * Bl)
        d, _ = ops.simce_grad(a[sl].contiguous(), b, sc, lse_r_cat[sl].contiguous(), lse_c_g, 0.5, 0.5, 1.0 / Bg,
                              label_offset=r * Bl)
        da_parts.append(d)
    assert torch.allclose(torch.cat(da_parts), da_g, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ row-wise
@pytest.mark.parametrize("rows,cols", [(256, 128), (1000, 480), (64, 1024), (33, 2560), (512, 768), (16, 120),
                                       (2500, 2560), (7, 5120), (9, 4100)])
@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_layernorm(dev, rows, cols, xdt):
    """clipk_layernorm_fwd / _bwd against torch autograd; cols > 2048 take the one-workgroup-per-row kernels (forward both
    input types, backward f32), with more rows than workgroups and a ragged width among the cases."""
    ops = _ops()
    x = _rand((rows, cols), dev, 20, 2.0).to(xdt)
    g, b = _rand((cols,), dev, 21) * 0.2 + 1.0, _rand((cols,), dev, 22) * 0.1
    for act, eps in ((None, 1e-12), ("gelu", 1e-5)):
        xf = x.float().requires_grad_(True)
        gg, bb = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = F.layer_norm(xf, (cols,), gg, bb, eps)
        if act == "gelu":
            y = F.gelu(y)
        y32, y16, mean, rstd = ops.layernorm_fwd(x, g, b, eps, act=act, want_f32=True, want_bf16=True)
        assert torch.allclose(y32, y.detach(), rtol=1e-5, atol=2e-5), (y32 - y).abs().max()
        assert torch.allclose(y16.float(), y.detach(), rtol=1e-2, atol=1e-2)
        dy = _rand((rows, cols), dev, 23)
        add = _rand((rows, cols), dev, 24)
        gx, ggm, gbt = torch.autograd.grad(y, (xf, gg, bb), dy)
        dx32, dx16, dgam, dbet = ops.layernorm_bwd(dy, x, g, b, mean, rstd, act=act, dx_add=add, want_f32=True,
                                                   want_bf16=True)
        assert torch.allclose(dx32, gx + add, rtol=1e-4, atol=5e-5), (dx32 - gx - add).abs().max()
        assert torch.allclose(dgam, ggm, rtol=1e-4, atol=1e-4 * math.sqrt(rows))
        assert torch.allclose(dbet, gbt, rtol=1e-4, atol=1e-4 * math.sqrt(rows))
        assert torch.allclose(dx16.float(), gx + add, rtol=2e-2, atol=2e-2)


def test_layernorm_bwd_deferred_parameter_gradients_equal_the_immediate_ones(dev):
    """clipk_layernorm_bwd with dgamma = dbeta = NULL leaves its partial rows in the caller's workspace; clipk_colreduce_batched
    reduces several LayerNorms' rows in one launch (`ops.colreduce_entries`): same dgamma / dbeta bits as the immediate form
    (same summation order), accumulate semantics, widths 120 .. 2560 (narrow and one-workgroup-per-row kernels) in one batch."""
    ops = _ops()
    entries, want = [], []
    for i, (rows, cols) in enumerate([(32, 1280), (32, 120), (256, 128), (32, 2560), (1000, 480)]):
        x, dy = _rand((rows, cols), dev, 130 + i, 1.5), _rand((rows, cols), dev, 140 + i)
        g, b = _rand((cols,), dev, 150 + i) * 0.2 + 1.0, _rand((cols,), dev, 160 + i) * 0.1
        _, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
        dg0, db0 = _rand((cols,), dev, 170 + i), _rand((cols,), dev, 180 + i)
        dg_ref, db_ref = dg0.clone(), db0.clone()
        dx_ref = ops.layernorm_bwd(dy, x, g, b, mean, rstd, dgamma=dg_ref, dbeta=db_ref, accumulate=True)[0]
        blocks, nfloat = ops.layernorm_bwd_partial_shape(rows, cols)
        part = torch.empty(nfloat, device=dev)
        dx = ops.layernorm_bwd(dy, x, g, b, mean, rstd, part_out=part)[0]
        assert torch.equal(dx, dx_ref)
        entries.append((part, blocks, cols, dg0, db0))
        want.append((dg_ref, db_ref))
    ops.colreduce_entries(entries)
    for (_, _, _, dg, db), (rg, rb) in zip(entries, want):
        assert torch.equal(dg, rg) and torch.equal(db, rb)


@pytest.mark.parametrize("rows,cols", [(256, 128), (512, 512), (10, 32), (64, 768)])
def test_l2norm(dev, rows, cols):
    ops = _ops()
    x = _rand((rows, cols), dev, 25)
    xr = x.clone().requires_grad_(True)
    y = F.normalize(xr, dim=-1)
    yk, n = ops.l2norm_fwd(x)
    assert torch.allclose(yk, y.detach(), rtol=1e-6, atol=1e-7)
    dy = _rand((rows, cols), dev, 26)
    gx, = torch.autograd.grad(y, xr, dy)
    dx = ops.l2norm_bwd(dy, yk, n)
    assert torch.allclose(dx, gx, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, L, H, D, mask, rope, scale):
    """f32 reference on the same bf16-rounded inputs; returns out [B*L, H*D] and lets autograd do bwd."""
    x = qkv.view(B, L, 3, H, D).permute(2, 0, 3, 1, 4)       # 3,B,H,L,D
    q, k, v = x[0], x[1], x[2]
    if rope is not None:
        cos, sin = rope                                       # [L, D/2]
        cosf = torch.cat([cos, cos], -1)[None, None]
        sinf = torch.cat([sin, sin], -1)[None, None]

        def rot(t):
            t1, t2 = t[..., : D // 2], t[..., D // 2:]
            return torch.cat([-t2, t1], -1)
        q = q * cosf + rot(q) * sinf
        k = k * cosf + rot(k) * sinf
    s = (q @ k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    o = p @ v                                                 # B,H,L,D
    return o.permute(0, 2, 1, 3).reshape(B * L, H * D), torch.logsumexp(s, -1)


def _rope_tables(L, D, dev):
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    return fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)


@pytest.mark.parametrize("B,L,H,D,use_rope,use_mask", [
    (2, 256, 20, 24, True, False),     # ESM-2-35M head shape
    (2, 256, 8, 96, False, False),     # RNA encoder head shape
    (3, 200, 4, 64, True, True),       # ragged length + key padding
    (2, 64, 8, 16, False, True),
    (1, 300, 2, 160, False, True),     # notebook RBP head dim (1280/8)
    (2, 1024, 2, 64, True, False),     # ESM-2-650M head shape, L = 1024
    (1, 130, 3, 128, True, True),
    (3, 200, 4, 24, True, True),       # whole-head backward kernel (hd <= 32, 128 < L <= 256): ragged + padding
    (2, 256, 4, 32, True, True),
    (2, 130, 3, 16, False, True),
    (9, 256, 5, 24, False, False),     # batch not a multiple of 8 (plain work-item order)
    (1, 2542, 8, 160, False, True),    # the longest RBP sequence of the reference's run at its head dim (1280 / 8):
                                       # current/rna_clip_codes.ipynb:2340 ([32, 2542, 1280])
    (2, 600, 8, 160, False, True),
])
def test_attention_fwd_bwd(dev, B, L, H, D, use_rope, use_mask):
    ops = _ops()
    qkv = _rand((B * L, 3 * H * D), dev, 30, 1.0, dtype=torch.bfloat16)
    scale = D ** -0.5
    rope = _rope_tables(L, D, dev) if use_rope else None
    mask = None
    if use_mask:
        lens = torch.randint(L // 3, L + 1, (B,), generator=torch.Generator().manual_seed(31))
        lens[0] = L
        mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(dev).contiguous()
    qf = qkv.float().requires_grad_(True)
    ref, lse_ref = _attn_ref(qf, B, L, H, D, mask, rope, scale)
    out, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=rope, q_scale=scale)
    err = (out.float() - ref.detach()).abs().max().item()
    assert err < 3e-2, f"fwd max err {err}"                   # bf16 P and bf16 output rounding
    assert torch.allclose(lse, lse_ref.detach(), rtol=1e-3, atol=2e-2), (lse - lse_ref).abs().max()
    dout = _rand((B * L, H * D), dev, 32, 1.0, dtype=torch.bfloat16)
    if mask is not None:      # gradients flowing from padded query rows are garbage-in; zero them like a pooled loss does
        dout = (dout.view(B, L, -1) * mask[..., None].to(dout.dtype)).view(B * L, -1).contiguous()
    gref, = torch.autograd.grad(ref, qf, dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=mask, rope=rope, q_scale=scale)
    g = dqkv.float()
    if mask is not None:      # padded key rows get no gradient from valid queries except through dq of padded rows
        m3 = mask.view(B * L, 1).float()
        g, gref = g * m3, gref * m3
    denom = gref.abs().max().item()
    rel = (g - gref).abs().max().item() / denom
    assert rel < 4e-2, f"bwd max rel err {rel}"
    # determinism: same inputs, bitwise same gradient
    dqkv2 = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=mask, rope=rope, q_scale=scale)
    assert torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize("B,L,H,D,use_rope", [(8, 256, 20, 24, True), (3, 190, 4, 32, False), (2, 256, 3, 16, True),
                                                 (2, 129, 2, 24, True), (5, 255, 3, 32, True), (1, 256, 1, 24, False)])
@pytest.mark.parametrize("waves", ["4", "8"])
def test_attention_bwd_whole_head_vs_two_kernel(dev, B, L, H, D, use_rope, waves, kopt):
    """The whole-head backward (one workgroup per head, 5 products) against the dQ + dK/dV kernel pair (7 products)
    on the same inputs: same arithmetic up to the f32 summation order over query / key blocks, so they agree to bf16 rounding."""
    ops = _ops()
    qkv = _rand((B * L, 3 * H * D), dev, 70, 1.0, dtype=torch.bfloat16)
    dout = _rand((B * L, H * D), dev, 71, 1.0, dtype=torch.bfloat16)
    scale = D ** -0.5
    rope = _rope_tables(L, D, dev) if use_rope else None
    lens = torch.tensor([L] + [max(1, L - 23 * (i + 1)) for i in range(B - 1)])
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).contiguous().to(dev)
    dout = (dout.view(B, L, -1) * mask[..., None].to(dout.dtype)).view(B * L, -1).contiguous()
    if use_rope:
        qkv = ops.rope_qk_(qkv, B, L, H, D, rope)               # the whole-head kernel wants q / k rotated already
    out, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=mask, rope=None, q_scale=scale)
    kw = dict(key_mask=mask, rope=rope, q_scale=scale, prerotated=use_rope)
    kopt("attn_fused_bwd", 0)
    g2 = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float()
    kopt("attn_fused_bwd", 1)
    kopt("attn_fused_waves", int(waves))      # 4 (default): two 256-thread workgroups per CU; 8: one of 512
    g1 = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float()
    assert torch.equal(g1, ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float())      # reproducible
    m3 = mask.view(B * L, 1).float()
    g1, g2 = g1 * m3, g2 * m3
    assert torch.isfinite(g1).all()
    denom = g2.abs().max().item()
    assert (g1 - g2).abs().max().item() / denom < 8e-3       # one bf16 ulp of the largest gradient


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("B,L,H", [(8, 256, 8), (3, 190, 4), (2, 129, 2), (5, 255, 3), (1, 256, 1), (16, 233, 8)])
def test_attention_bwd_whole_head_hd96_vs_two_kernel(dev, B, L, H, waves, kopt):
    """The whole-head backward for the RNA encoder's heads (hd 96, 128 < L <= 256: one workgroup per head, every operand
    read once, 5 products) against the dQ + dK/dV pair and against f32 autograd: ragged lengths + key padding, the
    delta = rowsum(dO * O) side output, bitwise reproducibility.  waves: four waves of 64 keys or eight of 32
    (option attn_fused_waves)."""
    ops = _ops()
    kopt("attn_fused_waves", waves)
    D = 96
    qkv = _rand((B * L, 3 * H * D), dev, 170, 1.0, dtype=torch.bfloat16)
    dout = _rand((B * L, H * D), dev, 171, 1.0, dtype=torch.bfloat16)
    scale = D ** -0.5
    lens = torch.tensor([L] + [max(1, L - 37 * (i + 1)) for i in range(B - 1)])
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).contiguous().to(dev)
    dout = (dout.view(B, L, -1) * mask[..., None].to(dout.dtype)).view(B * L, -1).contiguous()
    for m in (mask, None):
        out, lse = ops.attn_fwd(qkv, B, L, H, D, key_mask=m, rope=None, q_scale=scale)
        kw = dict(key_mask=m, rope=None, q_scale=scale)
        kopt("attn_fused_bwd", 0)
        g2 = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float()
        kopt("attn_fused_bwd", 1)
        g1 = ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float()
        assert torch.equal(g1, ops.attn_bwd(qkv, out, dout, lse, B, L, H, D, **kw).float())      # reproducible
        m3 = (m if m is not None else torch.ones_like(mask)).view(B * L, 1).float()
        g1, g2 = g1 * m3, g2 * m3
        assert torch.isfinite(g1).all()
        denom = g2.abs().max().item()
        assert (g1 - g2).abs().max().item() / denom < 8e-3, (g1 - g2).abs().max().item() / denom
        # ... and against f32 autograd on the same bf16-rounded inputs
        qf = qkv.float().requires_grad_(True)
        ref, _ = _attn_ref(qf, B, L, H, D, m, None, scale)
        gref, = torch.autograd.grad(ref, qf, dout.float())
        gref = gref * m3
        assert (g1 - gref).abs().max().item() / gref.abs().max().item() < 4e-2


@pytest.mark.parametrize("B,L,H,D", [(8, 256, 20, 24), (3, 190, 4, 32), (2, 129, 3, 16), (5, 255, 2, 24)])
def test_attention_fwd_whole_head_equals_general(dev, B, L, H, D, kopt):
    """Short heads whose rows need no rotation take the whole-head forward (K / V staged once per head); it keeps the
    general kernel's tiles, sub-block order and arithmetic, so outputs and LSE are bit-identical."""
    ops = _ops()
    qkv = _rand((B * L, 3 * H * D), dev, 80, 1.0, dtype=torch.bfloat16)
    lens = torch.tensor([L] + [max(1, L - 29 * (i + 1)) for i in range(B - 1)])
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).contiguous().to(dev)
    for m in (None, mask):
        kopt("attn_whole_fwd", 0)
        o0, l0 = ops.attn_fwd(qkv, B, L, H, D, key_mask=m, rope=None, q_scale=D ** -0.5)
        kopt("attn_whole_fwd", 1)
        o1, l1 = ops.attn_fwd(qkv, B, L, H, D, key_mask=m, rope=None, q_scale=D ** -0.5)
        assert torch.equal(o0, o1) and torch.equal(l0, l1)


@pytest.mark.parametrize("B,L,H,D", [(8, 256, 20, 24), (3, 190, 4, 32), (2, 129, 3, 16), (2, 100, 4, 24), (1, 300, 2, 64)])
def test_attention_fwd_rot_equals_rope_then_fwd(dev, B, L, H, D):
    """clipk_attn_fwd_rot == clipk_rope_qk + clipk_attn_fwd(rope = NULL), bit for bit: the rotated q / k left in the
    buffer, the output and the LSE.  Short heads at 128 < L <= 256 take the one-kernel path, the rest the two calls."""
    ops = _ops()
    qkv = _rand((B * L, 3 * H * D), dev, 90, 1.0, dtype=torch.bfloat16)
    rope = _rope_tables(L, D, dev)
    lens = torch.tensor([L] + [max(1, L - 31 * (i + 1)) for i in range(B - 1)])
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).contiguous().to(dev)
    for m in (None, mask):
        a = ops.rope_qk_(qkv.clone(), B, L, H, D, rope)
        o0, l0 = ops.attn_fwd(a, B, L, H, D, key_mask=m, rope=None, q_scale=D ** -0.5)
        b = qkv.clone()
        o1, l1 = ops.attn_fwd_rot_(b, B, L, H, D, rope, key_mask=m, q_scale=D ** -0.5)
        assert torch.equal(a, b)
        assert torch.equal(o0, o1) and torch.equal(l0, l1)


# ------------------------------------------------------------------------------------------------ misc
def test_cast_and_transpose(dev):
    ops = _ops()
    w = _rand((200, 136), dev, 40)
    wb, wt = ops.cast_transpose(w)
    assert torch.equal(wb, w.to(torch.bfloat16))
    assert torch.equal(wt, w.to(torch.bfloat16).t().contiguous())
    x = _rand((1000, 33), dev, 41).contiguous()
    assert torch.equal(ops.to_bf16(x), x.to(torch.bfloat16))
    assert torch.equal(ops.to_f32(x.to(torch.bfloat16)), x.to(torch.bfloat16).float())


def test_embed_pool(dev):
    ops = _ops()
    B, L, d, V = 4, 50, 480, 33
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(42)).to(dev)
    ids[0, 3] = 32
    table = _rand((V, d), dev, 43)
    mask = (torch.arange(L)[None] < torch.tensor([50, 40, 30, 50])[:, None]).to(torch.uint8).to(dev).contiguous()
    rs = torch.tensor([0.88, 1.0, 0.9, 1.1], device=dev)
    x = ops.embed_fwd(ids, table, row_scale=rs, mask=mask.view(-1), mask_token_id=32)
    ref = table[ids] * rs[:, None, None] * mask[..., None] * (ids != 32)[..., None]
    assert torch.allclose(x.view(B, L, d), ref, rtol=1e-6, atol=1e-7)
    dx = _rand((B * L, d), dev, 44)
    dt = torch.zeros_like(table)
    ops.embed_bwd(ids, dx, dt, row_scale=rs, mask=mask.view(-1), mask_token_id=32)
    w = (rs[:, None] * mask * (ids != 32)).view(-1, 1)
    dref = torch.zeros_like(table).index_add_(0, ids.view(-1), dx * w)
    assert torch.allclose(dt, dref, rtol=1e-4, atol=1e-4)
    xs = _rand((B * L, d), dev, 45)
    for mode in (0, 1):
        y = ops.pool_fwd(xs, B, L, mask=mask.view(-1), mode=mode)
        x3 = xs.view(B, L, d)
        yref = x3[:, 0] if mode == 0 else (x3 * mask[..., None]).sum(1) / mask.sum(1, keepdim=True)
        assert torch.allclose(y, yref, rtol=1e-5, atol=1e-6)
        dy = _rand((B, d), dev, 46)
        dxp = ops.pool_bwd(dy, B, L, mask=mask.view(-1), mode=mode).view(B, L, d)
        if mode == 0:
            dref = torch.zeros_like(x3); dref[:, 0] = dy
        else:
            dref = dy[:, None] * mask[..., None] / mask.sum(1, keepdim=True)[..., None]
        assert torch.allclose(dxp, dref, rtol=1e-5, atol=1e-6)


def test_adamw_clip(dev):
    ops = _ops()
    n = 100003
    w = _rand((n,), dev, 50); g = _rand((n,), dev, 51, 3.0)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    wk = w.clone()
    for step in (1, 2, 3):
        p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        opt.step()
        nsq = ops.sumsq(g)
        assert abs(nsq.item() - (g.double() ** 2).sum().item()) / (g.double() ** 2).sum().item() < 1e-5
        ops.adamw_step(wk, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, grad_norm_sq=nsq, max_norm=1.0)
        assert torch.allclose(wk, p.detach(), rtol=1e-5, atol=1e-6), (wk - p).abs().max()


# ------------------------------------------------------------------------------------------------ materialised logits
@pytest.mark.parametrize("B,Nc", [(256, 0), (128, 200), (1000, 37), (1536, 0)])
def test_ce_logits_kernels_vs_torch(dev, B, Nc):
    """clipk_ce_logits_{lse,bwd} (F.cross_entropy on the logits the module API returns: old/ablation.py:16,
    rna_clip_codes.ipynb:1952-1953, old/clip_opt.py:130-151 with cache columns) against f64 torch autograd."""
    from clip_dplm_amd import functional as KF
    g = torch.Generator().manual_seed(B + Nc)
    S = (torch.randn(B, B, generator=g) * 4).to(dev).requires_grad_(True)
    S2 = (torch.randn(B, Nc, generator=g) * 4).to(dev).requires_grad_(True) if Nc else None
    lab = torch.arange(B, device=dev)
    for symmetric in (True, False):
        Sd = S.detach().double().requires_grad_(True)
        S2d = S2.detach().double().requires_grad_(True) if Nc else None
        rows = Sd if not Nc else torch.cat([Sd, S2d], 1)
        ref = F.cross_entropy(rows, lab)
        if symmetric:
            ref = 0.5 * (ref + F.cross_entropy(Sd.t(), lab))
        (ref * 1.7).backward()
        S.grad = None
        if Nc:
            S2.grad = None
        loss = KF.cross_entropy_diag(S, S2, symmetric=symmetric)
        assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
        (loss * 1.7).backward()
        assert torch.allclose(S.grad.double(), Sd.grad, rtol=1e-4, atol=1e-8), (S.grad.double() - Sd.grad).abs().max()
        if Nc:
            assert torch.allclose(S2.grad.double(), S2d.grad, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("M,N,P", [(256, 256, 128), (100, 300, 64), (1100, 1100, 512)])
def test_sim_logits_backward_on_kernels(dev, M, N, P):
    """SimLogitsFn: forward clipk_sim_logits, backward exact-f32 MFMA products on transposed operands (no rocBLAS):
    gradients of sum(logits * G) w.r.t. a, b and the scale against f64 autograd; contraction lengths > 512 take the
    chunked path (M = N = 1100 is not a multiple of 4 either)."""
    from clip_dplm_amd import functional as KF
    a = _unit((M, P), dev, 21).requires_grad_(True)
    b = _unit((N, P), dev, 22).requires_grad_(True)
    sc = torch.tensor(14.2849, device=dev, requires_grad=True)
    G = _rand((M, N), dev, 23)
    (KF.sim_logits(a, b, sc) * G).sum().backward()
    ad, bd = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    sd = sc.detach().double().requires_grad_(True)
    ((ad @ bd.t()) * sd * G.double()).sum().backward()
    assert torch.allclose(a.grad.double(), ad.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(b.grad.double(), bd.grad, rtol=1e-4, atol=1e-4)
    assert abs(sc.grad.item() - sd.grad.item()) < 1e-4 * max(1.0, abs(sd.grad.item()))


def test_transpose_scale_f32(dev):
    from clip_dplm_amd import ops
    x = _rand((333, 130), dev, 24)
    s = torch.tensor([2.5], device=dev)
    assert torch.equal(ops.transpose_scale_f32(x), x.t().contiguous())
    assert torch.allclose(ops.transpose_scale_f32(x, s), x.t() * 2.5, rtol=1e-6, atol=0)


def test_options_api(dev):
    """clipk_set_option / get / reset: explicit kernel selection, unknown names and result-changing ablations refused."""
    from clip_dplm_amd import _ffi, ops
    assert ops.get_option("gemm_kernel") == -1
    ops.set_option("gemm_kernel", 2)
    assert ops.get_option("gemm_kernel") == 2
    ops.reset_options()
    assert ops.get_option("gemm_kernel") == -1
    with pytest.raises(_ffi.ClipkError):
        ops.set_option("no_such_option", 1)
    with pytest.raises(_ffi.ClipkError):
        ops.set_option("gemm_abl", 1)                       # timing ablations change results: experiment builds only


# ------------------------------------------------------------------------------------------------ packed varlen attention
@pytest.mark.parametrize("H,D,use_rope,lens", [(8, 96, False, [256, 131, 40, 700, 1, 129]),
                                                (20, 24, True, [256, 200, 37, 129]),
                                                (4, 64, True, [1024, 300, 513]),
                                                (3, 32, False, [90, 17, 128]),
                                                (2, 160, False, [140, 65])])
def test_attention_varlen_equals_padded_with_mask(dev, H, D, use_rope, lens):
    """clipk_attn_varlen_{fwd,bwd} on a packed batch against clipk_attn_{fwd,bwd} on the same sequences padded to the
    longest with a key mask (general kernels on both sides: option attn_*_whole = 0 is not needed because the padded
    reference may take the whole-head kernels, which are bit-identical to the general ones for the forward).  Outputs,
    LSE and gradients of the real rows."""
    ops = _ops()
    B, Lm = len(lens), max(lens)
    T = sum(lens)
    g = torch.Generator().manual_seed(T + D)
    qkv_pad = (torch.randn(B, Lm, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16)
    dout_pad = (torch.randn(B, Lm, H * D, generator=g) * 0.5).to(torch.bfloat16)
    mask = torch.zeros(B, Lm, dtype=torch.uint8)
    for i, l in enumerate(lens):
        mask[i, :l] = 1
    dout_pad = dout_pad * mask[..., None].to(torch.bfloat16)
    sel = mask.bool().view(-1)
    qkv_pk = qkv_pad.view(B * Lm, -1)[sel].contiguous().to(dev)
    dout_pk = dout_pad.view(B * Lm, -1)[sel].contiguous().to(dev)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens), 0)
    cu = cu.to(dev)
    rope = None
    if use_rope:
        inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        fr = torch.arange(Lm, dtype=torch.float32)[:, None] * inv[None]
        rope = (fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev))
    qs = D ** -0.5
    qp, dp, mk = qkv_pad.view(B * Lm, -1).contiguous().to(dev), dout_pad.view(B * Lm, -1).contiguous().to(dev), mask.to(dev)
    o_ref, lse_ref = ops.attn_fwd(qp, B, Lm, H, D, key_mask=mk, rope=rope, q_scale=qs)
    g_ref = ops.attn_bwd(qp, o_ref, dp, lse_ref, B, Lm, H, D, key_mask=mk, rope=rope, q_scale=qs)
    o, lse = ops.attn_varlen_fwd(qkv_pk, cu, Lm, H, D, rope=rope, q_scale=qs)
    gk = ops.attn_varlen_bwd(qkv_pk, o, dout_pk, lse, cu, Lm, H, D, rope=rope, q_scale=qs)
    seld = sel.to(dev)
    assert torch.allclose(o.float(), o_ref[seld].float(), rtol=0, atol=4e-3), (o.float() - o_ref[seld].float()).abs().max()
    lse_ref_pk = lse_ref.permute(1, 0, 2).reshape(H, B * Lm)[:, seld]
    assert torch.allclose(lse, lse_ref_pk, rtol=0, atol=1e-4), (lse - lse_ref_pk).abs().max()
    gr = g_ref[seld].float()
    assert torch.isfinite(gk.float()).all()
    assert (gk.float() - gr).abs().max().item() < 8e-3 * max(1.0, gr.abs().max().item())


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("H,lens", [(8, [256, 131, 40, 200, 1, 129]), (2, [255, 64, 256])])
def test_attention_varlen_whole_head_hd96(dev, kopt, H, lens, waves):
    """Packed batches whose longest sequence fits the hd-96 whole-head backward (one workgroup per (sequence, head), rows
    and length from cu_seqlens) against the general varlen kernels."""
    ops = _ops()
    kopt("attn_fused_waves", waves)
    D = 96
    B, Lm, T = len(lens), max(lens), sum(lens)
    g = torch.Generator().manual_seed(T + D)
    qkv = (torch.randn(T, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dout = (torch.randn(T, H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens), 0)
    cu = cu.to(dev)
    qs = D ** -0.5
    o, lse = ops.attn_varlen_fwd(qkv, cu, Lm, H, D, rope=None, q_scale=qs)
    g1 = ops.attn_varlen_bwd(qkv, o, dout, lse, cu, Lm, H, D, rope=None, q_scale=qs)
    kopt("attn_fused_bwd", 0)
    g0 = ops.attn_varlen_bwd(qkv, o, dout, lse, cu, Lm, H, D, rope=None, q_scale=qs)
    assert torch.isfinite(g1.float()).all()
    assert (g1.float() - g0.float()).abs().max().item() < 8e-3 * max(1.0, g0.float().abs().max().item())


@pytest.mark.parametrize("H,D,lens", [(20, 24, [256, 200, 37, 129, 1, 64]), (4, 32, [130, 255, 3]), (6, 16, [256, 256, 17])])
def test_attention_varlen_whole_head_kernels(dev, kopt, H, D, lens):
    """Packed batches of short sequences through the whole-head kernels: clipk_attn_varlen_fwd_rot (q / k rotated in
    place while staged, positions per sequence) + clipk_attn_varlen_bwd(prerotated = 1) (one fused kernel per
    (sequence, head)) against the general varlen kernels that rotate at staging; then the same without RoPE, where the
    dispatcher picks the whole-head kernels by itself (options attn_whole_fwd / attn_fused_bwd = 0: the general ones)."""
    ops = _ops()
    B, Lm, T = len(lens), max(lens), sum(lens)
    g = torch.Generator().manual_seed(T + D)
    qkv = (torch.randn(T, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dout = (torch.randn(T, H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens), 0)
    cu = cu.to(dev)
    rope = _rope_tables(Lm, D, dev)
    qs = D ** -0.5
    o_ref, lse_ref = ops.attn_varlen_fwd(qkv, cu, Lm, H, D, rope=rope, q_scale=qs)
    g_ref = ops.attn_varlen_bwd(qkv, o_ref, dout, lse_ref, cu, Lm, H, D, rope=rope, q_scale=qs)
    qr = qkv.clone()
    o, lse = ops.attn_varlen_fwd_rot_(qr, cu, Lm, H, D, rope, q_scale=qs)
    # q / k rotated in place, per sequence from position 0; v untouched
    off = 0
    for l in lens:
        exp = ops.rope_qk_(qkv[off:off + l].clone().contiguous(), 1, l, H, D, (rope[0][:l].contiguous(), rope[1][:l].contiguous()))
        assert torch.equal(qr[off:off + l], exp)
        off += l
    assert torch.allclose(o.float(), o_ref.float(), rtol=0, atol=8e-3), (o.float() - o_ref.float()).abs().max()
    assert torch.allclose(lse, lse_ref, rtol=0, atol=5e-3), (lse - lse_ref).abs().max()
    gk = ops.attn_varlen_bwd(qr, o, dout, lse, cu, Lm, H, D, rope=rope, q_scale=qs, prerotated=True)
    assert torch.isfinite(gk.float()).all()
    assert (gk.float() - g_ref.float()).abs().max().item() < 1.5e-2 * max(1.0, g_ref.float().abs().max().item())
    # no RoPE: same packed batch, whole-head kernels (default dispatch) vs general kernels
    o1, l1 = ops.attn_varlen_fwd(qkv, cu, Lm, H, D, rope=None, q_scale=qs)
    g1 = ops.attn_varlen_bwd(qkv, o1, dout, l1, cu, Lm, H, D, rope=None, q_scale=qs)
    kopt("attn_whole_fwd", 0)
    kopt("attn_fused_bwd", 0)
    o0, l0 = ops.attn_varlen_fwd(qkv, cu, Lm, H, D, rope=None, q_scale=qs)
    g0 = ops.attn_varlen_bwd(qkv, o0, dout, l0, cu, Lm, H, D, rope=None, q_scale=qs)
    assert torch.equal(o1, o0) and torch.equal(l1, l0)          # the whole-head forward is bit-identical to the general one
    assert (g1.float() - g0.float()).abs().max().item() < 8e-3 * max(1.0, g0.float().abs().max().item())


def test_pool_varlen(dev):
    ops = _ops()
    lens = [5, 1, 300, 64]
    T, d = sum(lens), 96
    x = _rand((T, d), dev, 41)
    cu = torch.tensor([0, 5, 6, 306, 370], dtype=torch.int32, device=dev)
    for mode in (0, 1):
        y = ops.pool_varlen_fwd(x, cu, mode)
        ref = torch.stack([x[a] if mode == 0 else x[a:b].mean(0) for a, b in zip(cu[:-1].tolist(), cu[1:].tolist())])
        assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)
        dy = _rand((4, d), dev, 42)
        dx = ops.pool_varlen_bwd(dy, cu, T, mode)
        xr = x.clone().requires_grad_(True)
        yr = torch.stack([xr[a] if mode == 0 else xr[a:b].mean(0) for a, b in zip(cu[:-1].tolist(), cu[1:].tolist())])
        yr.backward(dy)
        assert torch.allclose(dx, xr.grad, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ tiled f32 GEMM
@pytest.mark.parametrize("M,N,K", [(4096, 512, 512), (64, 256, 512), (1030, 130, 66), (300, 64, 1030), (17, 8, 4),
                                   (2048, 768, 2050),
                                   # M <= 64 rows against a large weight: the skinny kernel (16-way split of the contraction
                                   # inside the workgroup) for the untransposed-A layouts - the sliced notebook model's shapes
                                   (32, 3840, 1280), (32, 1280, 5120), (1, 40, 256), (33, 1000, 1284), (64, 520, 2560),
                                   (17, 31, 260), (32, 480, 120)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_f32_all_layouts(dev, M, N, K, ta, tb):
    """clipk_gemm_f32 (tiled v_mfma_f32_32x32x2_f32; skinny form for M <= 64, K >= 256): all four operand layouts, ragged
    M / N / K (K % 16, % 4 != 0, odd leading dimensions), bias and scaled addend, against f64; exact-f32 products: error
    ~1e-7 * sum |a b|."""
    ops = _ops()
    a = _rand((K, M) if ta else (M, K), dev, 51)
    b = _rand((K, N) if tb else (N, K), dev, 52, 0.1)
    bias, add = _rand((N,), dev, 53), _rand((M, N), dev, 54)
    sc = torch.tensor([0.37], device=dev)
    A = a.double().t() if ta else a.double()
    Bm = b.double() if tb else b.double().t()
    ref = A @ Bm
    bound = 4e-7 * (A.abs() @ Bm.abs()) + 1e-6
    out = ops.gemm_f32(a, b, trans_a=ta, trans_b=tb)
    assert ((out.double() - ref).abs() <= bound).all(), (out.double() - ref).abs().max()
    out2 = ops.gemm_f32(a, b, trans_a=ta, trans_b=tb, bias=bias, addend=add, addend_scale=sc)
    ref2 = ref + bias.double() + 0.37 * add.double()
    assert ((out2.double() - ref2).abs() <= bound + 1e-6).all()
    assert torch.equal(out, ops.gemm_f32(a, b, trans_a=ta, trans_b=tb))        # deterministic


@pytest.mark.parametrize("M,N,K", [(32, 1280, 1280), (32, 512, 1280), (7, 40, 100), (33, 65, 129), (64, 96, 200),
                                   (1, 32, 32), (100, 64, 72)])
def test_gemm_wgrad_f32_one_launch_for_few_rows(dev, M, N, K):
    """clipk_gemm_wgrad_f32: dW = dy^T x and db = dy.sum(0) of an exact-f32 Linear (reference: autograd of nn.Linear,
    old/clip.py:11; rna_clip_codes.ipynb:1911-1954 at M = batch rows).  M <= 64 is the one-pass kernel (ragged M / N / K, both
    MT forms), M = 100 the tiled fallback; dW must be bit-identical to clipk_gemm_f32 on contraction-major operands (same
    contraction order), db against f64; accumulate adds into existing buffers, also through a row stride."""
    ops = _ops()
    dy, x = _rand((M, N), dev, 71), _rand((M, K), dev, 72)
    dw, db = ops.gemm_wgrad_f32(dy, x)
    assert torch.equal(dw, ops.gemm_f32(dy, x, trans_a=True, trans_b=True))
    ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
    assert ((dw.double() - ref_w).abs() <= 4e-7 * (dy.double().abs().t() @ x.double().abs()) + 1e-6).all()
    assert ((db.double() - ref_b).abs() <= 4e-7 * dy.double().abs().sum(0) + 1e-6).all()
    # accumulate into a strided dW (a slice of a wider buffer, like a parameter's .grad view) and an existing db
    wide = _rand((N, K + 4), dev, 73)
    w0, b0 = wide.clone(), _rand((N,), dev, 74)
    bacc = b0.clone()
    ops.gemm_wgrad_f32(dy, x, dw=wide[:, :K], dbias=bacc, accumulate=True)
    assert torch.equal(wide[:, :K], ops.gemm_f32(dy, x, trans_a=True, trans_b=True, addend=w0[:, :K]))
    assert torch.equal(wide[:, K:], w0[:, K:])                                  # nothing written past the K columns
    assert ((bacc.double() - (b0.double() + ref_b)).abs() <= 4e-7 * dy.double().abs().sum(0) + 2e-6).all()
    # overwrite form and the weight-only form
    ops.gemm_wgrad_f32(dy, x, dw=wide[:, :K], dbias=bacc, accumulate=False)
    assert torch.equal(wide[:, :K], dw) and torch.equal(bacc, db)
    dw2, none = ops.gemm_wgrad_f32(dy, x, want_bias=False)
    assert none is None and torch.equal(dw2, dw)


def test_dropout_epoch_word_shifts_the_seed_of_every_site(dev):
    """clipk_set_dropout_epoch (hipGraph replay of a training step with nn.TransformerEncoderLayer's dropout active,
    rna_clip_codes.ipynb:1915): while a device word e is registered every dropout site must behave exactly as with
    seed + e * 0x9E3779B9 and nothing registered; after clearing it the plain seed is back.  Sites: f32 dropout, GEMM
    epilogue, bf16 and f32 attention (forward and backward), LayerNorm-backward's bf16 output."""
    ops = _ops()
    p, seed, e = 0.25, 1234567, 3
    seed2 = (seed + e * 0x9E3779B9) & 0xFFFFFFFF
    B, L, H, D = 2, 64, 2, 32
    x = _rand((96, 160), dev, 81)
    a16, b16 = _rand((96, 64), dev, 82).to(torch.bfloat16), _rand((160, 64), dev, 83).to(torch.bfloat16)
    qkv16 = _rand((B * L, 3 * H * D), dev, 84).to(torch.bfloat16)
    do16 = _rand((B * L, H * D), dev, 85).to(torch.bfloat16)
    qkv32, do32 = qkv16.float(), do16.float()
    gamma, beta = _rand((160,), dev, 86), _rand((160,), dev, 87)
    _, _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)

    def run(sd):
        out = [ops.dropout_f32(x, (p, sd)), ops.gemm_nt(a16, b16, dropout=(p, sd))]
        o, lse = ops.attn_fwd(qkv16, B, L, H, D, q_scale=D ** -0.5, dropout=(p, sd))
        out += [o, ops.attn_bwd(qkv16, o, do16, lse, B, L, H, D, q_scale=D ** -0.5, dropout=(p, sd))]
        o32, lse32 = ops.attn_f32_fwd(qkv32, B, L, H, D, q_scale=D ** -0.5, dropout=(p, sd))
        out += [o32, ops.attn_f32_bwd(qkv32, o32, do32, lse32, B, L, H, D, q_scale=D ** -0.5, dropout=(p, sd))]
        out.append(ops.layernorm_bwd(x, x, gamma, beta, mean, rstd, want_f32=False, want_bf16=True, dropout_bf16=(p, sd))[1])
        return [t.clone() for t in out]
    plain, shifted = run(seed), run(seed2)
    assert not torch.equal(plain[0], shifted[0])
    word = torch.tensor([e], dtype=torch.int32, device=dev)
    ops.set_dropout_epoch(word)
    try:
        with_word = run(seed)
        word.fill_(0)
        zero_word = run(seed)
    finally:
        ops.set_dropout_epoch(None)
    for i, (a, b) in enumerate(zip(with_word, shifted)):
        assert torch.equal(a, b), f"site {i}: epoch word != shifted seed"
    for i, (a, b) in enumerate(zip(zero_word, plain)):
        assert torch.equal(a, b), f"site {i}: epoch 0 != plain seed"
    for i, (a, b) in enumerate(zip(run(seed), plain)):
        assert torch.equal(a, b), f"site {i}: cleared word still in effect"


def test_embed_fwd_out_of_range_id_is_loud(dev):
    """An id outside the table must never read memory: its row is NaN, every other row is exact."""
    ops = _ops()
    table = _rand((33, 64), dev, 61)
    ids = torch.tensor([[0, 5, 32, 40], [7, -1, 2, 3]], device=dev)
    x = ops.embed_fwd(ids, table).view(2, 4, 64)
    assert torch.isnan(x[0, 3]).all() and torch.isnan(x[1, 1]).all()
    assert torch.equal(x[0, 1], table[5]) and torch.equal(x[1, 3], table[3])


def test_gemm_nt_output_over_2gib_is_cut_into_slabs(dev):
    """ESM-2-650M at B = 256, L = 1024 writes a 262144 x 5120 bf16 FFN activation (2.7 GB): beyond the 32-bit byte offsets of
    the buffer-descriptor epilogues.  clipk_gemm_nt cuts such problems along M and keeps the specialised kernels:
    results equal a direct f32 product, rows on both sides of the cut included; GELU with and without the saved
    pre-activation (frozen encoders keep none)."""
    ops = _ops()
    M, N, K = 270336, 4096, 192
    a = _rand((M, K), dev, 71, dtype=torch.bfloat16)
    b = _rand((N, K), dev, 72, 0.05, dtype=torch.bfloat16)
    bias = _rand((N,), dev, 73)
    g, u = ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True)
    g2 = ops.gemm_nt(a, b, bias=bias, act="gelu")
    assert torch.equal(g, g2)
    for lo in (0, 209664 - 128, 209664, M - 256):               # around the slab boundary and at both ends
        ref = a[lo:lo + 256].float() @ b.float().t() + bias
        assert torch.allclose(u[lo:lo + 256].float(), ref, rtol=1e-2, atol=1e-2)
        assert torch.allclose(g[lo:lo + 256].float(), F.gelu(ref), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("xdt", ["f32", "bf16"])
@pytest.mark.parametrize("B,L,cols,masked", [(5, 37, 480, True), (3, 256, 768, False), (4, 64, 1280, True), (2, 9, 64, True)])
def test_layernorm_meanpool_equals_layernorm_then_pool(dev, B, L, cols, masked, xdt):
    """clipk_layernorm_meanpool_{fwd,bwd}: the encoders' final LayerNorm + masked mean in one pass against the two
    separate kernels (statistics bit for bit; sums to f32 rounding: another summation order), ragged masks and a sample
    with no valid row (pooled 0, no gradient)."""
    ops = _ops()
    x = _rand((B * L, cols), dev, 101, 1.5) + 0.3
    if xdt == "bf16":                                           # post-LN stacks hand the last layer's bf16 output over
        x = x.to(torch.bfloat16)
    g, b = _rand((cols,), dev, 102) * 0.2 + 1.0, _rand((cols,), dev, 103, 0.1)
    mask = None
    if masked:
        lens = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(5))
        lens[0] = L
        if B > 2:
            lens[2] = 0                                         # nothing valid
        mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(dev).contiguous()
    pooled, mean, rstd, wrow = ops.layernorm_meanpool_fwd(x, g, b, 1e-5, B, L, mask=None if mask is None else mask.view(-1))
    y, _, m2, r2 = ops.layernorm_fwd(x, g, b, 1e-5)
    assert torch.equal(mean, m2) and torch.equal(rstd, r2)
    ref = ops.pool_fwd(y, B, L, mask=mask, mode=1)
    assert torch.allclose(pooled, ref, rtol=1e-5, atol=1e-6), (pooled - ref).abs().max()
    if masked and B > 2:
        assert (pooled[2] == 0).all() and (wrow.view(B, L)[2] == 0).all()
    dp = _rand((B, cols), dev, 104)
    dy = ops.pool_bwd(dp, B, L, mask=mask, mode=1)
    rx, rxb, rg, rb = ops.layernorm_bwd(dy, x, g, None, m2, r2, want_f32=True, want_bf16=True)
    dx, dxb, dg, db = ops.layernorm_meanpool_bwd(dp, wrow, x, g, mean, rstd, B, L, want_f32=True, want_bf16=True)
    assert torch.allclose(dx, rx, rtol=1e-5, atol=1e-7), (dx - rx).abs().max()
    assert torch.allclose(dxb.float(), rxb.float(), rtol=1e-2, atol=1e-6)
    assert torch.allclose(dg, rg, rtol=1e-4, atol=1e-5) and torch.allclose(db, rb, rtol=1e-4, atol=1e-5)
    acc_g = torch.ones_like(dg)
    acc_b = torch.ones_like(db)
    ops.layernorm_meanpool_bwd(dp, wrow, x, g, mean, rstd, B, L, dgamma=acc_g, dbeta=acc_b, accumulate=True)
    assert torch.allclose(acc_g, dg + 1, rtol=1e-5, atol=1e-6) and torch.allclose(acc_b, db + 1, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kernel,M", [("v2", 1000), ("v3", 2048 + 24), ("v4", 2048 + 24)])
@pytest.mark.parametrize("H,D", [(3, 64), (4, 32), (8, 16)])
def test_gemm_nt_rope_epilogue(dev, kopt, kernel, M, H, D):
    """RoPE in the qkv projection's epilogue (EPI_ROPE): q and k thirds rotated on the f32 value (product + bias)
    before the one bf16 rounding, v third untouched; positions wrap at L (several sequences per launch, ragged M).
    Reference: f32 product of the same bf16 operands + rotate-half (transformers modeling_esm.py:48-52,74-79)."""
    import ops_emulator
    ops = _ops()
    kopt("gemm_kernel", _GEMM_KERNEL[kernel])
    d, L, K = H * D, 250, 192
    a = _rand((M, K), dev, 91, dtype=torch.bfloat16)
    w = _rand((3 * d, K), dev, 92, 0.08, dtype=torch.bfloat16)
    bias = _rand((3 * d,), dev, 93)
    cos, sin = _rope_tables(L, D, dev)
    out = ops.gemm_nt(a, w, bias=bias, rope=(cos, sin, L, D, 2 * d))
    ref = ops_emulator.gemm_nt(a.cpu(), w.cpu(), bias=bias.cpu(), out_dtype=torch.float32,
                               rope=(cos.cpu(), sin.cpu(), L, D, 2 * d))
    err = (out.float().cpu() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3).all(), float(err.max())       # one bf16 rounding of an f32 value
    plain = ops.gemm_nt(a, w, bias=bias)
    assert torch.equal(out[:, 2 * d:], plain[:, 2 * d:])                         # v: the plain epilogue's bits
    rot2 = ops.rope_qk_(plain.clone()[: (M // L) * L].contiguous(), M // L, L, H, D, (cos, sin))
    assert torch.allclose(out[: (M // L) * L].float(), rot2.float(), rtol=2e-2, atol=2e-2)   # vs the two-pass path


def test_gemm_nt_rope_epilogue_refuses_what_it_cannot_rotate(dev):
    """Heads that do not tile the 64-column wave slices (hd = 24) or a rotation combined with another epilogue must be
    an error, never an unrotated result."""
    from clip_dplm_amd._ffi import ClipkError
    ops = _ops()
    a = _rand((512, 64), dev, 94, dtype=torch.bfloat16)
    w = _rand((144, 64), dev, 95, dtype=torch.bfloat16)
    cos, sin = _rope_tables(128, 24, dev)
    with pytest.raises(ClipkError):
        ops.gemm_nt(a, w, rope=(cos, sin, 128, 24, 96))
    cos, sin = _rope_tables(128, 16, dev)
    with pytest.raises(ClipkError):
        ops.gemm_nt(a, w, act="gelu", rope=(cos, sin, 128, 16, 96))


@pytest.mark.parametrize("Mx,Ny,Nc,P,off", [(512, 4096, 0, 512, 1024), (100, 300, 0, 64, 0), (128, 128, 200, 128, 0),
                                            (1000, 1000, 37, 768, 0), (64, 64, 0, 36, 0)])
def test_simce_tiled_lse_pass(dev, kopt, Mx, Ny, Nc, P, off):
    """simce_tiled.hip (64 x 64 tiles, K-loop over P) against f64 and against the first-generation kernel: row LSE and
    positive logit, ragged query / key / P edges, cache columns, label offsets."""
    ops = _ops()
    x, y = _unit((Mx, P), dev, 81), _unit((Ny, P), dev, 82)
    cache = _unit((Nc, P), dev, 83) if Nc else None
    sc = torch.tensor([14.2849], device=dev)
    off = min(off, max(Ny - Mx, 0))
    keys = y if cache is None else torch.cat([y, cache], 0)
    S = (x.double() @ keys.double().t()) * 14.2849
    ref_lse = torch.logsumexp(S, 1)
    ref_pos = S[torch.arange(Mx, device=dev), torch.arange(Mx, device=dev) + off]
    out = {}
    for mode in (1, 2):
        kopt("simce_kernel", mode)
        lse, pos = ops.simce_lse(x, y, sc, label_offset=off, cache=cache)
        assert torch.allclose(lse.double(), ref_lse, rtol=0, atol=2e-5), (mode, (lse.double() - ref_lse).abs().max())
        assert torch.allclose(pos.double(), ref_pos, rtol=0, atol=2e-5), mode
        out[mode] = lse
        assert torch.equal(lse, ops.simce_lse(x, y, sc, label_offset=off, cache=cache)[0])      # deterministic
    assert torch.allclose(out[1], out[2], rtol=0, atol=1e-5)


@pytest.mark.parametrize("Mx,Ny,Nc,P,off,wr,wc", [(512, 4096, 0, 512, 1024, 0.5, 0.5), (100, 300, 0, 64, 0, 1.0, 0.0),
                                                  (128, 128, 200, 128, 0, 0.5, 0.5), (200, 200, 0, 36, 0, 0.5, 0.5),
                                                  (256, 256, 0, 384, 0, 0.5, 0.5)])
def test_simce_tiled_grad_pass(dev, kopt, Mx, Ny, Nc, P, off, wr, wc):
    """simce_tiled.hip gradient pass against f64 autograd of  w_row CE(rows of [S | S_cache]) + w_col CE(columns of S)
    restricted to this block's queries, and against the first-generation kernel: dX and the d scale partials."""
    ops = _ops()
    x, y = _unit((Mx, P), dev, 91), _unit((Ny, P), dev, 92)
    cache = _unit((Nc, P), dev, 93) if Nc else None
    sc = torch.tensor([14.2849], device=dev)
    off = min(off, max(Ny - Mx, 0))
    inv_bg = 1.0 / Ny
    # arbitrary (consistent) LSE vectors are enough to pin the kernel's arithmetic: use the true ones
    keys = y if cache is None else torch.cat([y, cache], 0)
    Sx = (x.double() @ keys.double().t()) * 14.2849
    lse_x = torch.logsumexp(Sx, 1).float().contiguous()
    allq = _unit((Ny, P), dev, 94)                                     # the other ranks' queries: only their LSE matters
    allq[off:off + Mx] = x
    lse_y = torch.logsumexp((allq.double() @ y.double().t()) * 14.2849, 0).float().contiguous()
    xd = x.double()
    S = (xd @ keys.double().t()) * 14.2849
    lab = torch.arange(Mx, device=dev) + off
    G = wr * torch.exp(S - lse_x.double()[:, None])
    G[:, :Ny] += wc * torch.exp(S[:, :Ny] - lse_y.double()[None, :])
    G[torch.arange(Mx, device=dev), lab] -= (wr + wc)
    G *= inv_bg
    ref_dx = (G @ keys.double()) * 14.2849
    ref_dsc = (G * (S / 14.2849)).sum(1)
    out = {}
    for mode in (1, 2):
        kopt("simce_kernel", mode)
        dx, dsc = ops.simce_grad(x, y, sc, lse_x, lse_y, wr, wc, inv_bg, label_offset=off, cache=cache)
        assert torch.allclose(dx.double(), ref_dx, rtol=1e-4, atol=1e-7), (mode, (dx.double() - ref_dx).abs().max())
        assert torch.allclose(dsc.double(), ref_dsc, rtol=1e-4, atol=1e-7), mode
        assert torch.equal(dx, ops.simce_grad(x, y, sc, lse_x, lse_y, wr, wc, inv_bg, label_offset=off, cache=cache)[0])
        # clipk_simce_grad_scaled: the loss' grad_output as a device scalar inside the kernel; 1.0 gives the same bits
        one, g = torch.ones(1, device=dev), torch.tensor([-2.5], device=dev)
        d1, s1 = ops.simce_grad(x, y, sc, lse_x, lse_y, wr, wc, inv_bg, label_offset=off, cache=cache, upstream=one)
        assert torch.equal(d1, dx) and torch.equal(s1, dsc)
        dg, sg = ops.simce_grad(x, y, sc, lse_x, lse_y, wr, wc, inv_bg, label_offset=off, cache=cache, upstream=g)
        assert torch.allclose(dg.double(), -2.5 * ref_dx, rtol=1e-4, atol=1e-7)
        assert torch.allclose(sg.double(), -2.5 * ref_dsc, rtol=1e-4, atol=1e-7)
        out[mode] = dx
    assert torch.allclose(out[1], out[2], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("n", [1, 32, 256, 1000, 4099])
def test_ce_combine_is_the_weighted_mean_of_the_two_directions(dev, n):
    """clipk_ce_combine: (w_row sum(lse_r - pos_r) + w_col sum(lse_c - pos_c)) / bg in one launch (the average of the two
    F.cross_entropy means, rna_clip_codes.ipynb:1952-1953) against f64; the one-sided form (old/ablation.py:16) with the
    column vectors absent."""
    ops = _ops()
    lr, pr, lc, pc = (_rand((n,), dev, 120 + i, 3.0) for i in range(4))
    got = ops.ce_combine(lr, pr, lc, pc, 0.5, 0.5, float(n))
    ref = (0.5 * (lr.double() - pr.double()).sum() + 0.5 * (lc.double() - pc.double()).sum()) / n
    assert got.shape == () and abs(got.item() - ref.item()) <= 1e-6 * max(1.0, (lr - pr).abs().double().sum().item() / n)
    one = ops.ce_combine(lr, pr, None, None, 1.0, 0.0, float(2 * n))
    ref1 = (lr.double() - pr.double()).sum() / (2 * n)
    assert abs(one.item() - ref1.item()) <= 1e-6 * max(1.0, (lr - pr).abs().double().sum().item() / n)
    assert torch.equal(got, ops.ce_combine(lr, pr, lc, pc, 0.5, 0.5, float(n)))


@pytest.mark.parametrize("rows,cols,act", [(64, 512, "celu"), (100, 256, "celu"), (33, 512, "softplus"), (257, 512, None),
                                            (16, 1024, "celu"), (5, 120, None)])
def test_layernorm_second_order_backward_vs_autograd(dev, rows, cols, act):
    """clipk_layernorm_bwd2 (the backward of the LayerNorm(+activation) backward: what training THROUGH the transport
    map T = dPsi/dx needs, triple_flow/2_icnn_core.py:181-211) against f64 torch autograd of the first backward."""
    ops = _ops()
    g_ = torch.Generator().manual_seed(rows + cols)
    a = torch.randn(rows, cols, generator=g_) * 1.5 + 0.3
    dy = torch.randn(rows, cols, generator=g_)
    g = torch.randn(rows, cols, generator=g_)
    gamma = torch.rand(cols, generator=g_) + 0.5
    beta = torch.randn(cols, generator=g_) * 0.3
    eps = 1e-5
    ad, dyd, gd, gmd, btd = (t.double().requires_grad_(True) for t in (a, dy, g, gamma, beta))
    mu = ad.mean(-1, keepdim=True)
    r = torch.rsqrt(((ad - mu) ** 2).mean(-1, keepdim=True) + eps)
    xh = (ad - mu) * r
    n = xh * gmd + btd
    y = n if act is None else (torch.nn.functional.celu(n) if act == "celu" else torch.nn.functional.softplus(n))
    da, = torch.autograd.grad(y, ad, dyd, create_graph=True)              # the first backward, as autograd builds it
    refs = torch.autograd.grad(da, [dyd, ad, gmd, btd], gd, allow_unused=True)
    _, _, mean, rstd = ops.layernorm_fwd(a.to(dev), gamma.to(dev), beta.to(dev), eps, act=act)
    da_k, _, _, _ = ops.layernorm_bwd(dy.to(dev), a.to(dev), gamma.to(dev), beta.to(dev), mean, rstd, act=act)
    assert (da_k.cpu().double() - da.detach()).abs().max().item() < 1e-4
    d_dy, d_a, d_gamma, d_beta = ops.layernorm_bwd2(g.to(dev), dy.to(dev), a.to(dev), gamma.to(dev), beta.to(dev), mean,
                                                    rstd, act=act)
    for name, got, ref in (("d_dy", d_dy, refs[0]), ("d_a", d_a, refs[1]), ("d_gamma", d_gamma, refs[2]),
                           ("d_beta", d_beta, refs[3])):
        ref = torch.zeros_like(got.cpu().double()) if ref is None else ref
        err = (got.cpu().double() - ref).abs().max().item()
        assert err < 2e-4 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())
    # accumulate into existing parameter gradients
    dg0, db0 = torch.ones(cols, device=dev), torch.full((cols,), 2.0, device=dev)
    ops.layernorm_bwd2(g.to(dev), dy.to(dev), a.to(dev), gamma.to(dev), beta.to(dev), mean, rstd, act=act, dgamma=dg0,
                       dbeta=db0, accumulate=True)
    assert torch.allclose(dg0, d_gamma + 1.0, rtol=1e-5, atol=1e-5) and torch.allclose(db0, d_beta + 2.0, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ exact-f32 attention
@pytest.mark.parametrize("B,L,H,D,use_mask", [
    (1, 32, 8, 15, False),      # the notebook's RNA tower after the position-0 slice: one "sequence" of B = 32, 120 / 8
    (1, 32, 8, 160, False),     # its RBP tower: 1280 / 8
    (1, 64, 8, 64, True),       # tri-modal encoders (512 / 8), isolated cells masked as keys
    (3, 130, 4, 96, True),      # several sequences, > 2 key blocks, ragged tail
    (2, 300, 2, 24, True),
    (5, 1, 3, 8, False),        # a single key
    (2, 77, 2, 192, False),     # the largest head dim
    (4, 200, 8, 64, True),      # enough (batch, head, query-block) workgroups for four query rows per wave
    (2, 300, 8, 160, True),
    (16, 40, 8, 15, False),
])
def test_attention_f32_fwd_bwd_vs_f64(dev, B, L, H, D, use_mask):
    """clipk_attn_f32_fwd / _bwd against f64 torch on the same inputs: output, lse and dqkv at f32 rounding level."""
    from clip_dplm_amd import ops
    g = torch.Generator().manual_seed(L * 7 + D)
    qkv = torch.randn(B * L, 3 * H * D, generator=g).to(dev)
    dout = torch.randn(B * L, H * D, generator=g).to(dev)
    mask = None
    if use_mask:
        lens = torch.randint(max(1, L // 3), L + 1, (B,), generator=g)
        mask = (torch.arange(L)[None] < lens[:, None])
        mask[0, :] = True
        if L > 4:
            mask[0, 2] = False                                             # a hole, not only a tail
        mask = mask.to(torch.uint8).to(dev)
    scale = D ** -0.5
    out, lse = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=mask, q_scale=scale)
    dqkv = ops.attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=mask, q_scale=scale)
    q64 = qkv.double().requires_grad_(True)
    ref, lse_ref = _attn_ref(q64, B, L, H, D, mask, None, scale)
    ref.backward(dout.double())
    assert (out.double() - ref).abs().max().item() < 2e-5
    assert (lse.double() - lse_ref).abs().max().item() < 2e-5
    assert (dqkv.double() - q64.grad).abs().max().item() < 2e-5 * max(1.0, q64.grad.abs().max().item())
    out2, lse2 = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=mask, q_scale=scale)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)              # fixed summation order
    assert torch.equal(dqkv, ops.attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=mask, q_scale=scale))


def test_attention_f32_fully_masked_sequence_and_dropout(dev):
    """A sequence whose keys are all masked gives zero rows / lse = -inf / zero gradients (as clipk_attn_fwd); dropout on
    the probabilities draws the bf16 kernels' mask (same hash, same element index), checked against torch with the mask
    rebuilt in integer arithmetic (tests/ops_emulator.py)."""
    from clip_dplm_amd import ops
    import ops_emulator as E
    B, L, H, D = 3, 70, 4, 15
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * L, 3 * H * D, generator=g).to(dev)
    dout = torch.randn(B * L, H * D, generator=g).to(dev)
    mask = torch.ones(B, L, dtype=torch.uint8)
    mask[1] = 0
    mask[2, 50:] = 0
    mask = mask.to(dev)
    out, lse = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=mask, q_scale=0.3)
    assert out.view(B, L, -1)[1].abs().max().item() == 0.0 and torch.isinf(lse[1]).all()
    dqkv = ops.attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=mask, q_scale=0.3)
    assert torch.isfinite(dqkv).all() and dqkv.view(B, L, -1)[1].abs().max().item() == 0.0
    drop = (0.2, 1234567)
    out, lse = ops.attn_f32_fwd(qkv, B, L, H, D, key_mask=None, q_scale=0.3, dropout=drop)
    dqkv = ops.attn_f32_bwd(qkv, out, dout, lse, B, L, H, D, key_mask=None, q_scale=0.3, dropout=drop)
    qc = qkv.cpu().double().requires_grad_(True)
    ref, _ = E._attn_math(qc, B, L, H, D, None, None, 0.3, drop)
    ref.backward(dout.cpu().double())
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    assert (dqkv.cpu().double() - qc.grad).abs().max().item() < 2e-5 * max(1.0, qc.grad.abs().max().item())


def test_dropout_f32_kernel_matches_the_epilogue_mask(dev):
    from clip_dplm_amd import ops
    import ops_emulator as E
    x = torch.randn(37, 120, device=dev)
    a = torch.randn(37, 120, device=dev)
    y = ops.dropout_f32(x, (0.1, 99), addend=a)
    ref = E.dropout_f32(x.cpu(), (0.1, 99), addend=a.cpu())
    assert torch.equal(y.cpu(), ref)
    # the same elements a GEMM epilogue drops (index m * N + n): identity weight, f32 out
    eye = torch.eye(128, device=dev).to(torch.bfloat16)
    xb = torch.randn(64, 128, device=dev).to(torch.bfloat16)
    yg = ops.gemm_nt(xb, eye, out_dtype=torch.float32, dropout=(0.1, 99))
    yd = ops.dropout_f32(xb.float(), (0.1, 99))
    assert torch.equal(yg == 0, yd == 0)


@pytest.mark.parametrize("kernel", ["v2", "v3"])
def test_gelu_grad_code_grid_has_zero_and_one_as_code_points(dev, kopt, kernel):
    """the 8-bit GELU' levels are -0.13 + 0.005 k, so a dead unit (GELU' = 0) is code 26 and decodes
    to 0 (1.9e-9 in f32) and a saturated one (GELU' = 1) is code 226 and decodes to exactly 1 - no bias of one sign per class
    of unit (the round-3 grid decoded them to -0.0015 / 1.0015)."""
    ops = _ops()
    kopt("gemm_kernel", _GEMM_KERNEL[kernel])
    M, N, K = 2304, 256, 64
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    b = _rand((N, K), dev, 3, 0.05, dtype=torch.bfloat16)
    bias = torch.full((N,), -30.0, device=dev)
    bias[N // 2:] = 30.0
    g, q = ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True, aux_u8=True)
    assert (q[:, : N // 2] == 26).all() and (q[:, N // 2:] == 226).all()
    assert (g[:, : N // 2] == 0).all() and (g[:, N // 2:].float() == 30.0).all()
    x = _rand((M, K), dev, 4, dtype=torch.bfloat16)
    ref = ops.gemm_nt(x, b).float()                             # the same product through the plain epilogue
    d = ops.gemm_nt(x, b, dact_aux=q, dact="gelu").float()
    assert torch.equal(d[:, N // 2:], ref[:, N // 2:])                                      # x 1.0 exactly
    assert d[:, : N // 2].abs().max().item() <= 4e-9 * ref.abs().max().item()              # x 1.9e-9


# ------------------------------------------------------------------------------------------------ pair-interleaved RoPE
def _il_perm(n, hd, il_rows, dev):
    from clip_dplm_amd import ops
    return ops.il_source_rows(n, hd, il_rows).to(dev)


@pytest.mark.parametrize("kernel,M", [("v2", 1000), ("v3", 2048 + 24), ("v4", 2048 + 24)])
@pytest.mark.parametrize("H,D", [(20, 24), (4, 24), (3, 64), (5, 40), (6, 16)])
def test_gemm_nt_rope_interleaved_epilogue(dev, kopt, kernel, M, H, D):
    """clipk_gemm_nt with rope_interleaved (EPI_ROPE_IL): the fused qkv projection computed with pair-interleaved q / k rows
    (cast_transpose il) and rotated as neighbouring pairs in the epilogue, for head dims the tiled RoPE epilogue cannot take
    (24, 40) and those it can: against the f32 rotate-half rotation of the plain projection, column for column (undoing the
    permutation), within one bf16 rounding; v section = the plain epilogue's bits."""
    ops = _ops()
    kopt("gemm_kernel", _GEMM_KERNEL[kernel])
    d, L = H * D, 100
    K = 96 if d % 32 else d
    a = _rand((M, K), dev, 101, dtype=torch.bfloat16)
    w = _rand((3 * d, K), dev, 102, 0.2)
    bias = _rand((3 * d,), dev, 103)
    cos, sin = _rope_tables(L, D, dev)
    src = _il_perm(3 * d, D, 2 * d, dev)
    wb_il, _ = ops.cast_transpose(w, il=(D, 2 * d))
    out = ops.gemm_nt(a, wb_il, bias=bias[src].contiguous(), rope=(cos, sin, L, D, 2 * d), rope_interleaved=True)
    # reference: plain product in the ORIGINAL column order, rotate-half in f32
    pre = a.float() @ w.to(torch.bfloat16).float().t() + bias
    pos = torch.arange(M, device=dev) % L
    c = torch.cat([cos[pos], cos[pos]], -1)[:, None, :]
    s_ = torch.cat([sin[pos], sin[pos]], -1)[:, None, :]
    x = pre[:, : 2 * d].reshape(M, 2 * H, D)
    rot = torch.cat([-x[..., D // 2:], x[..., : D // 2]], -1)
    ref = torch.cat([(x * c + rot * s_).reshape(M, 2 * d), pre[:, 2 * d:]], 1)
    got = torch.empty_like(ref)
    got[:, src] = out.float()                                 # permuted column j holds original column src[j]
    err = (got - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 2e-3).all(), float(err.max())
    plain = ops.gemm_nt(a, wb_il, bias=bias[src].contiguous())
    assert torch.equal(out[:, 2 * d:], plain[:, 2 * d:])


@pytest.mark.parametrize("B,L,H,D", [(8, 256, 20, 24), (3, 190, 4, 24), (2, 100, 4, 24), (2, 300, 2, 24)])
def test_attention_bwd_interleaved_rope_equals_rotate_half(dev, B, L, H, D):
    """clipk_attn_bwd(prerotated = 2): q / k rotated and laid out in pair-interleaved head order give the gradients of the
    rotate-half layout, column for column (q . k does not depend on a common order of the head dim; RoPE^T knows the order)
    - whole-head fused kernel (128 < L <= 256) and the general dQ / dK-dV pair."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    qkv = (torch.randn(B * L, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dout = (torch.randn(B * L, H * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    cos, sin = _rope_tables(L, D, dev)
    std = ops.rope_qk_(qkv.clone(), B, L, H, D, (cos, sin))              # rotated, rotate-half column order
    src = _il_perm(3 * H * D, D, 2 * H * D, dev)
    il = std[:, src].contiguous()                                         # the same values, pair-interleaved
    o1, lse1 = ops.attn_fwd(std, B, L, H, D, rope=None, q_scale=D ** -0.5)
    o2, lse2 = ops.attn_fwd(il, B, L, H, D, rope=None, q_scale=D ** -0.5)
    assert (o1.float() - o2.float()).abs().max().item() < 2e-2 and (lse1 - lse2).abs().max().item() < 1e-4
    d1 = ops.attn_bwd(std, o1, dout, lse1, B, L, H, D, rope=(cos, sin), q_scale=D ** -0.5, prerotated=1)
    d2 = ops.attn_bwd(il, o2, dout, lse2, B, L, H, D, rope=(cos, sin), q_scale=D ** -0.5, prerotated=2)
    a, b = d1.float(), torch.empty_like(d1.float())
    b[:, src] = d2.float()
    assert (a - b).abs().max().item() <= 2e-2 * max(1.0, a.abs().max().item())
    assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item() > 0.9999


@pytest.mark.parametrize("M,N,K,hd,ilr", [(4096, 1440, 480, 24, 960), (20000, 1440, 480, 24, 960), (3000, 288, 96, 24, 192)])
def test_gemm_wgrad_interleaved_rows_land_in_the_original_order(dev, M, N, K, hd, ilr):
    """clipk_gemm_wgrad with (il_hd, il_rows): dY's leading columns are in pair-interleaved head order; dW rows and dbias
    entries are written (and accumulated) in the ORIGINAL order - bit-identical to the plain call on the un-permuted dY."""
    ops = _ops()
    dy = _rand((M, N), dev, 111, dtype=torch.bfloat16)
    x = _rand((M, K), dev, 112, dtype=torch.bfloat16)
    src = _il_perm(N, hd, ilr, dev)
    dw0, db0 = ops.gemm_wgrad(dy, x, want_bias=True)
    dw1, db1 = ops.gemm_wgrad(dy[:, src].contiguous(), x, want_bias=True, il=(hd, ilr))
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    acc_w, acc_b = torch.ones_like(dw0), torch.ones_like(db0)
    ops.gemm_wgrad(dy[:, src].contiguous(), x, dw=acc_w, dbias=acc_b, accumulate=True, il=(hd, ilr))
    assert torch.equal(acc_w, dw0 + 1.0) and torch.equal(acc_b, db0 + 1.0)
