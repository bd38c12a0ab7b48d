"""GPU parity at the BASELINE configurations' own sizes: the HIP path through the C ABI against
the CPU oracle / f64 torch on the same seeded inputs.

  * config 2 — the FULL-DEPTH model (esm2_t12_35M_UR50D: 12 x 480, hd 24 + 6 x 768 RNA encoder, L = 256), loss bar
    |gpu - oracle| <= 1e-3 (north star), parameter-gradient direction;
  * config 3 — fused similarity + CE at one rank's shape of the 8-rank global batch (Mx = 512 local rows against
    Ny = 4096 gathered keys, P = 512, label_offset = rank * 512) against f64 torch, plus the whole 8-virtual-rank loss;
  * config 4 — layers of the ESM-2-650M shape (d = 1280, 20 x 64, ffn 5120) at L = 1024 with padding;
  * config 5 — the ICNN transport system at its factory dims 512 / [512, 256], B = 64 and 4096.
Tolerances are stated per assert.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return F.cosine_similarity(a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten(), dim=0).item()


# ----------------------------------------------------------------------------------------------------- config 2
def _c2_batch(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    rna = torch.randn(B, L, 768, generator=g)
    return rna, ids


def test_c2_full_depth_loss_and_grads_vs_oracle(dev):
    """BASELINE config 2 at full depth (12 ESM layers + 6 RNA layers, L = 256, the bench model) on B = 64 pairs:
    |loss_gpu - loss_oracle| <= 1e-3 with bf16 activations between the GEMMs of all 18 layers, and the direction of
    parameter gradients from both ends of both towers (cosine; entries of InfoNCE gradients are cancellation sums,
    see test_gpu_models.assert_grad_close)."""
    import clip_dplm_amd as K
    from oracle import model_ref
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="esm2_t12_35M_UR50D").eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    B, L = 64, 256
    rna, ids = _c2_batch(B, L, 1234)
    m = m.to(dev)
    loss = m.loss(rna.to(dev), ids.to(dev))
    loss.backward()
    names = ["protein_model.encoder.layer.0.attention.self.query.weight",
             "protein_model.encoder.layer.11.output.dense.weight",
             "protein_model.encoder.layer.5.intermediate.dense.weight",
             "rna_model.layers.0.self_attn.in_proj_weight", "rna_model.layers.5.linear2.weight",
             "rna_projection.projection.0.weight", "protein_projection.projection.4.weight", "logit_scale"]
    sdr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ref, _, _ = model_ref.protein_rna_clip_loss(sdr, rna, ids, None, None, esm_layers=12, esm_heads=20, rna_layers=6,
                                                rna_heads=8)
    ref.backward()
    err = abs(loss.item() - ref.item())
    print(f"c2 full depth: loss_gpu={loss.item():.6f} loss_oracle={ref.item():.6f} |diff|={err:.2e}")
    assert err <= 1e-3, (loss.item(), ref.item())
    got = dict(m.named_parameters())
    for n in names:
        c = _cos(got[n].grad, sdr[n].grad)
        print(f"  grad cos {n}: {c:.5f}")
        assert c > 0.98, (n, c)


def test_c2_full_depth_padded_loss_vs_oracle(dev):
    """Same model with ragged lengths in [64, 256] on both sides (key-padding masks + masked-mean pooling)."""
    import clip_dplm_amd as K
    from oracle import model_ref
    torch.manual_seed(1)
    m = K.ProteinRNACLIP(esm="esm2_t12_35M_UR50D").eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    B, L = 32, 256
    rna, ids = _c2_batch(B, L, 99)
    g = torch.Generator().manual_seed(5)
    lp = torch.randint(64, L + 1, (B,), generator=g)
    lr = torch.randint(64, L + 1, (B,), generator=g)
    pmask = (torch.arange(L)[None] < lp[:, None]).long()
    rmask = (torch.arange(L)[None] < lr[:, None]).long()
    ids = torch.where(pmask.bool(), ids, torch.ones_like(ids))                 # <pad> = 1 beyond the length
    m = m.to(dev)
    with torch.no_grad():
        loss = m.loss(rna.to(dev), ids.to(dev), rna_mask=rmask.to(dev), protein_mask=pmask.to(dev))
        ref, _, _ = model_ref.protein_rna_clip_loss(sd, rna, ids, rmask, pmask, esm_layers=12, esm_heads=20,
                                                    rna_layers=6, rna_heads=8)
    err = abs(loss.item() - ref.item())
    print(f"c2 padded: loss_gpu={loss.item():.6f} loss_oracle={ref.item():.6f} |diff|={err:.2e}")
    assert err <= 1e-3, (loss.item(), ref.item())


def test_c2_full_size_properties_at_the_metric_batch(dev):
    """The bench workload itself (B = 1024 pairs, L = 256, full depth: 262144 token rows through every kernel at the
    shapes bench.py times), checked through size-independent properties the CPU oracle is too slow for:
      * a sample's embedding does not depend on its position in the batch or on its neighbours: encoding a permuted
        batch gives the permuted embeddings BIT FOR BIT (GEMM rows, LayerNorm rows, attention heads, pooling are
        per-sample; the tile a row lands in changes, its arithmetic does not);
      * a 64-pair slice of the batch, encoded alone, gives the same embeddings (the kernels the small problem selects
        differ - 128x128 instead of persistent 256x256 GEMM tiles - so this one is a tolerance, 2e-3 on unit vectors);
      * the fused loss at B = 1024 equals the f64 cross-entropy of the same embeddings to 1e-5, and the two directions
        of the symmetric loss swap under swapping the towers' embeddings."""
    import clip_dplm_amd as K
    from clip_dplm_amd.loss import clip_loss
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="esm2_t12_35M_UR50D").eval().to(dev)
    B, L = 1024, 256
    rna, ids = _c2_batch(B, L, 4321)
    rna, ids = rna.to(dev), ids.to(dev)
    with torch.no_grad():
        er, ep = m.embed(rna, ids)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(7)).to(dev)
        er_p, ep_p = m.embed(rna[perm].contiguous(), ids[perm].contiguous())
        assert torch.equal(er_p, er[perm]) and torch.equal(ep_p, ep[perm])
        er_s, ep_s = m.embed(rna[100:164].contiguous(), ids[100:164].contiguous())
        assert (er_s - er[100:164]).abs().max().item() < 2e-3 and (ep_s - ep[100:164]).abs().max().item() < 2e-3
        assert torch.allclose(er.norm(dim=-1), torch.ones(B, device=dev), atol=1e-5)
        sc = m.logit_scale.exp()
        loss = clip_loss(er, ep, sc, symmetric=True).item()
        S = (er.double() @ ep.double().t()) * sc.double()
        lab = torch.arange(B, device=dev)
        ref = 0.5 * (F.cross_entropy(S, lab) + F.cross_entropy(S.t(), lab)).item()
        assert abs(loss - ref) < 1e-5, (loss, ref)
        row = clip_loss(er, ep, sc, symmetric=False).item()
        col = clip_loss(ep, er, sc, symmetric=False).item()
        assert abs(row - F.cross_entropy(S, lab).item()) < 1e-5 and abs(col - F.cross_entropy(S.t(), lab).item()) < 1e-5
        assert abs(0.5 * (row + col) - loss) < 1e-6


# ----------------------------------------------------------------------------------------------------- config 3
@pytest.mark.parametrize("rank,Bl", [(0, 512), (3, 512), (7, 512), (5, 1024)])
def test_c3_simce_rank_block_vs_f64(dev, rank, Bl):
    """One rank's share of the global-batch InfoNCE: B_l local rows against the 8 * B_l gathered keys, P = 512: config
    3 (512 x 4096) and the shape of bench.py --gpus 8 at the metric batch (1024 x 8192)."""
    from clip_dplm_amd import ops
    W, P, scale = 8, 512, 14.2849
    Bg = Bl * W
    g = torch.Generator().manual_seed(31)
    a = F.normalize(torch.randn(Bg, P, generator=g), dim=-1).to(dev)
    b = F.normalize(torch.randn(Bg, P, generator=g), dim=-1).to(dev)
    sc = torch.tensor([scale], device=dev)
    off = rank * Bl
    sl = slice(off, off + Bl)
    S = (a.double() @ b.double().t()) * scale                                   # [Bg, Bg] f64 reference
    lse_r, pos_r = ops.simce_lse(a[sl].contiguous(), b, sc, label_offset=off)
    lse_c, pos_c = ops.simce_lse(b[sl].contiguous(), a, sc, label_offset=off)
    assert torch.allclose(lse_r.double(), torch.logsumexp(S[sl], 1), rtol=0, atol=2e-5)
    assert torch.allclose(lse_c.double(), torch.logsumexp(S[:, sl], 0), rtol=0, atol=2e-5)
    assert torch.allclose(pos_r.double(), S.diag()[sl], rtol=0, atol=2e-5)
    assert torch.allclose(pos_c.double(), S.diag()[sl], rtol=0, atol=2e-5)
    # gradient of the GLOBAL symmetric loss w.r.t. this rank's rows of a (needs the global LSE vectors)
    lse_r_g = torch.logsumexp(S, 1).float().contiguous()
    lse_c_g = torch.logsumexp(S, 0).float().contiguous()
    ad = a.double().requires_grad_(True)
    Sd = (ad @ b.double().t()) * scale
    lab = torch.arange(Bg, device=dev)
    Lr = 0.5 * (F.cross_entropy(Sd, lab) + F.cross_entropy(Sd.t(), lab))
    ga, = torch.autograd.grad(Lr, ad)
    da, _ = ops.simce_grad(a[sl].contiguous(), b, sc, lse_r_g[sl].contiguous(), lse_c_g, 0.5, 0.5, 1.0 / Bg,
                           label_offset=off)
    assert torch.allclose(da.double(), ga[sl], rtol=1e-4, atol=1e-7), (da.double() - ga[sl]).abs().max()


def test_c3_global_loss_from_8_virtual_ranks(dev):
    """The whole config-3 loss assembled the way loss.ClipLossFn does per rank (row + column LSE of each rank's 512
    pairs against the 4096 gathered keys), summed over 8 virtual ranks == f64 symmetric CE of the 4096 x 4096 logits."""
    from clip_dplm_amd import ops
    Bl, W, P, scale = 512, 8, 512, 14.2849
    Bg = Bl * W
    g = torch.Generator().manual_seed(32)
    a = F.normalize(torch.randn(Bg, P, generator=g), dim=-1).to(dev)
    b = F.normalize(torch.randn(Bg, P, generator=g), dim=-1).to(dev)
    sc = torch.tensor([scale], device=dev)
    total = torch.zeros((), dtype=torch.float64, device=dev)
    for r in range(W):
        sl = slice(r * Bl, (r + 1) * Bl)
        lr_, pr_ = ops.simce_lse(a[sl].contiguous(), b, sc, label_offset=r * Bl)
        lc_, pc_ = ops.simce_lse(b[sl].contiguous(), a, sc, label_offset=r * Bl)
        total += 0.5 * ((lr_ - pr_).double().sum() + (lc_ - pc_).double().sum())
    S = (a.double() @ b.double().t()) * scale
    lab = torch.arange(Bg, device=dev)
    ref = 0.5 * (F.cross_entropy(S, lab) + F.cross_entropy(S.t(), lab))
    assert abs(total.item() / Bg - ref.item()) < 1e-5, (total.item() / Bg, ref.item())


# ----------------------------------------------------------------------------------------------------- config 4
def test_c4_esm650m_shape_layers_vs_oracle(dev):
    """Two layers of the ESM-2-650M shape (d = 1280, 20 heads x 64, ffn 5120) at L = 1024 with ragged lengths against
    oracle.encoder_ref.esm_encoder (transformers.EsmModel arithmetic); forward element-wise at bf16 level, weight
    gradients by direction.  (33 layers would only repeat the same kernels on the same shapes.)"""
    import clip_dplm_amd as K
    from oracle import encoder_ref
    torch.manual_seed(3)
    enc = K.ESM2Encoder(num_layers=2, hidden_size=1280, num_heads=20, intermediate_size=5120).eval()
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    B, L = 4, 1024
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0] = 0
    lens = torch.tensor([1024, 700, 513, 130])
    am = (torch.arange(L)[None] < lens[:, None]).long()
    ids = torch.where(am.bool(), ids, torch.ones_like(ids))
    dy = torch.randn(B, L, 1280, generator=g) * am[..., None]
    enc = enc.to(dev)
    y = enc(ids.to(dev), attention_mask=am.to(dev))
    (y * dy.to(dev)).sum().backward()
    names = ["encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.value.weight",
             "encoder.layer.1.intermediate.dense.weight", "encoder.layer.1.output.dense.weight",
             "encoder.layer.0.attention.output.dense.weight", "encoder.layer.1.LayerNorm.weight"]
    sdr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ref = encoder_ref.esm_encoder(ids, am, sdr, 2, 20, 1e-5)
    (ref * dy).sum().backward()
    m = am[..., None].float()
    err = ((y.detach().cpu() - ref.detach()) * m).abs().max().item()
    print(f"c4 layers: max |y - ref| over valid tokens = {err:.4f} (|ref| max {ref.abs().max().item():.2f})")
    assert err < 0.05, err                                  # final-LayerNorm outputs are O(1); bf16 GEMM operands
    got = dict(enc.named_parameters())
    for n in names:
        c = _cos(got[n].grad, sdr[n].grad)
        assert c > 0.99, (n, c)


@pytest.mark.parametrize("M,N,K", [(4096, 3840, 1280), (4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120),
                                   (33000, 1280, 5120)])
def test_c4_gemm_shapes(dev, M, N, K):
    """The Linear / weight-gradient shapes of the 650M encoder (K = 1280 and 5120) through gemm_nt and gemm_wgrad."""
    from clip_dplm_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.03).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.float() @ b.float().t()
    c32 = ops.gemm_nt(a, b, out_dtype=torch.float32)
    assert torch.allclose(c32, ref, rtol=1e-4, atol=1e-4 * math.sqrt(K)), (c32 - ref).abs().max()
    gl, u = ops.gemm_nt(a, b, bias=bias, act="gelu", out_preact=True)
    assert torch.allclose(u.float(), ref + bias, rtol=1e-2, atol=2e-2)
    assert torch.allclose(gl.float(), F.gelu(ref + bias), rtol=1e-2, atol=2e-2)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    dw, db = ops.gemm_wgrad(dy, a, want_bias=True)
    wref = dy.float().t() @ a.float()
    tol = 2e-4 * math.sqrt(M)
    assert torch.allclose(dw, wref, rtol=1e-4, atol=tol), (dw - wref).abs().max()
    assert torch.allclose(db, dy.float().sum(0), rtol=1e-4, atol=tol)


def test_c4_full_size_frozen_encoder_properties(dev):
    """Config 4 at its own size (B = 256 sequences x L = 1024 = 262144 token rows, frozen encoder: the no-grad forward
    bench.py --config c4 times, with the FFN activation cut into M slabs and q / k rotated in the qkv epilogue), two
    layers of the 650M shape.  Size-independent properties: (1) a permuted batch gives the permuted outputs bit for
    bit; (2) sequences 64..67 encoded alone (different kernels: 128x128 GEMM tiles, no slabs) agree at bf16 level;
    (3) rotating in the epilogue agrees with the separate rotation pass at bf16 level."""
    import clip_dplm_amd as K
    from clip_dplm_amd import encoders
    torch.manual_seed(5)
    enc = K.ESM2Encoder(num_layers=2, hidden_size=1280, num_heads=20, intermediate_size=5120).eval().to(dev)
    for q in enc.parameters():
        q.requires_grad_(False)
    B, L = 256, 1024
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    ids = ids.to(dev)
    with torch.no_grad():
        y = enc(ids)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(dev)
        assert torch.equal(enc(ids[perm].contiguous()), y[perm])
        ys = enc(ids[64:68].contiguous())
        assert (ys - y[64:68]).abs().max().item() < 0.05                # final-LayerNorm outputs are O(1)
        old = encoders.ROPE_IN_QKV_EPILOGUE
        try:
            encoders.ROPE_IN_QKV_EPILOGUE = False
            y2 = enc(ids[:32].contiguous())
        finally:
            encoders.ROPE_IN_QKV_EPILOGUE = old
        assert (y2 - y[:32]).abs().max().item() < 0.05
        assert torch.isfinite(y).all()


# ----------------------------------------------------------------------------------------------------- config 5
@pytest.mark.parametrize("B", [64, 4096])
def test_c5_icnn_factory_dims_vs_oracle(dev, B):
    """create_transport_system(512, 512, 512) (hidden [512, 256], 4_transport_maps.py:256-267) in eval mode: the three
    transport maps and the reported cost against the CPU oracle's autograd-of-autograd; f32 path, tolerance 2e-4
    relative to the output scale (LayerNorm'ed outputs are O(1))."""
    from clip_dplm_amd import icnn
    from oracle import clip_ref, icnn_ref
    torch.manual_seed(11)
    model = icnn.create_transport_system(512, 512, 512)
    with torch.no_grad():                                   # move the z-path away from its near-zero init (scale 0.01)
        for mp in (model.cell_to_pert, model.cell_to_protein, model.pert_to_protein):
            mp.transport_net.layers[1].scale.fill_(0.05)
            mp.transport_net.final.weight.mul_(2.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    cell, pert, prot = (torch.randn(B, 512, generator=g) for _ in range(3))
    model = model.to(dev).eval()
    out = model(cell.to(dev), pert.to(dev), prot.to(dev))
    srcs = {"cell_to_pert": cell, "cell_to_protein": cell, "pert_to_protein": pert}
    tgts = {"cell_to_pert": pert, "cell_to_protein": prot, "pert_to_protein": prot}
    for name, src in srcs.items():
        ref = icnn_ref.single_cell_transport(src, sd, name, 2)
        err = (out[name].cpu() - ref).abs().max().item()
        assert err < 2e-4 * max(1.0, ref.abs().max().item()), (name, err)
        c = getattr(model, name).cost(src.to(dev), tgts[name].to(dev))
        cref, _, _ = icnn_ref.transport_cost(ref, clip_ref._ln(tgts[name], sd, f"{name}.output_norm", 1e-5))
        assert abs(c.cost.item() - cref.item()) < 1e-3, (name, c.cost.item(), cref.item())


# ------------------------------------------------------------------------------ the notebook model at its own dims
def test_notebook_model_at_its_own_dims_vs_oracle(dev):
    """the ONE variant the reference trained (current/rna_clip_codes.ipynb:2312-2360: 71,646,299 parameters,
    RNA features [32, L, 120] -> head dim 15, RBP features [32, 557-2542, 1280] -> head dim 160, 3 post-LN layers each,
    projection 512) at B = 32 with L_rbp = 600 and ragged NaN padding, against the CPU oracle computing EVERY position as
    the notebook does (pinned to the reference by notebook_model(_b32).npz at reduced widths).  The product slices to
    position 0 and runs in exact f32: loss at the north-star bar 1e-3 (measured ~1e-6), embeddings and gradients at f32
    level.  The optional bf16 kernels are printed beside it as a diagnostic (round 3: 1.3e-4 .. 1.05e-3 over six seeds)."""
    import clip_dplm_amd as K
    from oracle import model_ref
    torch.manual_seed(0)
    m = K.RNARBPCLIPModel(rna_dim=120, rbp_dim=1280, projection_dim=512).eval()
    assert sum(p.numel() for p in m.parameters()) == 71_646_299                  # the logged parameter count (:2312)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    B, Lr, Lp = 32, 48, 600
    rna = torch.randn(B, Lr, 120, generator=g)
    rbp = torch.randn(B, Lp, 1280, generator=g)
    lr = torch.randint(10, Lr + 1, (B,), generator=g)
    lp = torch.randint(Lp // 3, Lp + 1, (B,), generator=g)
    lr[0], lp[0] = Lr, Lp
    for i in range(B):
        rna[i, lr[i]:] = float("nan")
        rbp[i, lp[i]:] = float("nan")
    m = m.to(dev)
    ea, eb, loss = m(rna.to(dev), rbp.to(dev))
    loss.backward()
    names = ["rbp_encoder.layers.0.self_attn.in_proj_weight", "rbp_encoder.layers.2.linear2.weight",
             "rna_encoder.layers.0.self_attn.in_proj_weight", "rna_encoder.layers.2.linear1.weight",
             "rbp_projection.projection.0.weight", "rna_projection.skip.weight", "logit_scale"]
    sdr = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    oa, ob, ref = model_ref.rnarbp_clip_forward(sdr, rna, rbp)
    ref.backward()
    err = abs(loss.item() - ref.item())
    mb = K.RNARBPCLIPModel(rna_dim=120, rbp_dim=1280, projection_dim=512, precision="bf16").eval()
    mb.load_state_dict(sd)
    mb = mb.to(dev)
    with torch.no_grad():
        _, _, lb = mb(rna.to(dev), rbp.to(dev))
    print(f"notebook own dims: loss_gpu={loss.item():.6f} loss_oracle={ref.item():.6f} |diff|={err:.2e}; optional bf16 "
          f"kernels (diagnostic) |diff|={abs(lb.item() - ref.item()):.2e}")
    assert err < 1e-4, err                                                       # bar: 1e-3
    assert (ea.cpu() - oa.detach()).abs().max().item() < 1e-4 and (eb.cpu() - ob.detach()).abs().max().item() < 1e-4
    got = dict(m.named_parameters())
    for n in names:
        r = sdr[n].grad
        e = (got[n].grad.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-12)
        assert e < 1e-3, (n, e)
    assert abs(lb.item() - ref.item()) < 2e-2
