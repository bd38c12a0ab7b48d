"""GPU parity of the drop-in modules / encoders (HIP path through the C ABI) against
  (1) the committed golden vectors generated from the reference, and
  (2) the CPU oracle on the same seeded inputs and weights.
Tolerances: the path computes Linear layers in bf16 MFMA with f32 accumulation, so element-wise tolerances are
bf16-level (stated per assert); the LOSS tolerance is the north-star bar |loss_gpu - loss_ref| <= 1e-3.
"""
import math
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(G, name))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    return z, sd


def t(z, k, dev=None):
    x = torch.from_numpy(z[k])
    return x.to(dev) if dev is not None else x


def sub(h, n=2, eps=1e-12):
    return NS(hidden_size=h, num_hidden_layers=n, layer_norm_eps=eps)


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def bf16_weights(sd):
    """The state_dict with every GEMM weight (2-D, or in_proj) rounded to bf16 and back: the MODEL the bf16-MFMA path
    evaluates (north star: bf16 operands, f32 accumulate).  An f32 oracle on these weights isolates what is left -
    activation rounding, summation order, the kernels themselves - from the perturbation the weight rounding makes to
    the model, which is the same for every sample and therefore not averaged away by any batch size."""
    return {k: (v.to(torch.bfloat16).float() if (v.dim() == 2 and v.is_floating_point()) else v) for k, v in sd.items()}


def assert_grad_close(g, ref, name, cos_min=0.99, rel_max=0.25):
    """bf16-operand GEMMs: gradient DIRECTION must match tightly; single entries (sums with cancellation, ReLU
    decisions flipped by rounding) only loosely.  See DESIGN.md §precision."""
    g, ref = g.detach().float().cpu(), ref.detach().float().cpu()
    cos = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
    assert cos > cos_min, (name, "cos", cos)
    assert relerr(g, ref) < rel_max, (name, "rel", relerr(g, ref))


def test_clip_c1_golden_forward_loss_and_grads(dev):
    """BASELINE config 1 (old/clip.py, 2 layers, d=128, B=256): golden weights/inputs from the reference."""
    import clip_dplm_amd as K
    z, sd = load("clip_c1.npz")
    cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    m.load_state_dict(sd)                                   # reference state_dict keys load as they are
    m = m.to(dev).eval()
    out = m(t(z, "rna", dev), t(z, "protein", dev))
    assert set(out) == {"logits_per_rna_protein", "rna_embeds", "protein_embeds"}
    logits = out["logits_per_rna_protein"]
    assert logits.shape == (256, 256)
    assert (logits.cpu() - t(z, "logits")).abs().max().item() < 0.15        # bf16 GEMMs, logits scaled by 14.28
    assert (out["rna_embeds"].norm(dim=-1) - 1).abs().max().item() < 1e-5
    lab = torch.arange(256, device=dev)
    l1 = torch.nn.functional.cross_entropy(logits, lab)
    assert abs(l1.item() - float(z["loss_one_sided"])) < 1e-3
    # fused loss (no materialised logits) against the golden symmetric loss; then parameter gradients
    loss = m.loss(t(z, "rna", dev), t(z, "protein", dev), symmetric=True)
    assert abs(loss.item() - float(z["loss_symmetric"])) < 1e-3, (loss.item(), float(z["loss_symmetric"]))
    loss.backward()
    # InfoNCE parameter gradients are sums over the batch with heavy cancellation, so bf16 rounding of the GEMM
    # operands alone moves individual entries by up to ~16 % of the largest entry (reproduced with the f32 oracle
    # + bf16 straight-through rounding: DESIGN.md §precision).  Check direction tightly, entries loosely.
    for n, p in m.named_parameters():
        ref = t(z, "g:" + n, dev)
        cos = torch.nn.functional.cosine_similarity(p.grad.flatten(), ref.flatten(), dim=0).item()
        assert cos > 0.99, (n, cos)
        assert relerr(p.grad, ref) < 0.25, (n, relerr(p.grad, ref))
    # the one-sided fused loss is what old/ablation.py:16 trains with
    l1f = m.loss(t(z, "rna", dev), t(z, "protein", dev), symmetric=False)
    assert abs(l1f.item() - float(z["loss_one_sided"])) < 1e-3


def test_clip_diffmap_golden(dev):
    import clip_dplm_amd as K
    z, sd = load("clip_diffmap.npz")
    cfg = NS(rna_config=sub(64), protein_config=sub(96, 3, 1e-5), diffmap_config=sub(64, 1), projection_dim=32,
             logit_scale_init_value=2.6592)
    m = K.DiffMapProteinCLIPModule(cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    out = m(t(z, "diffmap", dev), t(z, "protein", dev))
    assert (out["logits_per_diffmap_protein"].cpu() - t(z, "logits")).abs().max().item() < 0.15


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_clip_opt_golden_cache_loss(dev, precision):
    """OptimizedCLIPModule (old/clip_opt.py): cache columns + clamp; module loss and fused loss against the reference's
    values at B = 32.  bf16 operands (the default arithmetic of the MLP towers): 32 rows do not average the rounding of a
    bf16 GEMM away - the 1e-3 bar is asserted at the reference's own batch size in
    test_clip_opt_b128_golden_loss_at_the_north_star_bar; here 5e-3 and the measured value is printed.  Exact-f32 Linears
    (`set_linear_precision(m, "f32")`, the arithmetic of the reference's fp32 caller): 1e-4."""
    import clip_dplm_amd as K
    z, sd = load("clip_opt.npz")
    cfg = NS(diffmap_config=sub(48), protein_config=sub(96), projection_dim=32, cache_size=256)
    m = K.OptimizedCLIPModule(cfg)
    m.load_state_dict(sd)
    K.set_linear_precision(m, precision)
    m = m.to(dev).eval()
    # put the cache in the state the reference had before this batch (first 32 rows = earlier batch)
    cache = t(z, "cache", dev)
    m.protein_embedding_cache[:32] = cache[:32]
    m.cache_ptr = 32
    out = m(t(z, "diffmap", dev), t(z, "protein", dev), gather_distributed=False)
    assert m.cache_ptr == int(z["cache_ptr"])
    e_logit = (out["logits_per_diffmap_protein"].cpu() - t(z, "logits")).abs().max().item()
    assert e_logit < (0.3 if precision == "bf16" else 2e-3), e_logit
    assert out["logits_per_diffmap_cache"].shape == tuple(z["logits_cache"].shape)
    loss = K.optimized_clip_loss(out)
    e_loss = abs(loss.item() - float(z["loss"]))
    print(f"clip_opt B=32 [{precision}]: |dloss| {e_loss:.2e}, max |dlogit| {e_logit:.2e}")
    assert e_loss < (5e-3 if precision == "bf16" else 1e-4), e_loss
    m.cache_ptr = 32
    lf = m.loss(t(z, "diffmap", dev), t(z, "protein", dev))
    assert abs(lf.item() - loss.item()) < 1e-4               # fused == materialised on the same embeddings


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_transformer_layer_golden(dev, act):
    """One nn.TransformerEncoderLayer (golden from torch): forward, input grad and parameter grads."""
    import clip_dplm_amd as K
    z, sd = load(f"tlayer_{act}.npz")
    enc = K.TransformerSeqEncoder(64, 1, 8, 128, act, 1e-12)
    own = {k.replace("l.", "layers.0.", 1): v for k, v in sd.items()}
    own["layernorm.weight"], own["layernorm.bias"] = torch.ones(64), torch.zeros(64)
    enc.load_state_dict(own)
    enc = enc.to(dev)
    x = t(z, "x", dev).requires_grad_(True)
    valid = t(z, "valid", dev)
    y = enc(x, src_key_padding_mask=~valid)
    # the golden y is the layer output; our encoder adds a final LayerNorm with weight 1 / bias 0 and eps 1e-12
    yref = torch.nn.functional.layer_norm(t(z, "y", dev), (64,), eps=1e-12)
    m = valid[..., None].float()
    assert ((y - yref) * m).abs().max().item() < 0.06
    y.backward(t(z, "dy", dev))
    # the extra final LN changes the upstream gradient of the layer, so gradients are compared with the oracle
    # (itself pinned to torch's layer by tests/test_oracle_golden.py) run through the same extra LN
    from oracle import encoder_ref
    xs = t(z, "x").requires_grad_(True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = encoder_ref.post_ln_layer(xs, sdo, "l", 8, t(z, "valid"), act, 1e-12)
    yo = torch.nn.functional.layer_norm(yo, (64,), eps=1e-12)
    yo.backward(t(z, "dy"))
    assert_grad_close(x.grad.cpu() * m.cpu(), xs.grad * m.cpu(), "dx", 0.995, 0.1)
    for n, p in enc.named_parameters():
        if n.startswith("layers.0."):
            assert_grad_close(p.grad, sdo["l." + n[len("layers.0."):]].grad, n)


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_gelu_stack_whose_width_is_not_a_multiple_of_32(dev, precision):
    """a post-LN gelu stack at the notebook's width 120 (K % 32 != 0: the generic-K GEMM, which has no
    8-bit GELU' operand) must train - the u8 request is gated by shape (ops.gelu_aux_u8_applies), such widths keep the
    bf16 pre-activation.  Forward and parameter gradients against the oracle; the f32 arithmetic on the same stack."""
    import clip_dplm_amd as K
    from oracle import encoder_ref
    torch.manual_seed(3)
    enc = K.TransformerSeqEncoder(120, 2, 8, 200, "gelu", 1e-5, precision=precision)
    sd = {"e." + k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    enc = enc.to(dev).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 40, 120, generator=g)
    valid = torch.arange(40)[None] < torch.tensor([40, 17, 33])[:, None]
    y = enc(x.to(dev), src_key_padding_mask=(~valid).to(dev))
    dy = torch.randn(3, 40, 120, generator=g) * valid[..., None]
    y.backward(dy.to(dev))
    yo = encoder_ref.post_ln_encoder(x, sd, "e", 2, 8, valid, "gelu", 1e-5, 1e-5)
    yo.backward(dy)
    tol = 0.06 if precision == "bf16" else 2e-4
    assert ((y.detach().cpu() - yo.detach()) * valid[..., None]).abs().max().item() < tol
    for n, p in enc.named_parameters():
        r = sd["e." + n].grad
        if precision == "bf16":
            assert_grad_close(p.grad, r, n)
        else:
            assert (p.grad.cpu() - r).abs().max().item() <= 2e-4 * max(r.abs().max().item(), 1e-3), n


def _slice_equivalence(build, run, rel=1e-5):
    """embeds / loss / EVERY parameter gradient of a position-0-pooled model with the dead positions removed before the
    encoders (slice_first_position=True, the default) against the full computation the notebook does."""
    outs = []
    for sl in (True, False):
        m = build(sl)
        o = run(m)
        o[-1].backward()
        outs.append(([x.detach().cpu() for x in o], {n: p.grad.detach().cpu() for n, p in m.named_parameters()}))
    (a, ga), (b, gb) = outs
    exact = all(torch.equal(x, y) for x, y in zip(a, b)) and all(torch.equal(ga[n], gb[n]) for n in ga)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= rel * max(1.0, y.abs().max().item())
    for n in ga:
        assert (ga[n] - gb[n]).abs().max().item() <= rel * max(1e-6, gb[n].abs().max().item()), n
    return exact


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_notebook_model_golden(dev, precision):
    """RNARBPCLIPModel (rna_clip_codes.ipynb:1925-1954) incl. the batch-axis attention quirk and NaN padding, against the
    values the REFERENCE class produced (tools/make_golden.py gen_notebook).  Default arithmetic (exact f32, position 0
    sliced before the encoders): the north-star bar |loss - reference| <= 1e-3 holds with three orders of margin.
    precision="bf16" (optional bf16-MFMA kernels) is checked for agreement at bf16 level only: this model pools ONE
    position, so bf16-rounded weights move its loss by ~2e-3 - that is why f32 is its default."""
    import clip_dplm_amd as K
    z, sd = load("notebook_model.npz")
    m = K.RNARBPCLIPModel(rna_dim=24, rbp_dim=64, projection_dim=32, precision=precision)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    ea, eb, loss = m(t(z, "rna", dev), t(z, "rbp", dev))
    e_emb = max((ea.cpu() - t(z, "rna_embed")).abs().max().item(), (eb.cpu() - t(z, "rbp_embed")).abs().max().item())
    e_loss = abs(loss.item() - float(z["loss"]))
    print(f"notebook B=8 [{precision}]: |dloss| {e_loss:.2e}, max |dembed| {e_emb:.2e}")
    if precision == "f32":
        assert e_loss < 1e-4 and e_emb < 1e-4, (e_loss, e_emb)       # bar: 1e-3
    else:
        assert e_loss < 2e-2 and e_emb < 0.03, (e_loss, e_emb)       # diagnostic: the optional arithmetic runs and agrees
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_notebook_model_position0_slice_is_exact(dev, precision):
    """VERDICT r03 #1a: slicing to position 0 before the encoders (rna_clip_codes.ipynb:1944-1949 reads only `enc[:, 0]`
    of a stack whose attention mixes the batch axis per position) changes nothing: embeds, loss and every parameter
    gradient equal the full computation (bit for bit where the summation grouping does not depend on the row count)."""
    import clip_dplm_amd as K
    z, sd = load("notebook_model_b32.npz")

    def build(sl):
        m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64, precision=precision, slice_first_position=sl)
        m.load_state_dict(sd)
        return m.to(dev).eval()
    exact = _slice_equivalence(build, lambda m: m(t(z, "rna", dev), t(z, "rbp", dev)),
                               rel=1e-5 if precision == "f32" else 2e-2)
    print(f"notebook position-0 slice [{precision}]: bitwise equal = {exact}")


def _load_esm(K, sd, dev, nl=2, d=96, h=4, f=384):
    enc = K.ESM2Encoder(nl, d, h, f)
    missing = enc.load_state_dict({k: v for k, v in sd.items() if v.is_floating_point() and "inv_freq" not in k
                                   and "position" not in k and "contact" not in k}, strict=False)
    assert not missing.missing_keys, missing
    return enc.to(dev)


def test_esm_tiny_golden(dev):
    """ESM-2 arithmetic (third-party EsmModel, golden from transformers): hd = 24 like ESM-2-35M, padding +
    <mask> token-dropout rescale; forward and parameter gradients."""
    import clip_dplm_amd as K
    z, sd = load("esm_tiny.npz")
    enc = _load_esm(K, sd, dev)
    ids, am = t(z, "ids", dev), t(z, "attention_mask", dev)
    y = enc(ids, am)
    m = am[..., None].float()
    ref = t(z, "last_hidden_state", dev)
    assert ((y - ref) * m).abs().max().item() < 0.05, ((y - ref) * m).abs().max().item()
    zg = np.load(os.path.join(G, "esm_tiny_grads.npz"))
    (y * torch.from_numpy(zg["dy"]).to(dev)).sum().backward()
    checked = 0
    for n, p in enc.named_parameters():
        k = "g:" + n
        if k in zg.files and zg[k].size:
            ref = torch.from_numpy(zg[k]).to(dev)
            if ref.abs().max() < 1e-6:
                continue
            assert_grad_close(p.grad, ref, n)
            checked += 1
    assert checked > 30


def test_protein_rna_clip_vs_oracle(dev):
    """Reduced BASELINE-config-2 model (same code path: ESM hd=24 + post-LN gelu encoder + heads + fused loss)
    against the CPU oracle on identical seeded weights/inputs; loss bar 1e-3."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    from oracle import clip_ref, encoder_ref
    ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=128, projection_dim=64).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    B, L = 48, 40
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(4, 24, (B, L), generator=g); ids[:, 0] = 0; ids[:, -1] = 2
    rna = torch.randn(B, L, 64, generator=g)
    lens = torch.randint(10, L + 1, (B,), generator=g)
    pmask = (torch.arange(L)[None] < lens[:, None]).long()
    rmask = (torch.arange(L)[None] < lens.flip(0)[:, None]).long()
    m = m.to(dev)
    loss = m.loss(rna.to(dev), ids.to(dev), rna_mask=rmask.to(dev), protein_mask=pmask.to(dev))
    # oracle
    esd = {k[len("protein_model."):]: v for k, v in sd.items() if k.startswith("protein_model.")}
    hp = encoder_ref.esm_encoder(ids, pmask, esd, 2, 4, 1e-5)
    rsd = {k[len("rna_model."):]: v for k, v in sd.items() if k.startswith("rna_model.")}
    hr = encoder_ref.post_ln_encoder(rna, {"e." + k: v for k, v in rsd.items()}, "e", 2, 8, rmask.bool(), "gelu", 1e-12, 1e-12)
    er = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hr, rmask.bool(), "mean"), sd, "rna_projection"))
    ep = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hp, pmask.bool(), "mean"), sd, "protein_projection"))
    ref = clip_ref.clip_loss_symmetric((er @ ep.t()) * sd["logit_scale"].exp())
    print(f"protein_rna_clip (96-wide toy): loss_gpu={loss.item():.6f} loss_oracle={ref.item():.6f} "
          f"|diff|={abs(loss.item() - ref.item()):.2e}")
    assert abs(loss.item() - ref.item()) < 1e-3, (loss.item(), ref.item())
    loss.backward()
    gn = sum((p.grad.float() ** 2).sum() for p in m.parameters() if p.grad is not None).sqrt().item()
    assert math.isfinite(gn) and gn > 0


@pytest.mark.parametrize("dual_stream", [False, True])
def test_training_step_is_bitwise_reproducible(dev, dual_stream):
    """No float atomics anywhere on the path (split-M / split-key / slice partials are combined in a fixed order,
    the embedding gradient is a one-hot product on the f32 matrix pipe): two identical forward + backward passes give
    bit-identical losses and parameter gradients, also with the two towers on separate HIP streams.  Large enough
    (M = 16384 token rows) to go through the 256x256 GEMM / weight-gradient kernels."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_repro"] = (2, 480, 20, 1920)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="test_repro", rna_dim=768, rna_layers=1, rna_heads=8, rna_ffn=2048, projection_dim=128).eval()
    m = m.to(dev)
    m.dual_stream = dual_stream
    B, L = 64, 256
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(4, 24, (B, L), generator=g).to(dev)
    rna = torch.randn(B, L, 768, generator=g).to(dev)
    opt = K.FusedAdamW(m, lr=1e-4)
    out = []
    for _ in range(2):
        opt.zero_grad()
        loss = m.loss(rna, ids)
        loss.backward()
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), opt.flat.grad.detach().clone()))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert out[0][1].abs().sum().item() > 0


def test_large_tile_kernels_agree_with_small_tile_kernels_in_the_model(dev, kopt):
    """The persistent 256x256 Linear kernel and the 256x256 weight-gradient kernel (picked automatically at
    benchmark scale) against the 128x128 kernels, which the golden-vector tests above pin to the reference: same
    model, same inputs, loss within 2e-4 and every parameter gradient with cosine > 0.9999 (both accumulate exact
    bf16 products in f32, only the summation order differs)."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_repro"] = (2, 480, 20, 1920)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="test_repro", rna_dim=768, rna_layers=1, rna_heads=8, rna_ffn=2048, projection_dim=128).eval()
    m = m.to(dev)
    B, L = 64, 256
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(4, 24, (B, L), generator=g).to(dev)
    rna = torch.randn(B, L, 768, generator=g).to(dev)
    res = {}
    for tag, v in (("small", 2), ("large", 3), ("two_per_cu", 4)):       # 4: gemm_nt_v4 (+ the pipelined wgrad schedule)
        kopt("gemm_kernel", v)
        kopt("wgrad_kernel", v)
        m.zero_grad(set_to_none=True)
        loss = m.loss(rna, ids)
        loss.backward()
        res[tag] = (loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert abs(res["small"][0] - res["large"][0]) < 2e-4, (res["small"][0], res["large"][0])
    assert abs(res["small"][0] - res["two_per_cu"][0]) < 2e-4, (res["small"][0], res["two_per_cu"][0])
    # and the large-tile path against the CPU oracle directly (loss bar 1e-3, as everywhere)
    from oracle import clip_ref, encoder_ref
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ones = torch.ones(B, L, dtype=torch.bool)
    esd = {k[len("protein_model."):]: v for k, v in sd.items() if k.startswith("protein_model.")}
    hp = encoder_ref.esm_encoder(ids.cpu(), ones.long(), esd, 2, 20, 1e-5)
    rsd = {k[len("rna_model."):]: v for k, v in sd.items() if k.startswith("rna_model.")}
    hr = encoder_ref.post_ln_encoder(rna.cpu(), {"e." + k: v for k, v in rsd.items()}, "e", 1, 8, ones, "gelu", 1e-12, 1e-12)
    er = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hr, ones, "mean"), sd, "rna_projection"))
    ep = clip_ref.l2_normalize(clip_ref.projection_head(encoder_ref.pool(hp, ones, "mean"), sd, "protein_projection"))
    ref = clip_ref.clip_loss_symmetric((er @ ep.t()) * sd["logit_scale"].exp()).item()
    assert abs(res["large"][0] - ref) < 1e-3, (res["large"][0], ref)
    for other in ("large", "two_per_cu"):
        for n, ga in res["small"][1].items():
            gb = res[other][1][n]
            if ga.abs().max() < 1e-12:
                continue
            cos = torch.nn.functional.cosine_similarity(ga.flatten().double(), gb.flatten().double(), dim=0).item()
            assert cos > 0.9999, (other, n, cos)


def test_fused_adamw_training_reduces_loss(dev):
    """A few fused optimiser steps on config 1: loss goes down, flat grads are used, weights stay in sync."""
    import clip_dplm_amd as K
    z, sd = load("clip_c1.npz")
    cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    a, b = t(z, "rna", dev), t(z, "protein", dev)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = m.loss(a, b, symmetric=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.05, losses
    assert m.logit_scale.data_ptr() >= opt.flat.data.data_ptr()


def test_training_trajectory_matches_the_oracle(dev):
    """the metric is a TRAINING step, so parity must hold beyond step 0.  Three optimiser updates of the
    reference's loop (rna_clip_codes.ipynb:2061-2089 / old/ablation.py:9-18: loss -> backward -> clip_grad_norm_(1.0) ->
    AdamW(lr 1e-4, wd 0.01)) on the smoke-size model (two layers of each tower at the metric model's widths): the fused
    path (hand-written backward, flat-buffer clip + AdamW kernels, refreshed bf16 weight copies) next to the CPU oracle
    under torch.optim.AdamW.  Asserted:
      (a) at EVERY trained step the GPU loss is within the north-star 1e-3 of the oracle evaluated at the GPU path's own
          weights - the parity claim, step after step;
      (b) the two trajectories: within 1e-3 at step 0 and within 0.2 % of the loss decrease afterwards.  They cannot stay
          within 1e-3 absolutely: AdamW's first update is lr * sign(g) for every entry however small, ~0.35 % of the
          entries (uniformly over all matrices, carrying ~3e-5 of sum|g|) have a bf16 gradient of the other sign and move
          by 2 lr, and at 32 memorised pairs the loss falls by ~1.0 per update - measured: 8e-4 / 1.5e-3 / 1.9e-3 of
          weight-divergence error after 1 / 2 / 3 updates (the numbers are printed; bench.py reports the same split);
      (c) the weights themselves after 3 updates: no entry of the big matrices further than 3 x 2 lr from the oracle's (an
          entry whose sign differed at every step), mean deviation below lr / 10."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    from oracle import model_ref
    ESM2_SHAPES["smoke"] = (2, 480, 20, 1920)
    torch.manual_seed(0)
    kw = dict(esm="smoke", rna_dim=768, rna_layers=2, rna_heads=8, rna_ffn=2048, projection_dim=512)
    m = K.ProteinRNACLIP(**kw).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    B, L = 32, 64
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    rna = torch.randn(B, L, 768, generator=g)
    m = m.to(dev)
    opt = K.FusedAdamW(m, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
    params = [v for v in sd.values() if v.requires_grad]
    topt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    okw = dict(esm_layers=2, esm_heads=20, rna_layers=2, rna_heads=8)
    got, want, at_own = [], [], []
    for _ in range(4):
        opt.zero_grad()
        loss = m.loss(rna.to(dev), ids.to(dev))
        got.append(loss.item())
        with torch.no_grad():
            own = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            at_own.append(model_ref.protein_rna_clip_loss(own, rna, ids, None, None, **okw)[0].item())
        loss.backward()
        opt.step()
        topt.zero_grad()
        ref, _, _ = model_ref.protein_rna_clip_loss(sd, rna, ids, None, None, **okw)
        want.append(ref.item())
        ref.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        topt.step()
    fwd = [abs(a - b) for a, b in zip(got, at_own)]
    traj = [abs(a - b) for a, b in zip(got, want)]
    print("trajectory: " + "; ".join(f"step {i}: gpu {a:.5f} oracle {b:.5f} |d| {e:.1e} (forward {f:.1e})"
                                    for i, (a, b, e, f) in enumerate(zip(got, want, traj, fwd))))
    assert want[-1] < want[0] - 0.01, want                     # the steps really train
    assert max(fwd) < 1e-3, fwd                                # (a)
    assert traj[0] < 1e-3, traj                                # (b)
    for i in range(1, 4):
        assert traj[i] < 1e-3 + 2e-3 * abs(want[0] - want[i]), (i, traj)
    own = dict(m.named_parameters())                           # (c)
    for n in ("protein_model.encoder.layer.1.intermediate.dense.weight", "rna_model.layers.0.self_attn.in_proj_weight",
              "rna_projection.projection.0.weight"):
        d = (own[n].detach().cpu() - sd[n].detach()).abs()
        assert d.max().item() <= 6.5e-4 and d.mean().item() <= 1e-5, (n, d.max().item(), d.mean().item())


@pytest.mark.parametrize("precision,two_streams", [("f32", False), ("bf16", False), ("f32", True), ("bf16", True)])
def test_graphed_train_step_equals_eager_steps(dev, precision, two_streams):
    """training.GraphedTrainStep: the whole training step (zero_grad -> forward -> symmetric InfoNCE -> backward -> clip ->
    AdamW, rna_clip_codes.ipynb:2061-2089) replayed from ONE hipGraph, inputs and AdamW's per-step scalars in device memory.
    Five steps on five different batches: losses and every weight bit-identical to the same steps issued eagerly, the
    model untouched by the capture's warm-up, and the optimiser's step count advanced by the replays.
    two_streams: the captured model runs its towers as two parallel branches of the graph (`dual_stream`), the eager one on
    one stream - same bits."""
    import clip_dplm_amd as K
    from clip_dplm_amd.training import GraphedTrainStep

    def build(dual=False):
        torch.manual_seed(1)
        m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64, dropout=0.0, precision=precision)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.dual_stream = dual
        return m.to(dev).train()
    g = torch.Generator().manual_seed(3)
    batches = []
    for _ in range(5):
        rna, rbp = torch.randn(32, 6, 40, generator=g), torch.randn(32, 9, 128, generator=g)
        rna[5, 4:] = float("nan")
        rbp[7, 3:] = float("nan")
        batches.append((rna.to(dev), rbp.to(dev)))
    me = build()
    oe = K.FusedAdamW(me, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    eager = []
    for i, (rna, rbp) in enumerate(batches):
        oe.zero_grad()
        loss = me(rna, rbp)[2]
        loss.backward()
        oe.step(lr=1e-3 * (1 + i))                          # a schedule: the learning rate changes every step
        eager.append(loss.item())
    mg = build(two_streams)
    og = K.FusedAdamW(mg, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    w0 = og.flat.data.clone()
    step = GraphedTrainStep(mg, og, lambda a, b: mg(a, b)[2], batches[0])
    assert torch.equal(og.flat.data, w0) and og.step_count == 0 and float(og.m.abs().max()) == 0.0
    graphed = [step(rna, rbp, lr=1e-3 * (1 + i)).item() for i, (rna, rbp) in enumerate(batches)]
    assert og.step_count == 5
    assert graphed == eager, (graphed, eager)
    for (n, p), (_, q) in zip(me.named_parameters(), mg.named_parameters()):
        assert torch.equal(p, q), n
    assert eager[-1] < eager[0]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graphed_train_step_falls_back_to_eager_for_a_short_last_batch(dev, precision):
    """An epoch's last batch is shorter than the captured shape (the notebook's DataLoader has no drop_last,
    rna_clip_codes.ipynb:2061-2089): GraphedTrainStep runs that step eagerly and carries on replaying - losses and weights of
    [full, full, short, full] identical to the same four steps issued eagerly."""
    import clip_dplm_amd as K
    from clip_dplm_amd.training import GraphedTrainStep

    def build():
        torch.manual_seed(1)
        m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64, dropout=0.0, precision=precision)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m.to(dev).train()
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(b, 6, 40, generator=g).to(dev), torch.randn(b, 9, 128, generator=g).to(dev)) for b in (32, 32, 20, 32)]
    me = build()
    oe = K.FusedAdamW(me, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    eager = []
    for rna, rbp in batches:
        oe.zero_grad()
        loss = me(rna, rbp)[2]
        loss.backward()
        oe.step()
        eager.append(loss.item())
    mg = build()
    og = K.FusedAdamW(mg, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    step = GraphedTrainStep(mg, og, lambda a, b: mg(a, b)[2], batches[0])
    got = [step(rna, rbp).item() for rna, rbp in batches]
    assert got == eager, (got, eager)
    assert og.step_count == 4 and torch.equal(og.flat.data, oe.flat.data)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graphed_train_step_with_active_dropout_draws_new_masks_per_replay(dev, precision, monkeypatch):
    """GraphedTrainStep with nn.TransformerEncoderLayer's dropout ACTIVE (the configuration the notebook trains,
    rna_clip_codes.ipynb:1915, :2061-2089).  Replay k must be the eager step whose dropout sites use seed + k * 0x9E3779B9
    (the epoch word the captured step increments): with the encoders' host seeds pinned, three replays equal three eager
    steps made with the word set to 0, 1, 2 - losses and every weight, bit for bit - and differ from steps that repeat
    epoch 0's masks."""
    import clip_dplm_amd as K
    from clip_dplm_amd import ops
    from clip_dplm_amd.encoders import TransformerSeqEncoder
    from clip_dplm_amd.training import GraphedTrainStep
    pdrop = 0.2
    monkeypatch.setattr(TransformerSeqEncoder, "_draw_dropout",
                        lambda self: (pdrop, [[101 + 4 * i, 102 + 4 * i, 103 + 4 * i, 104 + 4 * i]
                                              for i in range(self.num_layers)]) if self.training else None)

    def build():
        torch.manual_seed(1)
        m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64, dropout=pdrop, precision=precision)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0                                 # torch's own generator is not what this test pins
        return m.to(dev).train()
    g = torch.Generator().manual_seed(3)
    rna, rbp = torch.randn(32, 6, 40, generator=g).to(dev), torch.randn(32, 9, 128, generator=g).to(dev)

    def eager(epochs):
        m = build()
        o = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        word = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.set_dropout_epoch(word)
        try:
            losses = []
            for e in epochs:
                word.fill_(e)
                o.zero_grad()
                loss = m(rna, rbp)[2]
                loss.backward()
                o.step()
                losses.append(loss.item())
        finally:
            ops.set_dropout_epoch(None)
        return losses, o.flat.data.clone()
    want, w_want = eager([0, 1, 2])
    same_masks, _ = eager([0, 0, 0])
    mg = build()
    og = K.FusedAdamW(mg, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    step = GraphedTrainStep(mg, og, lambda a, b: mg(a, b)[2], (rna, rbp))
    got = [step(rna, rbp).item() for _ in range(3)]
    assert got == want, (got, want)
    assert torch.equal(og.flat.data, w_want)
    assert got[1:] != same_masks[1:]
    no_drop = build().eval()
    with torch.no_grad():
        assert abs(no_drop(rna, rbp)[2].item() - want[0]) > 1e-4       # the dropout was really active


@pytest.mark.parametrize("model_name", ["notebook", "trimodal", "pair_clip"])
def test_towers_on_side_streams_give_the_same_step(dev, model_name):
    """`RNARBPCLIPModel.dual_stream` / `ContrastiveModel.multi_stream`: the towers (independent up to the loss,
    rna_clip_codes.ipynb:1944-1953, tf_clip_codes (1).ipynb:13140-13176) enqueued on HIP streams of their own, issued
    eagerly.  Loss and EVERY parameter gradient - the kernels add them into FusedAdamW's .grad views on the side streams -
    must be bit-identical to the one-stream step when read right after backward() (the join is an autograd-engine callback),
    and so must the weights after three optimiser steps."""
    import clip_dplm_amd as K
    g = torch.Generator().manual_seed(11)
    if model_name == "notebook":
        def build(par):
            torch.manual_seed(2)
            m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=256, projection_dim=64, dropout=0.0)
            m.dual_stream = par
            return m
        rna, rbp = torch.randn(32, 5, 40, generator=g), torch.randn(32, 7, 256, generator=g)
        rna[3, 2:] = float("nan")
        inputs = (rna.to(dev), rbp.to(dev))
        loss_of = lambda m: m(*inputs)[2]
    elif model_name == "pair_clip":                       # old/clip.py's module (BASELINE config 1), bf16 Linears
        from types import SimpleNamespace as NS
        sub = lambda hh: NS(hidden_size=hh, num_hidden_layers=2, layer_norm_eps=1e-12)
        cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
                 logit_scale_init_value=2.6592)

        def build(par):
            torch.manual_seed(2)
            m = K.RNAProteinCLIPModule(cfg)
            m.dual_stream = par
            return m
        inputs = (torch.randn(256, 128, generator=g).to(dev), torch.randn(256, 128, generator=g).to(dev))
        loss_of = lambda m: m.loss(*inputs, symmetric=False)
    else:
        def build(par):
            torch.manual_seed(2)
            m = K.ContrastiveModel(21, 64, projection_dim=64, esm_dim=40)
            m.multi_stream = par
            return m
        z, _ = load("trimodal_model.npz")
        inputs = tuple(t(z, k, dev) for k in ("cell_state", "connectivity", "gene_esm", "gene_values", "protein_emb"))
        loss_of = lambda m: m(*inputs)["loss"]
    runs = []
    for par in (False, True):
        m = build(par)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m = m.to(dev).train()
        opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        opt.zero_grad()
        loss = loss_of(m)
        loss.backward()
        grads = opt.flat.grad.clone() if getattr(opt.flat, "grad", None) is not None else \
            torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        losses = [loss.item()]
        opt.step()
        for _ in range(2):
            opt.zero_grad()
            loss = loss_of(m)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        runs.append((losses, grads, opt.flat.data.clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]) and float(runs[0][1].abs().max()) > 0
    assert torch.equal(runs[0][2], runs[1][2])


def test_icnn_transport_golden(dev):
    """BASELINE config 5 (eval): T(x) = dPsi/dx from the hand-derived gradient on exact-f32 MFMA kernels vs the
    reference's autograd-of-autograd output (golden); f32 path, tolerance 2e-4."""
    from clip_dplm_amd import icnn
    z, sd = load("icnn_transport.npz")
    model = icnn.create_transport_system(64, 64, 64, hidden_dims=[64, 64, 32])
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    out = model(t(z, "cell", dev), t(z, "pert", dev), t(z, "protein", dev))
    assert set(out) == {"cell_to_pert", "cell_to_protein", "pert_to_protein"}
    for k, v in out.items():
        ref = t(z, "out_" + k, dev)
        assert (v - ref).abs().max().item() < 2e-4, (k, (v - ref).abs().max().item())
    psi, _ = model.cell_to_pert.transport_net(torch.nn.functional.layer_norm(
        t(z, "cell", dev), (64,), model.cell_to_pert.input_norm.weight, model.cell_to_pert.input_norm.bias))
    assert torch.allclose(psi, t(z, "psi_cell_to_pert", dev), atol=1e-4)
    c = model.cell_to_pert.cost(t(z, "cell", dev), t(z, "pert", dev))
    from oracle import icnn_ref, clip_ref
    tgt = clip_ref._ln(t(z, "pert"), sd, "cell_to_pert.output_norm", 1e-5)
    cref, _, _ = icnn_ref.transport_cost(t(z, "out_cell_to_pert"), tgt)
    assert abs(c.cost.item() - cref.item()) < 1e-3
    model.train()
    with pytest.raises(TypeError):                         # the reference's 3-modality training call is broken (A-12)
        model(t(z, "cell", dev), t(z, "pert", dev), t(z, "protein", dev))


def test_icnn_transport_maps_as_graph_branches_give_the_same_output(dev):
    """`TripleTransportMaps.multi_stream` (eval): the three maps (4_transport_maps.py:147-224, each reads only its source)
    on three HIP streams, issued eagerly and replayed from one hipGraph (`GraphedTransport`: three branches) - outputs
    bit-identical to the one-stream call, also for a second batch copied into the graph's static inputs."""
    from clip_dplm_amd import icnn
    torch.manual_seed(4)
    model = icnn.create_transport_system(96, 96, 96, hidden_dims=[96, 96, 48]).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    mk = lambda: tuple(torch.randn(200, 96, generator=g).to(dev) for _ in range(3))
    x1, x2 = mk(), mk()
    ref1 = {k: v.clone() for k, v in model(*x1).items()}
    ref2 = {k: v.clone() for k, v in model(*x2).items()}
    model.multi_stream = True
    out = model(*x1)
    torch.cuda.synchronize()
    assert set(out) == set(ref1) and all(torch.equal(out[k], ref1[k]) for k in ref1)
    graphed = icnn.GraphedTransport(model, *x1)
    for x, ref in ((x1, ref1), (x2, ref2), (x1, ref1)):
        o = graphed(*x)
        torch.cuda.synchronize()
        assert all(torch.equal(o[k], ref[k]) for k in ref)


@pytest.mark.parametrize("case", ["A", "B"])
def test_icnn_training_through_transport_map_golden(dev, case):
    """BASELINE config 5, training branch: cost and d cost / d parameters THROUGH T(x) = dPsi/dx (second derivatives
    of Psi) vs the reference's double backward (golden, tools/make_golden.py gen_icnn_train).  Case A: no train-time
    rescale, every parameter gets a gradient; case B: the rescale fires in the last layer, whose z contribution the
    reference then treats as a constant.  In both some rows of T are norm-clipped.  Every matrix product (forward,
    first and second order) runs on the exact-f32 MFMA kernel; tolerance 2e-4 relative to the largest entry."""
    from clip_dplm_amd import icnn
    zf = np.load(os.path.join(G, "icnn_train.npz"))
    pre = case + ":"
    sd = {k[len(pre) + 2:]: torch.from_numpy(zf[k]) for k in zf.files if k.startswith(pre + "w:")}
    m = icnn.SingleCellTransport(64, 64, icnn.ICNNConfig(input_dim=64, hidden_dims=[64, 64, 32]))
    m.load_state_dict(sd)
    m = m.to(dev).train()
    src, tgt = torch.from_numpy(zf[pre + "source"]).to(dev), torch.from_numpy(zf[pre + "target"]).to(dev)
    out = m(src, tgt)
    assert isinstance(out, icnn.TransportOutput)
    assert (out.transported.detach().cpu() - torch.from_numpy(zf[pre + "transported"])).abs().max().item() < 2e-4
    assert abs(out.cost.item() - float(zf[pre + "cost"])) < 2e-4
    assert abs(out.metrics["w2_cost"] - float(zf[pre + "w2"])) < 2e-4
    assert abs(out.metrics["sparsity_cost"] - float(zf[pre + "sparsity"])) < 2e-4
    out.cost.backward()
    n_with_grad = 0
    for name, p in m.named_parameters():
        ref = torch.from_numpy(zf[pre + "g:" + name])
        got = torch.zeros_like(ref) if p.grad is None else p.grad.detach().cpu()
        scale = max(ref.abs().max().item(), 1e-6)
        assert (got - ref).abs().max().item() < 2e-4 * max(scale, 1.0), (name, (got - ref).abs().max().item(), scale)
        n_with_grad += int(ref.abs().max().item() > 0)
    assert n_with_grad == (23 if case == "A" else 11)
    # eval mode of the same module is the kernel-only path and differs from train (no rescale / clip): still runs
    m.eval()
    assert m(src).shape == (24, 64)


def test_icnn_hessian_golden(dev):
    """SingleCellICNN.hessian (2_icnn_core.py:213-241) vs the reference's, eval and train mode: third-order use of the
    re-differentiable exact-f32 matrix product (autograd of autograd of the potential, one pass per coordinate)."""
    from clip_dplm_amd import icnn
    zf = np.load(os.path.join(G, "icnn_hessian.npz"))
    sd = {k[2:]: torch.from_numpy(zf[k]) for k in zf.files if k.startswith("w:")}
    m = icnn.SingleCellICNN(icnn.ICNNConfig(input_dim=16, hidden_dims=[16, 16, 8]))
    m.load_state_dict(sd)
    m = m.to(dev)
    x = torch.from_numpy(zf["x"]).to(dev)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        h = m.hessian(x.clone())
        ref = torch.from_numpy(zf["hessian_" + mode])
        assert h.shape == ref.shape and h.requires_grad            # graph kept, as in the reference
        err = (h.detach().cpu() - ref).abs().max().item()
        assert err < 2e-4 * max(ref.abs().max().item(), 1.0), (mode, err)
    # the eval-mode transport map (kernel-only path) is the gradient the Hessian differentiates: finite differences
    m.eval()
    eps = 1e-2
    e0 = torch.zeros_like(x); e0[:, 0] = eps
    fd = (m.gradient(x + e0) - m.gradient(x - e0)) / (2 * eps)     # d T / d x_0  = H[:, 0, :]
    h = m.hessian(x.clone()).detach()
    assert (fd - h[:, 0, :]).abs().max().item() < 5e-2 * max(h.abs().max().item(), 1.0)


def test_icnn_without_layer_norm_golden(dev):
    """ICNNConfig(use_layer_norm=False, activation="softplus"): the layers have no LayerNorm (nn.Identity, :72), so the
    fused LN + activation kernels do not apply; the same exact-f32 products with the activation by ATen.  Potential and
    transport map vs the reference, eval mode; the train mode of the same module runs and back-propagates."""
    from clip_dplm_amd import icnn
    zf = np.load(os.path.join(G, "icnn_noln.npz"))
    sd = {k[2:]: torch.from_numpy(zf[k]) for k in zf.files if k.startswith("w:")}
    m = icnn.SingleCellICNN(icnn.ICNNConfig(input_dim=16, hidden_dims=[16, 16, 8], use_layer_norm=False,
                                            activation="softplus"))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = torch.from_numpy(zf["x"]).to(dev)
    psi, _ = m(x)
    assert (psi.cpu() - torch.from_numpy(zf["psi"])).abs().max().item() < 2e-4
    assert (m.gradient(x).cpu() - torch.from_numpy(zf["gradient"])).abs().max().item() < 2e-4
    m.train()
    t = m.gradient(x)
    t.square().sum().backward()
    # (without LayerNorm the z contributions are large and the train-time rescale fires: as in fixture case B of the
    # training test the rescaled path is a constant, so only the last layer's x path and `final` receive gradients)
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert {"final.weight", "layers.2.linear.weight", "layers.2.linear.bias"} <= set(grads)
    assert all(torch.isfinite(g).all() for g in grads.values())


def test_icnn_training_large_ragged_batch_vs_oracle(dev):
    """Batch 1030 (not a multiple of 4, longer than one contraction chunk of the f32 kernel): the weight-gradient
    products of the double backward are chunked and zero-padded; cost and gradients vs the CPU oracle."""
    from clip_dplm_amd import icnn
    from oracle import icnn_ref
    torch.manual_seed(5)
    m = icnn.SingleCellTransport(64, 64, icnn.ICNNConfig(input_dim=64, hidden_dims=[64, 32]))
    with torch.no_grad():
        m.transport_net.layers[1].pos_weights.normal_(0, 0.5)
        m.transport_net.layers[1].scale.fill_(0.05)
        m.transport_net.final.weight.mul_(2.0)
    g = torch.Generator().manual_seed(6)
    src, tgt = torch.randn(1030, 64, generator=g), torch.randn(1030, 64, generator=g)
    sd = {"t." + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    _, cref, _, _ = icnn_ref.single_cell_transport_train(src, tgt, sd, "t", 2)
    cref.backward()
    m = m.to(dev).train()
    out = m(src.to(dev), tgt.to(dev))
    assert abs(out.cost.item() - cref.item()) < 2e-4
    out.cost.backward()
    for name, p in m.named_parameters():
        ref = sd["t." + name].grad
        ref = torch.zeros_like(p.detach().cpu()) if ref is None else ref
        got = torch.zeros_like(ref) if p.grad is None else p.grad.detach().cpu()
        assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item()), name


def test_esm_projections_golden(dev):
    """ProteinProjection / GeneProjection (3_esm_integration.py:137-213) vs the reference (golden)."""
    from clip_dplm_amd import esm_integration as E
    z, sd = load("esm_projections.npz")
    x = t(z, "x", dev)
    for name, cls in (("protein", E.ProteinProjection), ("gene", E.GeneProjection)):
        m = cls(esm_dim=64, output_dim=32)
        m.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")})
        m = m.to(dev).eval()
        y = m(x)
        ref = t(z, "y_" + name, dev)
        assert (y - ref).abs().max().item() < 0.06, (name, (y - ref).abs().max().item())   # LN output, bf16 GEMMs
        y.sum().backward()
    ids, mask = E.tokenize(["MKV", "ACDEFGHIK"])
    assert ids.tolist()[0] == [0, 20, 15, 7, 2, 1, 1, 1, 1, 1, 1] and mask.sum().item() == 5 + 11


def test_load_state_dict_after_forward_refreshes_fused_qkv_copies(dev):
    """the ESM fused qkv weight is a zero-copy as_strided view whose own _version never moves; its bf16
    copies are keyed on the three source Parameters' versions.  FusedAdamW model, one step, then load_state_dict of
    other weights: the next grad-enabled forward must equal a fresh model holding those weights (bitwise: same
    kernels, same inputs)."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_stale"] = (2, 96, 4, 384)
    kw = dict(esm="test_stale", rna_dim=64, rna_layers=1, rna_heads=8, rna_ffn=128, projection_dim=64)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(**kw).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    torch.manual_seed(1)
    other = {k: v.detach().clone() for k, v in K.ProteinRNACLIP(**kw).state_dict().items()}
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(4, 24, (16, 32), generator=g).to(dev)
    rna = torch.randn(16, 32, 64, generator=g).to(dev)
    opt = K.FusedAdamW(m, lr=1e-3)
    opt.zero_grad(); m.loss(rna, ids).backward(); opt.step()
    m.load_state_dict(other)                               # in place, bypasses the optimiser's dirty marking
    opt.zero_grad()
    l1 = m.loss(rna, ids)
    fresh = K.ProteinRNACLIP(**kw)
    fresh.load_state_dict(other)
    fresh = fresh.to(dev).train()
    for mod in fresh.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt2 = K.FusedAdamW(fresh, lr=1e-3)
    opt2.zero_grad()
    l2 = fresh.loss(rna, ids)
    assert torch.equal(l1, l2), (l1.item(), l2.item())


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_trimodal_contrastive_model_golden(dev, precision):
    """ContrastiveModel (current/tf_clip_codes (1).ipynb:13113-13176) with the reference's state_dict,
    against the fixture built from the reference's own sub-modules and loss expressions at B = 32 (tools/make_golden.py
    gen_trimodal; `[:, 0]` on the 2-D encoder outputs is upstream defect A-19): embeddings, the three pairwise losses,
    their sum and every parameter gradient.  Default arithmetic (exact f32): north-star bar 1e-3 on every loss, met with
    orders of margin; the optional bf16 kernels are checked at bf16 level (diagnostic)."""
    import clip_dplm_amd as K
    z, sd = load("trimodal_model.npz")
    m = K.ContrastiveModel(21, 64, projection_dim=64, esm_dim=40, precision=precision)
    m.load_state_dict(sd)                                  # reference keys load as they are
    m = m.to(dev).eval()
    out = m(t(z, "cell_state", dev), t(z, "connectivity", dev), t(z, "gene_esm", dev), t(z, "gene_values", dev),
            t(z, "protein_emb", dev))
    assert set(out) == {"cell_embed", "pert_embed", "protein_embed", "loss", "cell_pert_loss", "cell_protein_loss",
                        "pert_protein_loss"}
    e_emb = max((out[k].cpu() - t(z, k)).abs().max().item() for k in ("cell_embed", "pert_embed", "protein_embed"))
    e_loss = max(abs(out[k].item() - float(z[k])) for k in ("cell_pert_loss", "cell_protein_loss", "pert_protein_loss", "loss"))
    print(f"tri-modal B={out['cell_embed'].shape[0]} [{precision}]: max |dloss| {e_loss:.2e}, max |dembed| {e_emb:.2e}")
    if precision == "f32":
        assert e_loss < 1e-4 and e_emb < 1e-4, (e_loss, e_emb)       # bar: 1e-3
    else:
        assert e_loss < 2e-2 and e_emb < 0.03, (e_loss, e_emb)       # diagnostic
    out["loss"].backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    if precision == "f32":                                 # the reference's own gradients, parameter by parameter
        gmax = max(t(z, "g:" + n).abs().max().item() for n, _ in m.named_parameters())
        for n, p in m.named_parameters():
            ref = t(z, "g:" + n)
            assert (p.grad.cpu() - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-3 * gmax), n


def test_trimodal_position0_slice_is_exact(dev):
    """As test_notebook_model_position0_slice_is_exact for the perturbation encoder's [B, G, E] input
    (tf_clip_codes (1).ipynb:13140-13142 reads `pert_enc[:, 0]`)."""
    import clip_dplm_amd as K
    z, sd = load("trimodal_model.npz")

    def build(sl):
        m = K.ContrastiveModel(21, 64, projection_dim=64, esm_dim=40, slice_first_position=sl)
        m.load_state_dict(sd)
        return m.to(dev).eval()

    def run(m):
        o = m(t(z, "cell_state", dev), t(z, "connectivity", dev), t(z, "gene_esm", dev), t(z, "gene_values", dev),
              t(z, "protein_emb", dev))
        return [o["cell_embed"], o["pert_embed"], o["protein_embed"], o["loss"]]
    print(f"tri-modal position-0 slice: bitwise equal = {_slice_equivalence(build, run)}")


def test_trimodal_loss_pairs_kernels_vs_f64(dev):
    """loss.tri_modal_loss (six directed problems in one launch per pass) against f64 torch: the three losses, and
    the gradients w.r.t. the three embedding matrices and the scale under unequal upstream weights."""
    from clip_dplm_amd.loss import tri_modal_loss
    B, P = 192, 128
    g = torch.Generator().manual_seed(2)
    E = [torch.nn.functional.normalize(torch.randn(B, P, generator=g), dim=-1).to(dev).requires_grad_(True)
         for _ in range(3)]
    ls = torch.tensor(2.6592, device=dev, requires_grad=True)
    out = tri_modal_loss(E[0], E[1], E[2], ls.exp())
    w = (1.0, 0.3, 2.0)
    (w[0] * out["cell_pert_loss"] + w[1] * out["cell_protein_loss"] + w[2] * out["pert_protein_loss"]).backward()
    Ed = [e.detach().double().requires_grad_(True) for e in E]
    lsd = ls.detach().double().requires_grad_(True)
    lab = torch.arange(B, device=dev)

    def sym(a, b):
        S = (a @ b.t()) * lsd.exp()
        return 0.5 * (torch.nn.functional.cross_entropy(S, lab) + torch.nn.functional.cross_entropy(S.t(), lab))
    ref = (sym(Ed[0], Ed[1]), sym(Ed[0], Ed[2]), sym(Ed[1], Ed[2]))
    (w[0] * ref[0] + w[1] * ref[1] + w[2] * ref[2]).backward()
    for k, r in zip(("cell_pert_loss", "cell_protein_loss", "pert_protein_loss"), ref):
        assert abs(out[k].item() - r.item()) < 1e-5
    assert abs(out["loss"].item() - sum(r.item() for r in ref)) < 2e-5
    for e, ed in zip(E, Ed):
        assert torch.allclose(e.grad.double(), ed.grad, rtol=1e-4, atol=1e-7), (e.grad.double() - ed.grad).abs().max()
    assert abs(ls.grad.item() - lsd.grad.item()) < 1e-5 * max(1.0, abs(lsd.grad.item()))


def _ragged_batch(B, L, seed, dim):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0] = 0
    rna = torch.randn(B, L, dim, generator=g)
    lp = torch.randint(L // 4, L + 1, (B,), generator=g)
    lr = torch.randint(L // 4, L + 1, (B,), generator=g)
    lp[0], lr[1] = L, L
    pmask = (torch.arange(L)[None] < lp[:, None]).long()
    rmask = (torch.arange(L)[None] < lr[:, None]).long()
    ids = torch.where(pmask.bool(), ids, torch.ones_like(ids))
    return rna, ids, rmask, pmask


@pytest.mark.parametrize("interleaved", [False, True])
def test_packed_varlen_path_equals_padded_path(dev, interleaved, monkeypatch):
    """the packed variable-length path (no padded row reaches any kernel: cu_seqlens attention, packed
    Linear / LayerNorm / pooling) against the padded path with key-padding masks on the same ragged batch: loss and
    every parameter gradient.  interleaved = False: both paths rotate q / k in place (the packed path always does: the
    epilogue's position is a row index modulo L) - same kernels and the same per-row arithmetic, so the agreement is far
    inside the bf16 level (only GEMM tile membership of a row and the split-M order of the weight-gradient sums differ).
    interleaved = True (the padded path's default): RoPE in the qkv projection's epilogue on the f32 value, one bf16 rounding
    instead of two - the two paths then differ by that rounding (bound 1e-3)."""
    import clip_dplm_amd as K
    import clip_dplm_amd.encoders as enc
    from clip_dplm_amd.data import unpad
    from clip_dplm_amd.encoders import ESM2_SHAPES
    monkeypatch.setattr(enc, "ROPE_INTERLEAVED", interleaved)
    ESM2_SHAPES["test_packed"] = (2, 96, 4, 384)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="test_packed", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=128, projection_dim=64).eval()
    m = m.to(dev)
    B, L = 24, 200                                          # L > 128: more than one key block per sequence
    rna, ids, rmask, pmask = _ragged_batch(B, L, 5, 64)
    loss_pad = m.loss(rna.to(dev), ids.to(dev), rna_mask=rmask.to(dev), protein_mask=pmask.to(dev))
    loss_pad.backward()
    gpad = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    (rp, rcu, rmax), (ip, icu, imax) = unpad(rna, rmask.bool()), unpad(ids, pmask.bool())
    assert rp.shape[0] == int(rmask.sum()) and ip.shape[0] == int(pmask.sum())
    loss_pk = m.loss_packed(rp.to(dev), rcu.to(dev), rmax, ip.to(dev), icu.to(dev), imax)
    loss_pk.backward()
    assert abs(loss_pad.item() - loss_pk.item()) < (1e-3 if interleaved else 2e-4), (loss_pad.item(), loss_pk.item())
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        a, b = p.grad.flatten().double(), gpad[n].flatten().double()
        if b.abs().max() < 1e-10:
            continue
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        assert cos > (0.995 if interleaved else 0.999), (n, cos)


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_dropout_layer_vs_masked_oracle(dev, precision):
    """(precision = "f32": the same four sites and the same masks in the exact-f32 stack - clipk_attn_f32_* and
    clipk_dropout_f32 draw the bf16 kernels' mask - at f32 tolerances.)
    nn.TransformerEncoderLayer's four dropout sites (attention probabilities, out_proj output, FFN activation, linear2
    output; current/rna_clip_codes.ipynb:1915 trains with p = 0.1) inside the kernel stack, training mode: forward and
    parameter / input gradients against the CPU oracle given the SAME masks.  The masks are counter-based
    (csrc/common.h drop_keep); ops_emulator.drop_mult reproduces them in torch integer arithmetic from the seeds the
    encoder draws.  Also: the keep rate of every site."""
    import ops_emulator as E
    import clip_dplm_amd as K
    from oracle import encoder_ref
    torch.manual_seed(0)
    Ed, H, FF, B, L, nl, pdrop = 64, 8, 128, 6, 40, 2, 0.2
    for act in ("relu", "gelu"):
        enc = K.TransformerSeqEncoder(Ed, nl, H, FF, activation=act, layer_norm_eps=1e-5, dropout=pdrop, precision=precision)
        sd = {"e." + k: v.detach().clone() for k, v in enc.state_dict().items()}
        g = torch.Generator().manual_seed(4)
        x = torch.randn(B, L, Ed, generator=g)
        dy = torch.randn(B, L, Ed, generator=g)
        lens = torch.tensor([40, 33, 17, 40, 5, 29])
        valid = torch.arange(L)[None] < lens[:, None]
        enc = enc.to(dev).train()
        torch.manual_seed(123)
        xd = x.detach().clone().to(dev).requires_grad_(True)
        y = enc(xd, src_key_padding_mask=(~valid).to(dev))
        (y * dy.to(dev)).sum().backward()
        # the same seeds, then the same masks, on the host
        torch.manual_seed(123)
        seeds = torch.randint(0, 2 ** 31 - 1, (nl, 4), dtype=torch.int64).tolist()
        T = B * L
        drops = []
        for sa, s1, sf, s2 in seeds:
            qrow = torch.arange(T, dtype=torch.int64).view(B, L)
            aidx = (qrow[:, None, :, None] * H + torch.arange(H)[None, :, None, None]) * L + torch.arange(L)[None, None, None, :]
            drops.append({"attn": E.drop_mult(pdrop, sa, aidx),
                          "d1": E.drop_mult(pdrop, s1, torch.arange(T * Ed, dtype=torch.int64)).view(B, L, Ed),
                          "ffn": E.drop_mult(pdrop, sf, torch.arange(T * FF, dtype=torch.int64)).view(B, L, FF),
                          "d2": E.drop_mult(pdrop, s2, torch.arange(T * Ed, dtype=torch.int64)).view(B, L, Ed)})
        for d in drops:
            for k, v in d.items():
                keep = (v > 0).float().mean().item()
                assert abs(keep - (1 - pdrop)) < 0.02, (k, keep)
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.detach().clone().requires_grad_(True)
        ref = encoder_ref.post_ln_encoder(xr, sdr, "e", nl, H, valid, act, 1e-5, 1e-5, drops=drops)
        (ref * dy).sum().backward()
        m = valid[..., None].float()
        err = ((y.detach().cpu() - ref.detach()) * m).abs().max().item()
        f32 = precision == "f32"
        assert err < (2e-4 if f32 else 0.06), (act, err)    # bf16 GEMM operands; LayerNorm'ed outputs are O(1)
        if f32:
            assert ((xd.grad.cpu() - xr.grad) * m).abs().max().item() < 2e-4 * max(1.0, xr.grad.abs().max().item())
        else:
            assert_grad_close(xd.grad.cpu() * m, xr.grad * m, f"{act} dx", cos_min=0.99, rel_max=0.3)
        got = dict(enc.named_parameters())
        for n in ("layers.0.self_attn.in_proj_weight", "layers.0.linear1.weight", "layers.1.linear2.weight",
                  "layers.1.self_attn.out_proj.weight", "layers.0.norm1.weight", "layernorm.weight"):
            if f32:
                r = sdr["e." + n].grad
                assert (got[n].grad.cpu() - r).abs().max().item() <= 2e-4 * max(r.abs().max().item(), 1e-3), (act, n)
            else:
                assert_grad_close(got[n].grad, sdr["e." + n].grad, f"{act} {n}", cos_min=0.99, rel_max=0.3)
        # eval mode: dropout off, deterministic
        enc.eval()
        with torch.no_grad():
            assert torch.equal(enc(xd, src_key_padding_mask=(~valid).to(dev)), enc(xd, src_key_padding_mask=(~valid).to(dev)))


def test_second_backward_through_a_stack_raises(dev):
    """The stacks free each layer's activations during the backward: a second backward is refused with a clear error
    instead of failing somewhere inside."""
    import clip_dplm_amd as K
    enc = K.TransformerSeqEncoder(64, 1, 8, 128).to(dev)
    x = torch.randn(2, 16, 64, device=dev, requires_grad=True)
    y = enc(x).sum()
    y.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        y.backward()


# ---------------------------------------------------------------------------------------------- round 3: tighter bars
def test_clip_opt_b128_golden_loss_at_the_north_star_bar(dev):
    """old/clip_opt.py at its caller's batch (B = 128, run1/full.py:189-198), the cache holding two
    earlier batches: |loss - reference| <= 1e-3 on the bf16 path, <= 2e-5 with exact-f32 Linears."""
    import clip_dplm_amd as K
    z, sd = load("clip_opt_b128.npz")
    cfg = NS(diffmap_config=sub(48), protein_config=sub(96), projection_dim=32, cache_size=512)
    for precision, bar, lbar in (("bf16", 1e-3, 0.3), ("f32", 2e-5, 2e-3)):
        m = K.OptimizedCLIPModule(cfg)
        m.load_state_dict(sd)
        m = K.set_linear_precision(m.to(dev).eval(), precision)
        nb = int(z["cache_ptr_before"])
        m.protein_embedding_cache[:nb] = t(z, "cache", dev)[:nb]
        m.cache_ptr = nb
        out = m(t(z, "diffmap", dev), t(z, "protein", dev), gather_distributed=False)
        assert m.cache_ptr == int(z["cache_ptr"])
        assert (out["logits_per_diffmap_protein"].cpu() - t(z, "logits")).abs().max().item() < lbar
        assert (out["logits_per_diffmap_cache"].cpu() - t(z, "logits_cache")).abs().max().item() < lbar
        loss = K.optimized_clip_loss(out)
        assert abs(loss.item() - float(z["loss"])) < bar, (precision, loss.item(), float(z["loss"]))
        m.cache_ptr = nb
        lf = m.loss(t(z, "diffmap", dev), t(z, "protein", dev))
        assert abs(lf.item() - float(z["loss"])) < bar, (precision, lf.item())
        lf.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_clip_c1_exact_f32_linears_reproduce_the_fp32_reference(dev):
    """the reference's caller of old/clip.py is plain fp32 (old/ablation.py:9-18).  With
    set_linear_precision(m, "f32") every Linear runs on the exact-f32 MFMA kernel: logits within 2e-4 (they are scaled by
    14.28), both losses within 1e-5, parameter gradients within 1e-4 relative of the reference's."""
    import clip_dplm_amd as K
    z, sd = load("clip_c1.npz")
    cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    m.load_state_dict(sd)
    m = K.set_linear_precision(m.to(dev).eval(), "f32")
    out = m(t(z, "rna", dev), t(z, "protein", dev))
    assert (out["logits_per_rna_protein"].cpu() - t(z, "logits")).abs().max().item() < 2e-4
    assert (out["rna_embeds"].cpu() - t(z, "rna_embeds")).abs().max().item() < 2e-6
    one = m.loss(t(z, "rna", dev), t(z, "protein", dev), symmetric=False)
    sym = m.loss(t(z, "rna", dev), t(z, "protein", dev), symmetric=True)
    assert abs(one.item() - 5.915865) < 1e-5 and abs(sym.item() - 5.939782) < 1e-5     # SURVEY §8c known answers
    sym.backward()                                          # the fixture's gradients are those of the symmetric loss
    checked = 0
    for n, p in m.named_parameters():
        key = "g:" + n
        if key in z.files:
            ref = t(z, key)
            assert (p.grad.cpu() - ref).abs().max().item() <= 1e-4 * max(1e-3, ref.abs().max().item()), n
            checked += 1
    assert checked > 0


def test_notebook_model_b32_golden_loss_and_gradients(dev):
    """the notebook model (rna_clip_codes.ipynb:1925-1954: batch-axis attention, NaN padding, position-0
    pooling) at the notebook's batch size with ragged lengths, against what the REFERENCE produced
    (tools/make_golden.py gen_notebook_b32): embeddings, loss (north-star bar 1e-3) and EVERY parameter gradient, entry
    by entry.  Default arithmetic = exact f32 with position 0 sliced before the encoders.  The optional bf16 kernels are
    run beside it as a printed diagnostic (round 3: 2.8e-3 against this fixture - a bf16-rounded weight perturbs the
    model identically for every sample and this model pools one row; that is why they are not this model's default)."""
    import clip_dplm_amd as K
    z, sd = load("notebook_model_b32.npz")
    m = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    ea, eb, loss = m(t(z, "rna", dev), t(z, "rbp", dev))
    e_emb = max((ea.cpu() - t(z, "rna_embed")).abs().max().item(), (eb.cpu() - t(z, "rbp_embed")).abs().max().item())
    e_loss = abs(loss.item() - float(z["loss"]))
    assert e_loss < 1e-4, (loss.item(), float(z["loss"]))              # bar: 1e-3
    assert e_emb < 1e-4, e_emb
    loss.backward()
    gmax = max(t(z, "g:" + n).abs().max().item() for n, _ in m.named_parameters())
    worst = 0.0
    for n, p in m.named_parameters():
        ref = t(z, "g:" + n)
        e = (p.grad.detach().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * gmax)
        worst = max(worst, e)
        assert e <= 2e-4, (n, e)
    mb = K.RNARBPCLIPModel(rna_dim=40, rbp_dim=128, projection_dim=64, precision="bf16")
    mb.load_state_dict(sd)
    mb = mb.to(dev).eval()
    with torch.no_grad():
        _, _, lb = mb(t(z, "rna", dev), t(z, "rbp", dev))
    print(f"notebook b32: |dloss| vs reference {e_loss:.2e}, max |dembed| {e_emb:.2e}, worst relative gradient error "
          f"{worst:.2e}; optional bf16 kernels (diagnostic): |dloss| {abs(lb.item() - float(z['loss'])):.2e}")
    assert abs(lb.item() - float(z["loss"])) < 2e-2


def test_esm_integration_get_embeddings_golden(dev):
    """VERDICT r02 #3b / SURVEY a12: ESMIntegration.get_embeddings end to end on the HIP path against what the
    REFERENCE's get_embeddings returned (tools/make_golden.py gen_esm_integration): tokenizer incl. truncation at
    max_sequence_length and <unk>, frozen ESM-2, both projections, the cache, get_embeddings_batch."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny96"] = (2, 96, 4, 384)
    z, sd = load("esm_integration.npz")
    m = K.create_esm_integration("test_tiny96", esm_dim=96, protein_dim=32, gene_dim=32, max_sequence_length=24)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.missing_keys, missing.missing_keys
    m = m.to(dev).eval()
    seqs = [str(s) for s in z["sequences"]]
    out = m.get_embeddings(seqs, K.BiologicalDataType.PROTEIN_SEQUENCE)
    mask = t(z, "attention_mask")[..., None].float()
    ref = t(z, "protein_embeddings")
    assert out.embeddings.shape == ref.shape                     # [6, 24, 32]: the 33-residue sequence was truncated
    err = ((out.embeddings.cpu() - ref) * mask).abs().max().item()
    assert err < 0.06, err                                       # LayerNorm outputs (O(1)) behind bf16 GEMMs
    assert m.get_embeddings(seqs, K.BiologicalDataType.PERTURBATION) is out      # cache hit, keyed on the sequences
    m.cache.clear()
    g = m.get_embeddings(seqs, K.BiologicalDataType.PERTURBATION).embeddings
    gerr = ((g.cpu() - t(z, "gene_embeddings")) * mask).abs().max().item()
    assert gerr < 0.08, gerr
    m.cache.clear()
    b = K.get_embeddings_batch(["ACDEFGHIKL", "LKIHGFEDCA", "MMMMMMMMMM"], m, K.BiologicalDataType.PROTEIN_SEQUENCE, 2)
    assert b.shape == (3, 12, 32) and torch.isfinite(b).all()


def test_cache_semantics_fifo_and_queue_loss_golden(dev):
    """VERDICT r02 #4 / SURVEY f1: true wrap-around cache (tong/utils/data.py:154-184) behind
    OptimizedCLIPModule(cache_semantics="fifo"), and the tau = 0.1 one-sided queue loss (tong/utils/losses.py:4-19) on
    the fused kernels against the reference's values."""
    import clip_dplm_amd as K
    from oracle import clip_ref
    z = np.load(os.path.join(G, "queue_loss.npz"))
    q = K.MemoryQueue(64, 16, device=dev)
    for step in range(5):
        x, y = t(z, f"x{step}", dev).requires_grad_(True), t(z, f"y{step}", dev)
        loss = K.contrastive_loss(x, y, 0.1, q.queue)
        assert abs(loss.item() - float(z[f"loss{step}"])) < 2e-5, (step, loss.item(), float(z[f"loss{step}"]))
        full = q.enqueue_dequeue(torch.nn.functional.normalize(y, dim=-1))
        assert (full.cpu() - t(z, f"queue{step}")).abs().max().item() < 1e-6 and q.ptr == int(z[f"ptr{step}"])
    loss.backward()
    xr = t(z, "x4").requires_grad_(True)
    clip_ref.contrastive_loss_queue(xr, t(z, "y4"), 0.1, t(z, "queue3")).backward()
    assert (x.grad.cpu() - xr.grad).abs().max().item() < 1e-5 * max(1.0, xr.grad.abs().max().item())
    assert abs(K.contrastive_loss(x.detach(), y, 0.1).item() - float(z["loss_noqueue"])) < 2e-5
    # module: identical to the reference semantics until the first wrap, every written row afterwards
    cfg = NS(diffmap_config=sub(48), protein_config=sub(96), projection_dim=32, cache_size=80)
    torch.manual_seed(0)
    ref_m = K.OptimizedCLIPModule(cfg).to(dev).eval()
    fifo_m = K.OptimizedCLIPModule(cfg, cache_semantics="fifo").to(dev).eval()
    fifo_m.load_state_dict(ref_m.state_dict())
    g = torch.Generator().manual_seed(3)
    oq, optr = torch.zeros(80, 32), 0
    for step in range(4):
        d, p = torch.randn(32, 48, generator=g).to(dev), torch.randn(32, 96, generator=g).to(dev)
        lr, lf = ref_m.loss(d, p), fifo_m.loss(d, p)
        _, ep = fifo_m.embed(d, p)
        oq, optr = clip_ref.memory_queue_enqueue(oq, optr, ep.detach().cpu())
        assert torch.equal(fifo_m.protein_embedding_cache.cpu(), oq) and fifo_m.cache_ptr == optr
        if step < 2:
            assert torch.equal(lr, lf)
        else:
            assert fifo_m.cache_rows().shape[0] == 80 and ref_m.cache_rows().shape[0] < 80
            # loss with all 80 rows == the oracle's loss on the same embeddings and cache rows
            ed, ep2 = fifo_m.embed(d, p)
            S = (ed @ ep2.t()).cpu() * fifo_m.logit_scale.exp().clamp(max=100).item()
            Sc = (ed.cpu() @ oq.t()) * fifo_m.logit_scale.exp().clamp(max=100).item()
            want = clip_ref.optimized_clip_loss({"logits_per_diffmap_protein": S, "logits_per_diffmap_cache": Sc})
            assert abs(lf.item() - want.item()) < 1e-4


def test_protein_rna_clip_from_config_runs(dev):
    """the model built from a HybridCLIPConfig (architectures['transformer'], projection_dim,
    logit_scale_init_value, use_mean_pooling) trains one step on the HIP path and equals the directly constructed one."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny96"] = (2, 96, 4, 384)
    cfg = K.HybridCLIPConfig(rna_config={}, protein_config={}, diffmap_config={}, projection_dim=64,
                             architectures={"transformer": K.ModelArchitectureConfig(
                                 type="transformer", num_layers=2, hidden_size=64, attention_heads=8,
                                 intermediate_size=128, dropout=0.0)})
    torch.manual_seed(0)
    a = K.ProteinRNACLIP.from_config(cfg, esm="test_tiny96").to(dev).eval()
    torch.manual_seed(0)
    b = K.ProteinRNACLIP(esm="test_tiny96", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=128,
                         projection_dim=64).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(4, 24, (16, 40), generator=g).to(dev)
    rna = torch.randn(16, 40, 64, generator=g).to(dev)
    la, lb = a.loss(rna, ids), b.loss(rna, ids)
    assert torch.equal(la, lb)
    la.backward()
    assert all(torch.isfinite(p.grad).all() for p in a.parameters() if p.grad is not None)


def test_training_with_8bit_gelu_codes_tracks_training_with_the_bf16_preactivation(dev, monkeypatch):
    """The FFN keeps GELU'(u) as 8-bit codes for its backward (encoders.GELU_AUX_U8, clipk.h aux_dtype) instead of the
    bf16 pre-activation.  The forward value is the same; the gradients differ by the code's rounding (<= 0.0025 on a factor
    in [-0.13, 1.13], no bias).  Eight optimiser steps of the same model on the same batch with either format: identical
    first loss, both going down, trajectories within 3e-3 of each other (AdamW turns ANY gradient perturbation into a
    step of size ~lr on near-zero-gradient entries, so the comparison is made at a small learning rate and over few steps;
    at lr = 2e-3 the two runs drift apart by 0.05 within six steps, as two summation orders would)."""
    import clip_dplm_amd as K
    from clip_dplm_amd import encoders
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_aux"] = (2, 96, 4, 384)
    B, L = 48, 64
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(4, 24, (B, L), generator=g).to(dev)
    rna = torch.randn(B, L, 64, generator=g).to(dev)
    traj = {}
    for u8 in (True, False):
        monkeypatch.setattr(encoders, "GELU_AUX_U8", u8)
        torch.manual_seed(0)
        m = K.ProteinRNACLIP(esm="test_aux", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=256, projection_dim=64).to(dev)
        opt = K.FusedAdamW(m, lr=3e-4, weight_decay=0.01, max_grad_norm=1.0)
        losses = []
        for _ in range(8):
            opt.zero_grad()
            loss = m.loss(rna, ids)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        traj[u8] = losses
    assert traj[True][0] == traj[False][0]                      # same forward
    assert traj[True][-1] < traj[True][0] - 0.02 and traj[False][-1] < traj[False][0] - 0.02, (traj[True], traj[False])
    assert max(abs(a - b) for a, b in zip(traj[True], traj[False])) < 3e-3, (traj[True], traj[False])
