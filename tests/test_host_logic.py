"""CPU: the HOST logic of the product (hand-written layer backward wiring in encoders.py, autograd Functions,
fused-loss bookkeeping, flat optimiser) exercised with tests/ops_emulator.py standing in for the HIP kernels,
checked against the oracle's autograd.  The kernels themselves are checked on the GPU (test_gpu_*.py)."""
import math
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

import ops_emulator
from oracle import clip_ref, model_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sub(h, n=2, eps=1e-12):
    return NS(hidden_size=h, num_hidden_layers=n, layer_norm_eps=eps)


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("direct_grads", [False, True])
def test_protein_rna_clip_wiring_vs_oracle(monkeypatch, direct_grads):
    """direct_grads: .grad buffers pre-allocated by FusedAdamW (views of the flat buffer), so the weight-gradient
    kernels accumulate into them and autograd receives None for those parameters (encoders._wgrad)."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
    torch.manual_seed(0)
    m = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=128, projection_dim=64).eval()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    B, L = 24, 20
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(4, 24, (B, L), generator=g); ids[:, 0] = 0; ids[:, -1] = 2
    ids[3, 4] = 32
    rna = torch.randn(B, L, 64, generator=g)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    pmask = (torch.arange(L)[None] < lens[:, None]).long()
    rmask = (torch.arange(L)[None] < lens.flip(0)[:, None]).long()
    if direct_grads:
        opt = K.FusedAdamW(m, lr=1e-3)
        opt.zero_grad()
        assert all(p.grad is not None for p in m.parameters())
    loss = m.loss(rna, ids, rna_mask=rmask, protein_mask=pmask)
    ref, _, _ = model_ref.protein_rna_clip_loss(sd, rna, ids, rmask, pmask, esm_layers=2, esm_heads=4, rna_layers=2,
                                                rna_heads=8)
    assert abs(loss.item() - ref.item()) < 2e-3, (loss.item(), ref.item())
    loss.backward()
    ref.backward()
    bad = []
    for n, p in m.named_parameters():
        r = sd[n].grad
        if r is None or r.abs().max() < 1e-9:
            continue
        e = relerr(p.grad, r)
        if e > 0.08:
            bad.append((n, round(e, 3)))
    assert not bad, bad


def test_clip_c1_wiring_vs_golden(monkeypatch):
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    z = np.load(os.path.join(G, "clip_c1.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    m.load_state_dict(sd)
    m.eval()
    a, b = torch.from_numpy(z["rna"]), torch.from_numpy(z["protein"])
    loss = m.loss(a, b, symmetric=True)
    assert abs(loss.item() - float(z["loss_symmetric"])) < 1e-3
    loss.backward()
    # bf16 operand rounding alone moves entries of these cancellation-heavy gradients by up to ~16 % of the max
    # entry (see test_gpu_models.py); direction must still agree
    for n, p in m.named_parameters():
        ref = torch.from_numpy(z["g:" + n])
        cos = torch.nn.functional.cosine_similarity(p.grad.flatten(), ref.flatten(), dim=0).item()
        assert cos > 0.99 and relerr(p.grad, ref) < 0.25, (n, cos, relerr(p.grad, ref))
    out = m(a, b)
    l1 = torch.nn.functional.cross_entropy(out["logits_per_rna_protein"], torch.arange(256))
    assert abs(l1.item() - float(z["loss_one_sided"])) < 1e-3


def test_notebook_model_wiring(monkeypatch):
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    z = np.load(os.path.join(G, "notebook_model.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = K.RNARBPCLIPModel(rna_dim=24, rbp_dim=64, projection_dim=32)
    m.load_state_dict(sd)
    m.eval()
    ea, eb, loss = m(torch.from_numpy(z["rna"]), torch.from_numpy(z["rbp"]))
    # default arithmetic of this model: exact f32, position 0 sliced before the encoders (wiring check on the emulator)
    assert (ea - torch.from_numpy(z["rna_embed"])).abs().max().item() < 1e-4
    assert (eb - torch.from_numpy(z["rbp_embed"])).abs().max().item() < 1e-4
    assert abs(loss.item() - float(z["loss"])) < 1e-4


def test_fused_adamw_matches_torch(monkeypatch):
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    torch.manual_seed(0)
    cfg = NS(rna_config=sub(32), protein_config=sub(32), diffmap_config=sub(32), projection_dim=16,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    ref = K.RNAProteinCLIPModule(cfg)
    ref.load_state_dict(m.state_dict())
    for mod in list(m.modules()) + list(ref.modules()):
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(16, 32, generator=g), torch.randn(16, 32, generator=g)
    for _ in range(3):
        opt.zero_grad(); m.loss(a, b, symmetric=True).backward(); opt.step()
        topt.zero_grad(); ref.loss(a, b, symmetric=True).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); topt.step()
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), n


def test_tri_modal_loss_and_collate(monkeypatch):
    ops_emulator.install(monkeypatch)
    from clip_dplm_amd.data import RNARBPDataset, collate_fn, create_padding_mask
    from clip_dplm_amd.loss import tri_modal_loss
    g = torch.Generator().manual_seed(0)
    embs = [clip_ref.l2_normalize(torch.randn(20, 16, generator=g)) for _ in range(3)]
    s = torch.tensor(14.2857)
    out = tri_modal_loss(*embs, s)
    ref = sum(clip_ref.clip_loss_symmetric((a @ b.t()) * s) for a, b in ((embs[0], embs[1]), (embs[0], embs[2]), (embs[1], embs[2])))
    assert abs(out["loss"].item() - ref.item()) < 1e-5
    assert set(out) == {"loss", "cell_pert_loss", "cell_protein_loss", "pert_protein_loss"}
    ds = RNARBPDataset([np.ones((3, 4), np.float32), np.ones((5, 4), np.float32)],
                       [np.ones((7, 6), np.float32), np.ones((2, 6), np.float32)])
    rna, rbp = collate_fn([ds[0], ds[1]])
    assert rna.shape == (2, 5, 4) and rbp.shape == (2, 7, 6)
    assert create_padding_mask(rna).tolist() == [[True] * 3 + [False] * 2, [True] * 5]
    assert torch.isnan(rbp[1, 2:]).all()


def test_flat_buffer_groups_qkv_for_zero_copy_fusion(monkeypatch):
    """FusedAdamW stores ESM's query / key / value weights (and biases) back to back, so the fused [3d, d] qkv weight
    and its gradient are views, not torch.cat results; state_dict keys stay the HF ones."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    from clip_dplm_amd import encoders as E
    torch.manual_seed(0)
    enc = K.ESM2Encoder(num_layers=2, hidden_size=96, num_heads=4, intermediate_size=384)
    ref_cat = [torch.cat([l.attention.self.query.weight, l.attention.self.key.weight, l.attention.self.value.weight], 0)
               .detach().clone() for l in enc.encoder.layer]
    assert E._fused_views([enc.encoder.layer[0].attention.self.query.weight,
                           enc.encoder.layer[0].attention.self.key.weight,
                           enc.encoder.layer[0].attention.self.value.weight]) is None      # no flat buffer yet
    opt = K.FusedAdamW(enc, lr=1e-3)
    opt.zero_grad()
    for i, l in enumerate(enc.encoder.layer):
        s_ = l.attention.self
        fw = E._fused_views([s_.query.weight, s_.key.weight, s_.value.weight])
        fb = E._fused_views([s_.query.bias, s_.key.bias, s_.value.bias])
        assert fw is not None and fb is not None
        assert fw[0].shape == (288, 96) and torch.equal(fw[0], ref_cat[i])
        assert fw[0].data_ptr() == s_.query.weight.data_ptr() and fw[1].data_ptr() == s_.query.weight.grad.data_ptr()
    assert any(k.endswith("attention.self.query.weight") for k in enc.state_dict())
    # a step through the fused views updates all three Linears
    ids = torch.randint(4, 24, (3, 10))
    before = enc.encoder.layer[1].attention.self.value.weight.detach().clone()
    enc(ids).square().mean().backward()
    assert enc.encoder.layer[1].attention.self.value.weight.grad.abs().max() > 0
    opt.step()
    assert not torch.equal(before, enc.encoder.layer[1].attention.self.value.weight.detach())


def test_dropout_training_mode_runs_and_is_seeded(monkeypatch):
    """The encoder stacks implement nn.TransformerEncoderLayer's dropout: a module built with the notebook's
    p = 0.1 trains, differs from p = 0, is reproducible under torch.manual_seed, and eval mode ignores it.  (The masked
    parity test against the oracle is tests/test_gpu_models.py::test_dropout_layer_vs_masked_oracle, also emulated.)"""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    torch.manual_seed(0)
    m = K.RNARBPCLIPModel(rna_dim=24, rbp_dim=64, projection_dim=32)            # dropout = 0.1 like the notebook
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                                         # heads: ATen dropout, not under test
    g = torch.Generator().manual_seed(0)
    rna, rbp = torch.randn(4, 6, 24, generator=g), torch.randn(4, 5, 64, generator=g)
    m.train()
    torch.manual_seed(5); l1 = m(rna, rbp)[2]
    torch.manual_seed(5); l2 = m(rna, rbp)[2]
    torch.manual_seed(6); l3 = m(rna, rbp)[2]
    assert torch.equal(l1, l2) and not torch.equal(l1, l3)
    l1.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    m.eval()
    e1, e2 = m(rna, rbp)[2], m(rna, rbp)[2]
    assert torch.equal(e1, e2) and not torch.equal(e1, l1.detach())


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two fresh ranks itself (before any HIP call) and
    returns the launcher's exit code.  Here there is no GPU, so each rank stops at the 'needs an MI355X' check: the
    test asserts the launch happened (both ranks reported) and that the failure propagated."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert r.stderr.count("needs an MI355X") >= 2, r.stderr[-2000:]


def _small_clip(K):
    cfg = NS(rna_config=sub(32), protein_config=sub(32), diffmap_config=sub(32), projection_dim=16,
             logit_scale_init_value=2.6592)
    m = K.RNAProteinCLIPModule(cfg)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def test_fused_adamw_state_dict_is_torch_adamw_state_dict(monkeypatch, tmp_path):
    """SURVEY §8f-2 / ADVICE r01: FusedAdamW.state_dict() is torch.optim.AdamW's format — the `optimizer_state` of the
    reference's checkpoints (triple_flow/5_training.py:335-358) — in BOTH directions, and the reference's checkpoint
    dict round-trips through torch.save / torch.load(weights_only=True)."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    torch.manual_seed(0)
    m, ref = _small_clip(K), _small_clip(K)
    ref.load_state_dict(m.state_dict())
    opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(16, 32, generator=g), torch.randn(16, 32, generator=g)

    def fstep(mm, oo):
        oo.zero_grad(); mm.loss(a, b, symmetric=True).backward(); oo.step()

    def tstep(mm, oo):
        oo.zero_grad(); mm.loss(a, b, symmetric=True).backward()
        torch.nn.utils.clip_grad_norm_(mm.parameters(), 1.0); oo.step()
    for _ in range(3):
        fstep(m, opt); tstep(ref, topt)
    fs, ts = opt.state_dict(), topt.state_dict()
    assert set(fs) == {"state", "param_groups"} and fs["param_groups"][0]["params"] == ts["param_groups"][0]["params"]
    for i, ent in ts["state"].items():
        assert float(fs["state"][i]["step"]) == float(ent["step"])
        assert fs["state"][i]["exp_avg"].shape == ent["exp_avg"].shape
        assert torch.allclose(fs["state"][i]["exp_avg"], ent["exp_avg"], rtol=1e-4, atol=1e-7), i
        assert torch.allclose(fs["state"][i]["exp_avg_sq"], ent["exp_avg_sq"], rtol=1e-4, atol=1e-9), i
    # torch -> fused: a NEW fused optimiser resumes from torch's state and stays on torch's trajectory
    m2 = _small_clip(K)
    m2.load_state_dict(ref.state_dict())
    opt2 = K.FusedAdamW(m2, lr=5e-4, weight_decay=0.0, max_grad_norm=1.0)
    opt2.load_state_dict(ts)
    assert opt2.step_count == 3 and opt2.lr == 1e-3 and opt2.wd == 0.01
    fstep(m2, opt2); tstep(ref, topt)
    for (n, p), (_, q) in zip(m2.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), n
    # fused -> torch: torch.optim.AdamW accepts the fused optimiser's state
    ref2 = _small_clip(K)
    ref2.load_state_dict(m.state_dict())
    topt2 = torch.optim.AdamW(ref2.parameters(), lr=1e-3, weight_decay=0.01)
    topt2.load_state_dict(fs)
    fstep(m, opt); tstep(ref2, topt2)
    for (n, p), (_, q) in zip(m.named_parameters(), ref2.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), n
    # the reference's checkpoint dict, through torch.save / weights_only load
    sched = K.CosineAnnealingLR(opt, T_max=20)
    sched.step()
    path = tmp_path / "ckpt.pt"
    K.save_checkpoint(path, m, opt, sched, training_state={"epoch": 2, "step": 4, "best_val_loss": 1.5},
                      config={"learning_rate": 1e-3})
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"model_state", "optimizer_state", "scheduler_state", "training_state", "config"}
    m3 = _small_clip(K)
    opt3 = K.FusedAdamW(m3, lr=1.0)
    sched3 = K.CosineAnnealingLR(opt3, T_max=5)
    ck = K.load_checkpoint(path, m3, opt3, sched3)
    assert ck["training_state"]["epoch"] == 2 and opt3.step_count == opt.step_count
    assert sched3.last_epoch == sched.last_epoch and abs(opt3.lr - opt.lr) < 1e-12
    for (n, p), (_, q) in zip(m3.named_parameters(), m.named_parameters()):
        assert torch.equal(p, q), n


def test_cosine_schedule_and_early_stopping_match_the_reference_loop():
    """CosineAnnealingLR == torch.optim.lr_scheduler.CosineAnnealingLR(T_max=20) (rna_clip_codes.ipynb:2034);
    EarlyStopping follows rna_clip_codes.ipynb:2002-2027 call by call."""
    import clip_dplm_amd as K
    p = torch.nn.Parameter(torch.zeros(3))
    topt = torch.optim.AdamW([p], lr=1e-4)
    tsch = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=20)
    holder = NS(lr=1e-4)
    sch = K.CosineAnnealingLR(holder, T_max=20)
    for _ in range(45):                                      # past T_max: the cosine keeps going, as torch's does
        assert abs(sch.get_last_lr()[0] - tsch.get_last_lr()[0]) < 1e-12
        assert abs(holder.lr - topt.param_groups[0]["lr"]) < 1e-12
        topt.step(); tsch.step(); sch.step()
    es = K.EarlyStopping(patience=2, min_delta=0.0)
    seq = [(1.0, False, False), (0.9, True, False), (0.95, False, False), (0.91, False, True)]
    for loss, ret, stop in seq:
        assert es(loss) is ret and es.early_stop is stop
    assert es.best_loss == 0.9 and es.counter == 2


def test_forward_pooled_equals_pooling_the_hidden_states(monkeypatch):
    """Host wiring of the fused final-LayerNorm + mean pooling (forward_pooled, the `pool` flag of the two stack
    Functions, the POOL backward): same pooled rows and the same parameter / input gradients as pooling forward()'s
    [B, L, d] output, with ragged masks; the kernels are emulated (tests/ops_emulator.py)."""
    import ops_emulator
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES, pool
    ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
    torch.manual_seed(0)
    B, L = 5, 12
    lens = torch.tensor([12, 7, 1, 12, 9])
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(B, 96, generator=g)

    enc = K.ESM2Encoder(num_layers=2, hidden_size=96, num_heads=4, intermediate_size=384).eval()
    ids = torch.randint(4, 24, (B, L), generator=g)
    ref = pool(enc(ids, attention_mask=mask), mask, "mean")
    (ref * w).sum().backward()
    gref = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    enc.zero_grad()
    got = enc.forward_pooled(ids, attention_mask=mask)
    (got * w).sum().backward()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
    for n, p in enc.named_parameters():
        if n in gref:
            assert torch.allclose(p.grad, gref[n], rtol=1e-3, atol=1e-5), n

    rna = K.TransformerSeqEncoder(embed_dim=64, num_layers=1, nhead=8, dim_feedforward=128).eval()
    x = torch.randn(B, L, 64, generator=g).requires_grad_(True)
    w2 = torch.randn(B, 64, generator=g)
    ref = pool(rna(x, src_key_padding_mask=~mask.bool()), mask, "mean")
    (ref * w2).sum().backward()
    gx, gref = x.grad.clone(), {n: p.grad.clone() for n, p in rna.named_parameters() if p.grad is not None}
    x.grad = None
    rna.zero_grad()
    got = rna.forward_pooled(x, src_key_padding_mask=~mask.bool())
    (got * w2).sum().backward()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(x.grad, gx, rtol=1e-2, atol=1e-4)          # (bf16 gradient streams inside the emulated stack)
    for n, p in rna.named_parameters():
        if n in gref:
            assert torch.allclose(p.grad, gref[n], rtol=1e-2, atol=1e-4), n


def test_fused_adamw_state_dict_with_frozen_submodule(monkeypatch):
    """the reference builds torch.optim.AdamW(model.parameters()) over ALL parameters, frozen encoder ones
    included (triple_flow/5_training.py:128 on a model whose ESM parameters are frozen, 3_esm_integration.py:83-84), so
    `param_groups[0]['params']` indexes every parameter and only the trainable ones carry state.  Both directions."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    torch.manual_seed(0)
    m, ref = _small_clip(K), _small_clip(K)
    ref.load_state_dict(m.state_dict())
    for mm in (m, ref):
        for p in mm.rna_model.parameters():            # a frozen tower in the middle of module.parameters()
            p.requires_grad_(False)
    opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=None)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(16, 32, generator=g), torch.randn(16, 32, generator=g)
    for _ in range(2):
        opt.zero_grad(); m.loss(a, b, symmetric=True).backward(); opt.step()
        topt.zero_grad(); ref.loss(a, b, symmetric=True).backward(); topt.step()
    fs, ts = opt.state_dict(), topt.state_dict()
    n_all = len(list(m.parameters()))
    assert fs["param_groups"][0]["params"] == ts["param_groups"][0]["params"] == list(range(n_all))
    assert set(fs["state"]) == set(ts["state"])                   # no entries for the frozen parameters
    frozen = {i for i, p in enumerate(m.parameters()) if not p.requires_grad}
    assert frozen and not (frozen & set(fs["state"]))
    for i, ent in ts["state"].items():
        assert torch.allclose(fs["state"][i]["exp_avg"], ent["exp_avg"], rtol=1e-4, atol=1e-7), i
    # torch -> fused and fused -> torch
    m2 = _small_clip(K)
    m2.load_state_dict(ref.state_dict())
    for p in m2.rna_model.parameters():
        p.requires_grad_(False)
    opt2 = K.FusedAdamW(m2, lr=1.0, max_grad_norm=None)
    opt2.load_state_dict(ts)
    assert opt2.step_count == 2
    opt2.zero_grad(); m2.loss(a, b, symmetric=True).backward(); opt2.step()
    topt.zero_grad(); ref.loss(a, b, symmetric=True).backward(); topt.step()
    for (n, p), (_, q) in zip(m2.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), n
    ref3 = _small_clip(K)
    for p in ref3.rna_model.parameters():
        p.requires_grad_(False)
    torch.optim.AdamW(ref3.parameters(), lr=1e-3).load_state_dict(fs)   # torch accepts the group size


def _one_rank_gloo(tmp_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg_init", rank=0, world_size=1)
    return dist


@pytest.mark.parametrize("mode", ["micro_batches", "two_backwards", "micro_batches+two_backwards",
                                  "micro_batches+zero_grad_after_forward"])
def test_overlapped_bucket_reduce_waits_for_every_backward(monkeypatch, tmp_path, mode):
    """with `overlap=True` the encoder stacks start their bucket's reduce-scatter from inside the
    backward.  When a stack runs backward more than once per optimiser step (micro-batches; gradient accumulation over
    several backward() calls) the bucket must be sent once, with the complete gradient: overlap == no overlap, bit for
    bit, and the flat gradient the step saw is the sum of every backward."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    dist = _one_rank_gloo(tmp_path)
    try:
        ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
        g = torch.Generator().manual_seed(9)
        ids = torch.randint(4, 24, (8, 10), generator=g)
        rna = torch.randn(8, 10, 64, generator=g)
        res = {}
        for overlap in (False, True):
            torch.manual_seed(0)
            m = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=1, rna_heads=8, rna_ffn=128,
                                 projection_dim=32).eval()
            if "micro_batches" in mode:
                m.micro_batches = 2
            opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, group=dist.group.WORLD, overlap=overlap)
            sent = []
            orig = opt._reduce_bucket
            opt._reduce_bucket = lambda b, _o=orig: (sent.append((b, opt._reduced[b])), _o(b))[1]
            if overlap:                                   # the callbacks captured the bound method: re-point them
                for b, (_, _, root) in enumerate(opt.flat.buckets):
                    if root is not None:
                        root._grad_bucket_done = (lambda b=b: opt._reduce_bucket(b))
            if "zero_grad_after_forward" in mode:         # forward -> zero_grad() -> backward (a common order)
                opt.zero_grad()
                loss = m.loss(rna, ids)
                opt.zero_grad()                           # must not forget that two backwards of each stack are outstanding
                loss.backward()
            else:
                opt.zero_grad()
                m.loss(rna, ids).backward()
            if "two_backwards" in mode:
                m.loss(rna.flip(0), ids).backward()
            gfull = opt.flat.grad.clone()
            nsq = opt.step().clone()
            res[overlap] = (torch.cat([p.detach().reshape(-1) for p in opt.flat.params]), gfull, nsq, opt.gshard.clone())
        p0, g0, n0, s0 = res[False]
        p1, g1, n1, s1 = res[True]
        assert torch.equal(g0, g1)
        assert torch.equal(s0, s1), "the reduced gradient shard differs: a bucket was sent before it was final"
        assert torch.equal(n0, n1) and torch.equal(p0, p1)
        assert torch.equal(s1, g1)                        # world 1: the shard IS the whole flat gradient
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- round-3 boundary rows
def test_esm_integration_surface_matches_the_reference():
    """VERDICT r02 #1 / SURVEY a12: ESMConfig (triple_flow/1_config.py:153-183), ESMIntegration(config)
    (3_esm_integration.py:53-72), create_esm_integration / get_embeddings_batch (:215-245): construction, attributes,
    state_dict keys (== the reference module's, recorded in the fixture), the tokenizer, the errors."""
    import dataclasses

    import clip_dplm_amd as K
    from clip_dplm_amd import esm_integration as E
    from clip_dplm_amd.encoders import ESM2_SHAPES
    cfg = K.ESMConfig()
    assert [f.name for f in dataclasses.fields(cfg)] == ["model_name", "esm_dim", "protein_dim", "gene_dim",
                                                          "num_attention_heads", "dropout", "use_sequence_context",
                                                          "max_sequence_length", "tokenizer_path"]
    assert (cfg.model_name, cfg.esm_dim, cfg.protein_dim, cfg.gene_dim, cfg.max_sequence_length) == \
        ("esm2_t33_650M_UR50D", 1280, 512, 512, 1024)
    with pytest.raises(ValueError, match="Invalid ESM model"):
        K.ESMConfig(model_name="esm1b").validate_model()
    assert {m.name: m.value for m in K.BiologicalDataType}["PROTEIN_SEQUENCE"] == "protein_sequence"
    ESM2_SHAPES["test_tiny96"] = (2, 96, 4, 384)
    m = K.ESMIntegration(K.ESMConfig(model_name="test_tiny96", esm_dim=96, protein_dim=32, gene_dim=32,
                                     max_sequence_length=24))
    assert m.config.max_sequence_length == 24 and m.cache == {}
    assert not any(p.requires_grad for p in m.model.parameters())                  # :83-84
    assert all(p.requires_grad for p in m.protein_projection.parameters())
    z = np.load(os.path.join(G, "esm_integration.npz"))
    ref_keys = {k[2:] for k in z.files if k.startswith("w:")}
    own = set(m.state_dict())
    # (EsmModel's buffers and its contact-prediction head — not on the path — are the only reference keys without a twin)
    extra = {k for k in ref_keys - own if not (k.endswith("position_ids") or "inv_freq" in k or "contact_head" in k)}
    assert not extra and not (own - ref_keys), (sorted(extra)[:5], sorted(own - ref_keys)[:5])
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}, strict=False)
    # keyword alias of earlier rounds, and the factory
    m2 = K.ESMIntegration("test_tiny96", protein_dim=16, gene_dim=8, max_sequence_length=50)
    assert m2.config.esm_dim == 96 and m2.max_sequence_length == 50
    m3 = K.create_esm_integration("test_tiny96", esm_dim=96, protein_dim=16)
    assert isinstance(m3, K.ESMIntegration) and m3.config.protein_dim == 16
    with pytest.raises(ValueError, match="hidden size"):
        K.ESMIntegration(K.ESMConfig(model_name="test_tiny96"))                    # esm_dim 1280 != 96
    # tokenizer == transformers.EsmTokenizer on the fixture's vectors
    ids, mask = E.tokenize([str(s) for s in z["sequences"]], 24)
    assert torch.equal(ids, torch.from_numpy(z["input_ids"])) and torch.equal(mask, torch.from_numpy(z["attention_mask"]))
    ids, mask = E.tokenize([str(s) for s in z["edge_sequences"]], 16)
    assert torch.equal(ids, torch.from_numpy(z["edge_input_ids"]))
    assert torch.equal(mask, torch.from_numpy(z["edge_attention_mask"]))


def test_esm_integration_get_embeddings_emulated(monkeypatch):
    """get_embeddings end to end on the emulated kernels vs the reference's outputs: truncation, cache hit keyed on the
    sequences only (3_esm_integration.py:100-102), get_embeddings_batch."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny96"] = (2, 96, 4, 384)
    z = np.load(os.path.join(G, "esm_integration.npz"))
    m = K.ESMIntegration(K.ESMConfig(model_name="test_tiny96", esm_dim=96, protein_dim=32, gene_dim=32,
                                     max_sequence_length=24)).eval()
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}, strict=False)
    seqs = [str(s) for s in z["sequences"]]
    out = m.get_embeddings(seqs, K.BiologicalDataType.PROTEIN_SEQUENCE)
    mask = torch.from_numpy(z["attention_mask"])[..., None].float()
    ref = torch.from_numpy(z["protein_embeddings"])
    assert out.embeddings.shape == ref.shape and out.attention_weights is None
    assert ((out.embeddings - ref) * mask).abs().max().item() < 0.08             # bf16 GEMM emulation, LN outputs
    assert m.get_embeddings(seqs, K.BiologicalDataType.PERTURBATION) is out       # cache ignores data_type
    m.cache.clear()
    g = m.get_embeddings(seqs, K.BiologicalDataType.GENE_EXPRESSION).embeddings
    assert (g - torch.from_numpy(z["gene_embeddings"])).abs().max().item() < 0.1
    m.cache.clear()
    same_len = ["ACDEFGHIKL", "LKIHGFEDCA", "MMMMMMMMMM", "ACDEFGHIKA"]
    b = K.get_embeddings_batch(same_len, m, K.BiologicalDataType.PROTEIN_SEQUENCE, batch_size=2)
    assert b.shape == (4, 12, 32)
    with pytest.raises(RuntimeError):                                           # ragged slices: torch.cat raises, as upstream
        K.get_embeddings_batch(["ACD", "ACDE", "ACDEFGH", "A"], m, K.BiologicalDataType.PROTEIN_SEQUENCE, batch_size=2)


def test_protein_rna_clip_from_config():
    """HybridCLIPConfig -> model.  architectures['transformer'] (run1/configuration_hybrid_clip.py
    :153-157 + :68-79), projection_dim, logit_scale_init_value, use_mean_pooling are consumed."""
    import clip_dplm_amd as K
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny96"] = (2, 96, 4, 384)
    cfg = K.HybridCLIPConfig(rna_config={}, protein_config={}, diffmap_config={})
    arch = cfg.architectures["transformer"]
    assert (arch.num_layers, arch.hidden_size, arch.attention_heads, arch.intermediate_size, arch.hidden_act,
            arch.layer_norm_eps) == (6, 768, 8, 2048, "gelu", 1e-12)
    m = K.ProteinRNACLIP.from_config(cfg, esm="test_tiny96")
    r = m.rna_model
    assert (r.embed_dim, r.num_layers, r.nhead, r.activation, r.eps, r.dropout) == (768, 6, 8, "gelu", 1e-12, 0.1)
    assert r.layers[0].linear1.weight.shape == (2048, 768)
    assert m.rna_projection.projection[4].weight.shape == (512, 1024) and m.pooling == "mean"
    assert abs(m.logit_scale.item() - 2.6592) < 1e-6 and m.config is cfg
    # the same keys as the directly constructed BASELINE model
    direct = K.ProteinRNACLIP(esm="test_tiny96")
    assert set(m.state_dict()) == set(direct.state_dict())
    cfg2 = K.HybridCLIPConfig(rna_config={}, protein_config={}, diffmap_config={}, projection_dim=64,
                              logit_scale_init_value=1.5, use_mean_pooling=False,
                              architectures={"transformer": K.ModelArchitectureConfig(
                                  type="transformer", num_layers=2, hidden_size=64, attention_heads=4,
                                  intermediate_size=128, hidden_act="relu", layer_norm_eps=1e-5, dropout=0.0)})
    m2 = K.ProteinRNACLIP.from_config(cfg2, esm="test_tiny96", freeze_protein_encoder=True)
    assert (m2.rna_model.embed_dim, m2.rna_model.num_layers, m2.rna_model.nhead, m2.rna_model.activation) == (64, 2, 4, "relu")
    assert m2.pooling == "first" and abs(m2.logit_scale.item() - 1.5) < 1e-6
    assert m2.protein_projection.projection[4].weight.shape == (64, 128)
    assert not any(p.requires_grad for p in m2.protein_model.parameters())
    with pytest.raises(ValueError, match="transformer"):
        K.ProteinRNACLIP.from_config(cfg, architecture="mlp")


def test_cache_semantics_reference_vs_fifo(monkeypatch):
    """SURVEY §8f-1 / VERDICT r02 #4: `cache_semantics="reference"` = old/clip_opt.py:76-81 (reset to 0 on overflow,
    cache[:ptr]); "fifo" = tong/utils/data.py:154-184 (true wrap-around, every row written so far).  Identical until
    the first wrap; the fifo rows equal the oracle's MemoryQueue restatement (pinned by queue_loss.npz)."""
    ops_emulator.install(monkeypatch)
    import clip_dplm_amd as K
    cfg = NS(diffmap_config=sub(16), protein_config=sub(16), projection_dim=8, cache_size=40)
    torch.manual_seed(0)
    ref_m, fifo_m = K.OptimizedCLIPModule(cfg).eval(), K.OptimizedCLIPModule(cfg, cache_semantics="fifo").eval()
    fifo_m.load_state_dict(ref_m.state_dict())
    with pytest.raises(ValueError):
        K.OptimizedCLIPModule(cfg, cache_semantics="lifo")
    g = torch.Generator().manual_seed(3)
    oq, optr = torch.zeros(40, 8), 0
    for step in range(4):                                              # 4 x 16 rows into 40: wraps at step 2
        d, p = torch.randn(16, 16, generator=g), torch.randn(16, 16, generator=g)
        lr, lf = ref_m.loss(d, p), fifo_m.loss(d, p)
        _, ep = fifo_m.embed(d, p)
        oq, optr = clip_ref.memory_queue_enqueue(oq, optr, ep.detach())
        assert fifo_m.cache_ptr == optr and fifo_m.cache_filled == min(40, 16 * (step + 1))
        assert torch.equal(fifo_m.protein_embedding_cache, oq)
        if step < 2:
            assert torch.equal(lr, lf) and ref_m.cache_ptr == fifo_m.cache_ptr
        else:
            assert ref_m.cache_rows().shape[0] < fifo_m.cache_rows().shape[0] == 40
            assert lf.item() > lr.item()                               # more negatives: larger partition function
    out = fifo_m(d, p, gather_distributed=False)
    assert out["logits_per_diffmap_cache"].shape == (16, 40)
    # MemoryQueue mirror: same states as the reference's class on the fixture's batches
    z = np.load(os.path.join(G, "queue_loss.npz"))
    q = K.MemoryQueue(64, 16)
    for step in range(5):
        full = q.enqueue_dequeue(torch.nn.functional.normalize(torch.from_numpy(z[f"y{step}"]), dim=-1))
        assert torch.equal(full, torch.from_numpy(z[f"queue{step}"])) and q.ptr == int(z[f"ptr{step}"])
    # contrastive_loss (tong/utils/losses.py:4-19) on the emulated fused kernels vs the reference's values
    q2 = torch.zeros(64, 16)
    for step in range(5):
        x, y = torch.from_numpy(z[f"x{step}"]), torch.from_numpy(z[f"y{step}"])
        assert abs(K.contrastive_loss(x, y, 0.1, q2).item() - float(z[f"loss{step}"])) < 1e-4
        q2 = torch.from_numpy(z[f"queue{step}"])
    assert abs(K.contrastive_loss(x, y, 0.1).item() - float(z["loss_noqueue"])) < 1e-4


def _run_bench(extra, env_extra=None, drop_dist_env=True):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if not (drop_dist_env and k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"))}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, capture_output=True, text=True,
                          env=env, timeout=600)


def test_bench_self_launch_plumbing_with_two_gloo_ranks():
    """VERDICT r02 #8 (first contact with a multi-GPU node, rehearsed on CPU): `bench.py --gpus 2` starts its own two
    ranks through torch.distributed.run; rank 0's JSON line is the ONLY thing on stdout (library chatter and the other
    rank stay off it); a failing child gives a non-zero return code; a WORLD_SIZE / --gpus mismatch exits cleanly."""
    import json
    r = _run_bench(["--config", "stub", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["checksum"] == 1024 * 2.0 * (1 + 2)                     # the all-reduce saw both ranks
    # the line says what the collective library saw (VERDICT r03 #6a; "rccl" on the GPU path, gloo in this rehearsal)
    assert out["rccl"]["world"] == 2 and out["rccl"]["ranks_seen"] == 2 and out["rccl"]["backend"] == "gloo"
    assert sorted(d["rank"] for d in out["rccl"]["devices"]) == [0, 1]
    assert "library chatter" in r.stderr and "library chatter" not in r.stdout
    # a rank that dies after the rendezvous: the launcher's return code is the bench's
    r = _run_bench(["--config", "stub", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                   {"CLIPK_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0 and not r.stdout.strip()
    assert "rank 1 fails on request" in r.stderr
    # under a launcher whose world size is not --gpus: one clean message, no traceback, no JSON
    r = _run_bench(["--config", "stub", "--gpus", "4", "--steps", "1", "--warmup", "0"],
                   {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop_dist_env=False)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr and "Traceback" not in r.stderr
    assert not r.stdout.strip()
    # one rank, no launcher: same line
    r = _run_bench(["--config", "stub", "--gpus", "1", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1


def test_weight_cache_never_refreshes_from_a_per_call_copy(monkeypatch):
    """a padded / concatenated operand is rebuilt per call; after a no_grad forward the cache's
    `src` was a LEAF copy of the weights at that moment, the batched refresh after the next optimiser step rebuilt the
    bf16 copies from it and stamped the new key on them: training went on with eval-time weights.  Sequence: train
    forward, no_grad forward, weight update (fused: epoch bump), batched-refresh selection, train forward."""
    ops_emulator.install(monkeypatch)
    import torch.nn.functional as F
    from clip_dplm_amd import functional as KF
    w = torch.nn.Parameter(torch.randn(16, 21))
    pad = lambda: F.pad(w, (0, 11))
    cache = KF.WeightCache()
    vf = KF.params_version(w)
    cache.get(pad(), vf)                                         # training forward: non-leaf copy
    with torch.no_grad():
        cache.get(pad(), vf)                                     # evaluation forward: a leaf copy, cache HIT
    assert cache.derived
    with torch.no_grad():
        w.mul_(2.0)                                              # optimiser step ...
    KF.mark_weights_dirty()                                      # ... by the fused kernels (no version bump of their own)
    assert all(c is not cache for c, _, _ in KF._stale_caches(require_cuda=False))
    wb, wtb = cache.get(pad(), vf)                               # next training forward: lazy rebuild from THIS call's copy
    assert torch.equal(wb.float(), pad().detach().to(torch.bfloat16).float())
    assert torch.equal(wtb.float(), pad().detach().t().to(torch.bfloat16).float())
    # a zero-copy view of the parameters' own storage (ESM's fused qkv) stays eligible for the batched refresh
    q, k = torch.nn.Parameter(torch.randn(8, 32)), torch.nn.Parameter(torch.randn(8, 32))
    flat = torch.cat([q.data.flatten(), k.data.flatten()])
    q.data, k.data = flat[:256].view(8, 32), flat[256:].view(8, 32)
    view = torch.as_strided(q.data, (16, 32), (32, 1))
    c2 = KF.WeightCache()
    c2.get(view, KF.params_version(q, k))
    assert not c2.derived
    KF.mark_weights_dirty()
    assert any(c is c2 for c, _, _ in KF._stale_caches(require_cuda=False))


def test_deferred_layernorm_parameter_gradients(monkeypatch):
    """functional.LayerNormFn with `.grad` buffers in place: the row pass leaves partial rows in a buffer of the LayerNorm's own
    and ONE batched reduce runs when the backward pass is over (nn.LayerNorm's weight.grad / bias.grad, old/clip.py:12,28,32).
    Host logic: gradients equal torch's, also when one LayerNorm is applied twice in a pass (the second use reduces at once),
    over two accumulating passes, and partial rows of a pass that never finished are discarded by zero_grad()."""
    ops_emulator.install(monkeypatch)
    import torch.nn.functional as F
    from clip_dplm_amd import functional as KF
    monkeypatch.setattr(KF, "DEFER_LN_PARAM_GRADS", True)
    calls = []
    real = KF.ops.colreduce_entries
    monkeypatch.setattr(KF.ops, "colreduce_entries", lambda e: (calls.append(len(e)), real(e))[1])
    torch.manual_seed(0)
    g1, b1 = torch.randn(24).requires_grad_(True), torch.randn(24).requires_grad_(True)
    g2, b2 = torch.randn(24).requires_grad_(True), torch.randn(24).requires_grad_(True)
    x = torch.randn(10, 24)

    def loss_of(ln):
        h = ln(x, g1, b1)
        h = ln(h * 1.5, g2, b2)
        return (ln(h + 0.3, g1, b1) ** 2).sum()             # g1 / b1 used twice
    ref = torch.autograd.grad(loss_of(lambda t, g, b: F.layer_norm(t, (24,), g, b, 1e-5)), (g1, b1, g2, b2))
    for p in (g1, b1, g2, b2):
        p.grad = torch.zeros_like(p)                         # FusedAdamW's situation: buffers exist
    loss_of(lambda t, g, b: KF.layer_norm(t, g, b, 1e-5)).backward()
    assert calls == [2]                                      # one batched reduce for the two deferred LayerNorms
    for p, r in zip((g1, b1, g2, b2), ref):
        assert torch.allclose(p.grad, r, rtol=1e-4, atol=1e-5)
    loss_of(lambda t, g, b: KF.layer_norm(t, g, b, 1e-5)).backward()      # a second pass accumulates
    assert calls == [2, 2]
    for p, r in zip((g1, b1, g2, b2), ref):
        assert torch.allclose(p.grad, 2 * r, rtol=1e-4, atol=1e-5)
    # a pass that raised leaves pending partial rows behind: zero_grad()'s discard forgets them
    KF._LN_PENDING.append((torch.ones(48), 1, 24, g1.grad, b1.grad))
    KF._LN_PENDING_KEYS.add(("stale",))
    KF.discard_deferred_ln_param_grads()
    for p in (g1, b1, g2, b2):
        p.grad.zero_()
    loss_of(lambda t, g, b: KF.layer_norm(t, g, b, 1e-5)).backward()
    for p, r in zip((g1, b1, g2, b2), ref):
        assert torch.allclose(p.grad, r, rtol=1e-4, atol=1e-5)
