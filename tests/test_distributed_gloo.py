"""CPU, world_size = 2 over gloo: the rank bookkeeping of the multi-GPU path.

The HIP kernels cannot run here, so tests/ops_emulator.py stands in for them inside each spawned rank; what is
under test is the product's own distributed logic: embedding all-gather + label offsets + LSE gather + local
backward (loss.py), the differentiable all-gather (distributed.py) and the sharded flat AdamW (optim.py).
Contract: distributed loss / gradients / updated weights == single-process results on the concatenated batch.
"""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _install_emulator():
    sys.path[:0] = [ROOT, HERE]
    import ops_emulator
    from clip_dplm_amd import ops
    for n in ops_emulator._NAMES:
        if hasattr(ops, n) and n != "KernelTimer":
            setattr(ops, n, getattr(ops_emulator, n))


def _unit(n, p, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, p, generator=g), dim=-1)


def _worker(rank, world, initfile, results):
    torch.set_num_threads(1)
    _install_emulator()
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace as NS

        import clip_dplm_amd as K
        from clip_dplm_amd.distributed import all_gather_with_grad
        from clip_dplm_amd.loss import clip_loss
        out = {}
        # ---- (1) fused global-batch loss: local rows of a global batch
        Bl, P = 12, 16
        a_g, b_g = _unit(world * Bl, P, 1), _unit(world * Bl, P, 2)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        a = a_g[sl].clone().requires_grad_(True)
        b = b_g[sl].clone().requires_grad_(True)
        s = torch.tensor(14.2849, requires_grad=True)
        loss = clip_loss(a, b, s, symmetric=True, group=dist.group.WORLD)
        loss.backward()
        out["loss"], out["da"], out["db"], out["ds"] = loss.item(), a.grad.clone(), b.grad.clone(), s.grad.clone()
        # ---- (2) differentiable all-gather (forward all-gather, backward reduce-scatter)
        x = a_g[sl].clone().requires_grad_(True)
        y = all_gather_with_grad(x)
        assert torch.equal(y.detach(), a_g)
        (y * torch.arange(y.numel()).view_as(y).float()).sum().backward()
        out["dx"] = x.grad.clone()
        # ---- (3) sharded flat AdamW == unsharded on the summed gradient
        sub = lambda h: NS(hidden_size=h, num_hidden_layers=1, layer_norm_eps=1e-12)
        cfg = NS(rna_config=sub(16), protein_config=sub(16), diffmap_config=sub(16), projection_dim=8,
                 logit_scale_init_value=2.6592)
        torch.manual_seed(0)
        m = K.RNAProteinCLIPModule(cfg).eval()
        opt = K.FusedAdamW(m, lr=1e-2, weight_decay=0.01, max_grad_norm=1.0, group=dist.group.WORLD)
        g = torch.Generator().manual_seed(5)
        xa, xb = torch.randn(world * 8, 16, generator=g), torch.randn(world * 8, 16, generator=g)
        for _ in range(2):
            opt.zero_grad()
            l = m.loss(xa[rank * 8:(rank + 1) * 8], xb[rank * 8:(rank + 1) * 8], symmetric=True, group=dist.group.WORLD)
            l.backward()
            opt.step()
        out["params"] = {n: p.detach().clone() for n, p in m.named_parameters()}
        out["train_loss"] = l.item()
        out["opt_state"] = opt.state_dict()                    # collective: gathers the sharded moments
        # ---- (4) the benchmark model in small: ESM stack (zero-copy fused qkv views of the sharded flat buffer,
        # pre-rotated q / k, weight gradients written straight into .grad) + post-LN stack, two sharded steps
        from clip_dplm_amd.encoders import ESM2_SHAPES
        ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
        torch.manual_seed(0)
        pm = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=1, rna_heads=8, rna_ffn=128, projection_dim=32).eval()
        popt = K.FusedAdamW(pm, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, group=dist.group.WORLD)
        g = torch.Generator().manual_seed(9)
        ids = torch.randint(4, 24, (world * 6, 10), generator=g)
        rna = torch.randn(world * 6, 10, 64, generator=g)
        sl6 = slice(rank * 6, (rank + 1) * 6)
        popt.zero_grad()
        pl = pm.loss(rna[sl6], ids[sl6], group=dist.group.WORLD)
        pl.backward()
        gsum = popt.flat.grad.detach().clone()               # this rank's contribution, written by the kernels
        dist.all_reduce(gsum)
        # by parameter: the flat layout (bucket padding) depends on the world size
        out["pgrad"] = torch.cat([gsum[o:o + p.numel()] for p, o in zip(popt.flat.params, popt.flat.offsets)])
        assert len(popt.flat.buckets) == 3                    # ESM stack | RNA stack | heads + logit_scale
        popt.step()                                           # bucketed reduce-scatter, piece update, all-gather
        out["pparams"] = torch.cat([p.detach().reshape(-1) for p in popt.flat.params])
        out["ptrain_loss"] = pl.item()
        # ---- (5) the same step with the buckets reduced FROM INSIDE the backward (the encoder stacks call back when
        # their gradients are final; on RCCL that collective runs on a side stream under the rest of the backward)
        torch.manual_seed(0)
        pm2 = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=1, rna_heads=8, rna_ffn=128, projection_dim=32).eval()
        popt2 = K.FusedAdamW(pm2, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, group=dist.group.WORLD, overlap=True)
        popt2.zero_grad()
        pm2.loss(rna[sl6], ids[sl6], group=dist.group.WORLD).backward()
        out["buckets_reduced_in_backward"] = list(popt2._reduced)
        popt2.step()
        out["pparams_overlap"] = torch.cat([p.detach().reshape(-1) for p in popt2.flat.params])
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_matches_single_process():
    world = 2
    mp.set_sharing_strategy("file_system")
    with tempfile.TemporaryDirectory() as d:
        mgr = mp.Manager()
        results = mgr.dict()
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), results), nprocs=world, join=True)
        res = [results[r] for r in range(world)]
    # ---- single-process references (same emulated kernels, no process group)
    _install_emulator()
    from types import SimpleNamespace as NS

    import clip_dplm_amd as K
    from clip_dplm_amd.loss import clip_loss
    Bl, P = 12, 16
    a = _unit(world * Bl, P, 1).requires_grad_(True)
    b = _unit(world * Bl, P, 2).requires_grad_(True)
    s = torch.tensor(14.2849, requires_grad=True)
    loss = clip_loss(a, b, s, symmetric=True, group=None)
    loss.backward()
    for r in range(world):
        sl = slice(r * Bl, (r + 1) * Bl)
        assert abs(res[r]["loss"] - loss.item()) < 1e-6
        assert torch.allclose(res[r]["da"], a.grad[sl], rtol=1e-5, atol=1e-7)
        assert torch.allclose(res[r]["db"], b.grad[sl], rtol=1e-5, atol=1e-7)
    assert abs(sum(res[r]["ds"].item() for r in range(world)) - s.grad.item()) < 1e-5   # summed by the optimiser
    # all-gather backward = reduce-scatter(sum): every rank back-propagated the same upstream gradient
    up = torch.arange(world * Bl * P).view(world * Bl, P).float() * world
    for r in range(world):
        assert torch.allclose(res[r]["dx"], up[r * Bl:(r + 1) * Bl])
    # sharded AdamW
    sub = lambda h: NS(hidden_size=h, num_hidden_layers=1, layer_norm_eps=1e-12)
    cfg = NS(rna_config=sub(16), protein_config=sub(16), diffmap_config=sub(16), projection_dim=8,
             logit_scale_init_value=2.6592)
    torch.manual_seed(0)
    m = K.RNAProteinCLIPModule(cfg).eval()
    opt = K.FusedAdamW(m, lr=1e-2, weight_decay=0.01, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(5)
    xa, xb = torch.randn(world * 8, 16, generator=g), torch.randn(world * 8, 16, generator=g)
    for _ in range(2):
        opt.zero_grad()
        l = m.loss(xa, xb, symmetric=True)
        l.backward()
        opt.step()
    ref_state = opt.state_dict()
    for r in range(world):
        assert abs(res[r]["train_loss"] - l.item()) < 1e-5
        for n, p in m.named_parameters():
            assert torch.allclose(res[r]["params"][n], p, rtol=1e-4, atol=1e-6), (r, n)
        # the sharded optimiser's checkpoint holds the FULL moments on every rank, in torch.optim.AdamW's layout
        st = res[r]["opt_state"]
        assert st["param_groups"][0]["params"] == ref_state["param_groups"][0]["params"]
        for i, ent in ref_state["state"].items():
            assert float(st["state"][i]["step"]) == float(ent["step"]) == 2.0
            assert torch.allclose(st["state"][i]["exp_avg"], ent["exp_avg"], rtol=1e-4, atol=1e-7), (r, i)
            assert torch.allclose(st["state"][i]["exp_avg_sq"], ent["exp_avg_sq"], rtol=1e-4, atol=1e-9), (r, i)
    # (4) ProteinRNACLIP, sharded vs single process on the concatenated batch
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny"] = (2, 96, 4, 384)
    torch.manual_seed(0)
    pm = K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=1, rna_heads=8, rna_ffn=128, projection_dim=32).eval()
    popt = K.FusedAdamW(pm, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(4, 24, (world * 6, 10), generator=g)
    rna = torch.randn(world * 6, 10, 64, generator=g)
    popt.zero_grad()
    pl = pm.loss(rna, ids)
    pl.backward()
    gref = torch.cat([popt.flat.grad[o:o + p.numel()] for p, o in zip(popt.flat.params, popt.flat.offsets)]).clone()
    for r in range(world):
        assert abs(res[r]["ptrain_loss"] - pl.item()) < 2e-3, (res[r]["ptrain_loss"], pl.item())
        # summed per-rank gradients == single-process gradient of the concatenated batch (bf16 operand rounding
        # differs with the batch split: compare against the largest entry, and in direction)
        g = res[r]["pgrad"]
        assert g.shape == gref.shape
        assert (g - gref).abs().max() < 0.05 * gref.abs().max(), ((g - gref).abs().max(), gref.abs().max())
        assert torch.nn.functional.cosine_similarity(g, gref, dim=0) > 0.995
    # after the sharded step every rank holds the same parameters (all-gather of the updated shards)
    assert torch.equal(res[0]["pparams"], res[1]["pparams"])
    # (5) reducing the two encoder buckets from inside the backward changes nothing but the schedule
    for r in range(world):
        assert res[r]["buckets_reduced_in_backward"] == [True, True, False]
        assert torch.equal(res[r]["pparams_overlap"], res[r]["pparams"])
