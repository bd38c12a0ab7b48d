"""GPU, real RCCL ("nccl" backend): the multi-rank path of DESIGN.md §6 with the HIP kernels, one process per GPU.

world = 1 runs on a one-GPU box (CLIPK_FORCE_DIST keeps the collective code path: all-gather of embeddings, LSE
gather, bucketed reduce-scatter from inside the backward on the side stream, sharded AdamW, parameter all-gather);
world = 2 needs two GPUs and is skipped otherwise.  Contract: distributed loss, gradients and updated
weights == the single-process result on the concatenated batch.  The rank bookkeeping itself is covered on CPU by
tests/test_distributed_gloo.py; this file is about the collectives on device buffers (in-place all_gather_into_tensor
on the flat parameter buffer, reduce_scatter on views of the flat gradient, side-stream ordering).
"""
import os
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TINY = (2, 96, 4, 384)          # layers, hidden, heads (hd 24, the ESM-2-35M head shape), ffn
BL, L = 24, 40                  # pairs per rank, tokens


def _model(K):
    from clip_dplm_amd.encoders import ESM2_SHAPES
    ESM2_SHAPES["test_tiny"] = TINY
    torch.manual_seed(0)
    return K.ProteinRNACLIP(esm="test_tiny", rna_dim=64, rna_layers=2, rna_heads=8, rna_ffn=128, projection_dim=64).eval()


def _batch(world):
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(4, 24, (world * BL, L), generator=g)
    ids[:, 0] = 0
    ids[:, -1] = 2
    return ids, torch.randn(world * BL, L, 64, generator=g)


def _steps(K, m, opt, ids, rna, group, dev):
    out = {"init": torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()}
    for it in range(2):
        opt.zero_grad()
        loss = m.loss(rna.to(dev), ids.to(dev), group=group)
        loss.backward()
        if it == 0:
            out["loss"] = loss.item()
            out["reduced_in_backward"] = list(getattr(opt, "_reduced", []))
            if not getattr(opt, "overlap", False):              # this rank's contribution, by parameter
                out["grad"] = torch.cat([opt.flat.grad[o:o + p.numel()].float()
                                         for p, o in zip(opt.flat.params, opt.flat.offsets)]).cpu()
                out["grad_mp"] = torch.cat([p.grad.detach().float().reshape(-1) for p in m.parameters()]).cpu()
        opt.step()
    torch.cuda.synchronize()
    out["loss2"] = loss.item()
    out["params"] = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
    return out


def _worker(rank, world, initfile, results, backend="nccl"):
    os.environ["CLIPK_FORCE_DIST"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path[:0] = [ROOT]
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", init_method=f"file://{initfile}", rank=rank, world_size=world, device_id=dev)
    else:                                   # every rank on cuda:0, collectives staged through the host by gloo
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        import clip_dplm_amd as K
        ids, rna = _batch(world)
        sl = slice(rank * BL, (rank + 1) * BL)
        res = {}
        for overlap in (False, True):
            m = _model(K).to(dev)
            opt = K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, group=dist.group.WORLD, overlap=overlap)
            r = _steps(K, m, opt, ids[sl], rna[sl], dist.group.WORLD, dev)
            res["overlap" if overlap else "plain"] = r
        results[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "nccl"), (2, "gloo")])
def test_rccl_ranks_match_single_process(dev, world, backend):
    """(2, "gloo"): two ranks on ONE GPU, the collectives through gloo - not the product's transport, but the only way a
    one-GPU box can run the HIP kernels with world > 1: sharded gradients / moments / parameter pieces of the flat buffers
    on the device, LSE gather with label offsets, the bucket bookkeeping of two ranks."""
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    import torch.multiprocessing as mp
    mp.set_sharing_strategy("file_system")
    with tempfile.TemporaryDirectory() as d:
        mgr = mp.Manager()
        results = mgr.dict()
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), results, backend), nprocs=world, join=True)   # fresh processes
        res = [results[r] for r in range(world)]
    import clip_dplm_amd as K
    ids, rna = _batch(world)
    m = _model(K).to(dev)
    ref = _steps(K, m, K.FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0), ids, rna, None, dev)
    cos = torch.nn.functional.cosine_similarity
    gsum = sum(res[r]["plain"]["grad"] for r in range(world))
    assert cos(gsum, ref["grad"], dim=0) > 0.995                                    # summed rank gradients
    assert (gsum - ref["grad"]).abs().max() < 0.05 * ref["grad"].abs().max()
    live, o = torch.zeros_like(ref["grad_mp"], dtype=torch.bool), 0
    for p in m.parameters():                                   # per parameter: small tensors are judged on their own scale
        gp = ref["grad_mp"][o:o + p.numel()].abs()
        live[o:o + p.numel()] = gp >= 0.01 * gp.max().clamp_min(1e-30)
        o += p.numel()
    assert live.float().mean() > 0.2
    for r in range(world):
        for mode in ("plain", "overlap"):
            got = res[r][mode]
            assert abs(got["loss"] - ref["loss"]) < 1e-3, (mode, got["loss"], ref["loss"])
            assert abs(got["loss2"] - ref["loss2"]) < 2e-3, (mode, got["loss2"], ref["loss2"])
            # AdamW's first steps move a weight by ~lr whatever its gradient's size, so entries whose gradient is
            # rounding noise (e.g. the key bias, analytically zero) may step the other way: the update's direction is
            # compared on the entries whose single-process gradient is not negligible (>= 1 % of the largest entry of
            # its own parameter), where a mis-sharded bucket of small parameters would show
            du, dr = got["params"] - got["init"], ref["params"] - ref["init"]
            assert cos(du[live], dr[live], dim=0) > 0.99, (mode, r, cos(du[live], dr[live], dim=0).item())
            assert cos(du, dr, dim=0) > 0.9, (mode, r)
        if backend == "nccl":
            assert res[r]["overlap"]["reduced_in_backward"] == [True, True, False]     # both encoder stacks' buckets
        assert torch.equal(res[r]["overlap"]["params"], res[r]["plain"]["params"])  # same arithmetic, other schedule
        assert torch.equal(res[r]["plain"]["params"], res[0]["plain"]["params"])    # ranks hold identical weights
