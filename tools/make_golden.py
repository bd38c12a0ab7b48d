#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run only in the build container (needs /root/reference and the third-party `transformers` package the
reference calls for ESM-2).  Nothing from the reference is copied: the fixtures are data — seeded weights,
inputs and the reference's outputs — stored as .npz.  Import recipe: SURVEY.md Appendix B.

    python tools/make_golden.py            # writes tests/golden/*.npz and checks the oracle against each
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)

from oracle import clip_ref, encoder_ref  # noqa: E402


def sd_np(module_or_sd, prefix="w:"):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def sub(h, n=2, eps=1e-12):
    return NS(hidden_size=h, num_hidden_layers=n, layer_norm_eps=eps)


def check(name, a, b, tol=1e-5):
    err = (a - b).abs().max().item()
    print(f"  oracle vs reference [{name}]: max abs err {err:.3e}")
    assert err <= tol, (name, err)


# ------------------------------------------------------------------------------------------------ old/clip.py
def gen_clip_c1():
    sys.path[:0] = [REF + "/run1", REF + "/old"]
    import clip as refclip
    cfg = NS(rna_config=sub(128), protein_config=sub(128), diffmap_config=sub(128), projection_dim=128,
             logit_scale_init_value=2.6592, cache_size=8192)
    torch.manual_seed(0)
    m = refclip.RNAProteinCLIPModule(cfg).eval()
    g = torch.Generator().manual_seed(1234)
    a, b = torch.randn(256, 128, generator=g), torch.randn(256, 128, generator=g)
    out = m(a, b)
    logits = out["logits_per_rna_protein"]
    lab = torch.arange(256)
    l1 = F.cross_entropy(logits, lab)
    l2 = 0.5 * (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab))
    grads = torch.autograd.grad(l2, list(m.parameters()))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    o = clip_ref.rna_protein_clip_forward(sd, a, b)
    check("C1 logits", o["logits_per_rna_protein"], logits.detach(), 2e-5)
    check("C1 sym loss", clip_ref.clip_loss_symmetric(o["logits_per_rna_protein"]), l2.detach(), 1e-6)
    arrays = sd_np(m)
    arrays.update({"g:" + n: gr.numpy() for (n, _), gr in zip(m.named_parameters(), grads)})
    save("clip_c1.npz", rna=a.numpy(), protein=b.numpy(), logits=logits.detach().numpy(),
         rna_embeds=out["rna_embeds"].detach().numpy(), protein_embeds=out["protein_embeds"].detach().numpy(),
         loss_one_sided=l1.item(), loss_symmetric=l2.item(), logit_scale_exp=m.logit_scale.exp().item(), **arrays)
    # DiffMap pairing (old/clip.py:75-110), smaller
    torch.manual_seed(1)
    cfg2 = NS(rna_config=sub(64), protein_config=sub(96, 3, 1e-5), diffmap_config=sub(64, 1), projection_dim=32,
              logit_scale_init_value=2.6592)
    m2 = refclip.DiffMapProteinCLIPModule(cfg2).eval()
    a2, b2 = torch.randn(40, 64, generator=g), torch.randn(40, 96, generator=g)
    o2 = m2(a2, b2)
    sd2 = {k: v.detach() for k, v in m2.state_dict().items()}
    oo = clip_ref.rna_protein_clip_forward(sd2, a2, b2, a="diffmap", b="protein", num_layers=(1, 3), eps=(1e-12, 1e-5))
    check("diffmap logits", oo["logits_per_diffmap_protein"], o2["logits_per_diffmap_protein"].detach(), 2e-5)
    save("clip_diffmap.npz", diffmap=a2.numpy(), protein=b2.numpy(),
         logits=o2["logits_per_diffmap_protein"].detach().numpy(), **sd_np(m2))
    return refclip


def gen_clip_opt(refclip):
    import clip_opt as refopt
    refopt.CLIPEncoder = refclip.CLIPEncoder            # App. A-4
    cfg = NS(diffmap_config=sub(48), protein_config=sub(96), projection_dim=32, cache_size=256)
    torch.manual_seed(0)
    m = refopt.OptimizedCLIPModule(cfg).eval()
    g = torch.Generator().manual_seed(7)
    # two batches so the cache holds an earlier batch too
    d0, p0 = torch.randn(32, 48, generator=g), torch.randn(32, 96, generator=g)
    d1, p1 = torch.randn(32, 48, generator=g), torch.randn(32, 96, generator=g)
    with torch.no_grad():
        m(d0, p0, gather_distributed=False)
        out = m(d1, p1, gather_distributed=False)
        loss = refopt.optimized_clip_loss(out)
    cache = m.protein_embedding_cache[: m.cache_ptr].clone()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    o = clip_ref.optimized_clip_forward(sd, d1, p1, cache)
    check("opt logits", o["logits_per_diffmap_protein"], out["logits_per_diffmap_protein"], 2e-5)
    check("opt cache logits", o["logits_per_diffmap_cache"], out["logits_per_diffmap_cache"], 2e-5)
    check("opt loss", clip_ref.optimized_clip_loss(o), loss, 1e-6)
    save("clip_opt.npz", diffmap=d1.numpy(), protein=p1.numpy(), cache=cache.numpy(), cache_ptr=m.cache_ptr,
         logits=out["logits_per_diffmap_protein"].numpy(), logits_cache=out["logits_per_diffmap_cache"].numpy(),
         loss=loss.item(), **sd_np(m))


# ------------------------------------------------------------------------------------------------ notebook
def gen_notebook():
    nb = json.load(open(REF + "/current/rna_clip_codes.ipynb"))
    ns = {"torch": torch, "nn": nn, "F": F, "np": np}
    for i in (24, 28):
        exec("".join(nb["cells"][i]["source"]), ns)          # dataset/mask helpers, model classes
    torch.manual_seed(0)
    model = ns["RNARBPCLIPModel"](rna_dim=24, rbp_dim=64, projection_dim=32).eval()
    g = torch.Generator().manual_seed(3)
    B = 8
    rna = torch.randn(B, 6, 24, generator=g)
    rbp = torch.randn(B, 20, 64, generator=g)
    for i, (lr, lp) in enumerate([(6, 20), (4, 20), (6, 11), (5, 17), (6, 20), (3, 9), (6, 14), (6, 20)]):
        rna[i, lr:] = float("nan")
        rbp[i, lp:] = float("nan")
    with torch.no_grad():
        ea, eb, loss = model(rna, rbp)
    sd = {k: v.detach() for k, v in model.state_dict().items()}

    # oracle: the notebook feeds (B, L, D) to batch_first=False layers, so attention runs over the batch
    # axis per position (App. A-8): equivalent to per-"sequence" attention on the transposed tensor.
    def enc(x, prefix):
        valid = ~torch.isnan(x).any(-1)                      # [B, L]
        xt = torch.nan_to_num(x, 0.0).transpose(0, 1)        # [L, B, D]: "batch" = position, "sequence" = sample
        kv = valid.transpose(0, 1)                           # [L, B] keys = samples
        y = encoder_ref.post_ln_encoder(xt, sd, prefix, 3, 8, kv, "relu", 1e-5, 1e-5)
        return y.transpose(0, 1)[:, 0]                       # position 0
    oa = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(rna, "rna_encoder"), sd, "rna_projection"))
    ob = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(rbp, "rbp_encoder"), sd, "rbp_projection"))
    ol = clip_ref.clip_loss_symmetric((oa @ ob.t()) * sd["logit_scale"].exp())
    check("notebook rna embed", oa, ea, 2e-5)
    check("notebook rbp embed", ob, eb, 2e-5)
    check("notebook loss", ol, loss, 2e-5)
    save("notebook_model.npz", rna=rna.numpy(), rbp=rbp.numpy(), rna_embed=ea.numpy(), rbp_embed=eb.numpy(),
         loss=loss.item(), **sd_np(model))


def gen_trimodal():
    """Tri-modal ContrastiveModel, current/tf_clip_codes (1).ipynb cell 41 (:13026-13176).  The cell was never executed
    upstream and its forward cannot run on the documented 2-D inputs (`cell_enc[:, 0]` of a [B, E] tensor is a [B]
    vector: defect A-19), so the fixture is built from the reference's OWN sub-modules, each called as the forward calls
    it, with `[:, 0]` applied to the 3-D perturbation output only, and the reference's own loss expressions."""
    from oracle import model_ref
    nb = json.load(open(REF + "/current/tf_clip_codes (1).ipynb"))
    cell = next(c for c in nb["cells"] if "class ContrastiveModel" in "".join(c["source"]))
    src = "".join(cell["source"])
    src = src[:src.index("# Helper function to get ESM embeddings")]          # model classes only
    ns = {}
    exec(src, ns)
    torch.manual_seed(0)
    gene_dim, protein_dim, P, esm_dim, B, G = 21, 64, 64, 40, 32, 5      # B = 32: the notebooks' batch size
    m = ns["ContrastiveModel"](gene_dim, protein_dim, projection_dim=P, esm_dim=esm_dim).eval()
    with torch.no_grad():                                   # move layer_scale off its 1e-4 init so the MLP branch counts
        for h in (m.cell_projection, m.pert_projection, m.protein_projection):
            h.layer_scale.fill_(0.5)
    g = torch.Generator().manual_seed(11)
    cell_state = torch.randn(B, gene_dim + 1, generator=g)
    conn = (torch.rand(B, B, generator=g) > 0.6).float()
    conn[3] = 0; conn[9] = 0                                 # two isolated cells: masked as attention keys
    gene_esm = torch.randn(B, G, esm_dim, generator=g)
    gene_values = torch.randn(B, G, generator=g)
    protein_emb = torch.randn(B, protein_dim, generator=g)
    cell_enc = m.cell_encoder(cell_state, conn)          # [B, P]
    pert_enc = m.pert_encoder(gene_esm, gene_values)     # [B, G, P]
    prot_enc = m.protein_encoder(protein_emb)            # [B, protein_dim]
    ce = F.normalize(m.cell_projection(cell_enc), dim=-1)
    pe = F.normalize(m.pert_projection(pert_enc[:, 0]), dim=-1)
    pr = F.normalize(m.protein_projection(prot_enc), dim=-1)
    ls = m.logit_scale.exp()
    lab = torch.arange(B)
    sym = lambda a, b: (F.cross_entropy((a @ b.t()) * ls, lab) + F.cross_entropy(((a @ b.t()) * ls).t(), lab)) / 2
    cp, cpr, ppr = sym(ce, pe), sym(ce, pr), sym(pe, pr)
    grads = torch.autograd.grad(cp + cpr + ppr, list(m.parameters()))        # the reference's own gradients
    ce, pe, pr, cp, cpr, ppr = (v.detach() for v in (ce, pe, pr, cp, cpr, ppr))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = model_ref.contrastive_model_forward(sdg, cell_state, conn, gene_esm, gene_values, protein_emb)
    check("trimodal cell embed", o["cell_embed"].detach(), ce, 3e-5)
    check("trimodal pert embed", o["pert_embed"].detach(), pe, 3e-5)
    check("trimodal protein embed", o["protein_embed"].detach(), pr, 3e-5)
    check("trimodal loss", o["loss"].detach(), cp + cpr + ppr, 3e-5)
    o["loss"].backward()
    for (n, _), gr in zip(m.named_parameters(), grads):
        err = (sdg[n].grad - gr).abs().max().item()
        assert err <= 1e-4 * max(1.0, gr.abs().max().item()), (n, err)
    print("  oracle gradients == reference gradients for every parameter")
    arrays = {"g:" + n: gr.numpy() for (n, _), gr in zip(m.named_parameters(), grads)}
    save("trimodal_model.npz", cell_state=cell_state.numpy(), connectivity=conn.numpy(), gene_esm=gene_esm.numpy(),
         gene_values=gene_values.numpy(), protein_emb=protein_emb.numpy(), cell_embed=ce.numpy(), pert_embed=pe.numpy(),
         protein_embed=pr.numpy(), cell_pert_loss=cp.item(), cell_protein_loss=cpr.item(), pert_protein_loss=ppr.item(),
         loss=(cp + cpr + ppr).item(), **arrays, **sd_np(m))


# ------------------------------------------------------------------------------------------------ torch layer
def gen_tlayer():
    torch.manual_seed(0)
    E, H, FF, B, L = 64, 8, 128, 4, 24
    for act in ("relu", "gelu"):
        layer = nn.TransformerEncoderLayer(E, H, FF, dropout=0.0, activation=act, layer_norm_eps=1e-12).eval()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, L, E, generator=g, requires_grad=True)
        lens = torch.tensor([24, 17, 9, 24])
        valid = torch.arange(L)[None] < lens[:, None]
        y = layer(x.transpose(0, 1), src_key_padding_mask=~valid).transpose(0, 1)    # conventional layout
        dy = torch.randn(B, L, E, generator=g) * valid[..., None]
        grads = torch.autograd.grad(y, [x] + list(layer.parameters()), dy)
        sd = {"l." + k: v.detach() for k, v in layer.state_dict().items()}
        yo = encoder_ref.post_ln_layer(x.detach(), sd, "l", H, valid, act, 1e-12)
        check(f"tlayer {act}", yo * valid[..., None], y.detach() * valid[..., None], 2e-5)
        arrays = {"w:" + k: v.numpy() for k, v in sd.items()}
        arrays.update({"g:l." + n: gr.numpy() for (n, _), gr in zip(layer.named_parameters(), grads[1:])})
        save(f"tlayer_{act}.npz", x=x.detach().numpy(), valid=valid.numpy(), y=y.detach().numpy(), dy=dy.numpy(),
             dx=grads[0].numpy(), **arrays)


# ------------------------------------------------------------------------------------------------ ESM-2
def gen_esm():
    from transformers import EsmConfig, EsmModel
    cfg = EsmConfig(vocab_size=33, hidden_size=96, num_hidden_layers=2, num_attention_heads=4, intermediate_size=384,
                    position_embedding_type="rotary", token_dropout=True, emb_layer_norm_before=False, pad_token_id=1,
                    mask_token_id=32, layer_norm_eps=1e-5, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    m = EsmModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(11)
    B, L = 4, 40
    ids = torch.randint(4, 24, (B, L), generator=g)
    ids[:, 0] = 0
    lens = [40, 33, 21, 40]
    am = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens):
        am[i, :n] = 1
        ids[i, n - 1] = 2
        ids[i, n:] = 1
    ids[1, 5] = 32                                           # one <mask> token: exercises the token-dropout rescale
    with torch.no_grad():
        y = m(input_ids=ids, attention_mask=am).last_hidden_state
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    yo = encoder_ref.esm_encoder(ids, am, sd, 2, 4, 1e-5)
    check("esm tiny", yo * am[..., None], y * am[..., None], 3e-5)
    save("esm_tiny.npz", ids=ids.numpy(), attention_mask=am.numpy(), last_hidden_state=y.numpy(), **sd_np(sd))
    # gradient fixture: d(sum(y * dy)) / d params and loss through mean pooling
    m.train(False)
    for p in m.parameters():
        p.requires_grad_(True)
    dy = torch.randn(B, L, 96, generator=g) * am[..., None]
    y2 = m(input_ids=ids, attention_mask=am).last_hidden_state
    grads = torch.autograd.grad((y2 * dy).sum(), [p for p in m.parameters()], allow_unused=True)
    arrays = {"g:" + n: (gr.numpy() if gr is not None else np.zeros(0)) for (n, _), gr in zip(m.named_parameters(), grads)}
    save("esm_tiny_grads.npz", dy=dy.numpy(), **arrays)


# ------------------------------------------------------------------------------------------------ ICNN transport
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def gen_icnn():
    from oracle import icnn_ref
    _load("config", REF + "/triple_flow/1_config.py")
    core = _load("core", REF + "/triple_flow/2_icnn_core.py")
    tm = _load("transport_maps", REF + "/triple_flow/4_transport_maps.py")
    gen_icnn_hessian(core, icnn_ref)
    torch.manual_seed(0)
    model = tm.create_transport_system(64, 64, 64, hidden_dims=[64, 64, 32]).eval()
    with torch.no_grad():                                     # make the positive path matter
        for m in model.modules():
            if hasattr(m, "pos_weights"):
                m.pos_weights.normal_(0, 0.5)
                m.scale.fill_(0.3)
    g = torch.Generator().manual_seed(21)
    cell, pert, prot = (torch.randn(24, 64, generator=g) for _ in range(3))
    out = model(cell, pert, prot)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    for name, src in (("cell_to_pert", cell), ("cell_to_protein", cell), ("pert_to_protein", pert)):
        o = icnn_ref.single_cell_transport(src, sd, name, 3)
        check(f"icnn {name}", o, out[name].detach(), 2e-5)
    psi = model.cell_to_pert.transport_net(model.cell_to_pert.input_norm(cell))[0].detach()
    gen_icnn_train(tm, icnn_ref)
    save("icnn_transport.npz", cell=cell.numpy(), pert=pert.numpy(), protein=prot.numpy(), psi_cell_to_pert=psi.numpy(),
         **{"out_" + k: v.detach().numpy() for k, v in out.items()}, **sd_np(model))


def gen_icnn_train(tm, icnn_ref):
    """Training branch of ONE transport map (the reference's 3-modality training call is broken):
    cost, metrics and d cost / d parameters through T = dPsi/dx (double backward).  Two cases: A = no train-time
    rescale (every parameter receives a gradient), B = the rescale fires in the last layer (the reference then treats
    the rescaled z contribution as a constant, so only the last layer's x-path parameters get gradients).  Some, not
    all, rows of T exceed the clip norm in both."""
    cfg = tm.ICNNConfig(input_dim=64, hidden_dims=[64, 64, 32])
    arrays = {}
    for case, (s1, s2, fmul) in (("A", (0.05, 0.05, 2.0)), ("B", (0.05, 0.8, 2.2))):
        torch.manual_seed(3)
        m = tm.SingleCellTransport(64, 64, cfg).train()
        with torch.no_grad():
            for mod in m.modules():
                if hasattr(mod, "pos_weights"):
                    mod.pos_weights.normal_(0, 0.5)
            m.transport_net.layers[1].scale.fill_(s1)
            m.transport_net.layers[2].scale.fill_(s2)
            m.transport_net.final.weight.mul_(fmul)
        g = torch.Generator().manual_seed(22)
        src, tgt = torch.randn(24, 64, generator=g), torch.randn(24, 64, generator=g)
        out = m(src.clone(), tgt.clone())
        out.cost.backward()
        sd = {"t." + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        t, cost, w2, sp = icnn_ref.single_cell_transport_train(src, tgt, sd, "t", 3)
        check(f"icnn train {case} transported", t.detach(), out.transported.detach(), 2e-5)
        check(f"icnn train {case} cost", cost.detach(), out.cost.detach(), 2e-5)
        cost.backward()
        for k, v in m.named_parameters():
            if v.grad is not None:
                check(f"icnn train {case} grad " + k, sd["t." + k].grad, v.grad, 5e-5)
        with torch.no_grad():
            s = m.input_norm(src)
        sreq = s.clone().requires_grad_(True)
        raw = torch.autograd.grad(m.transport_net.forward(sreq)[0].sum(), sreq)[0]
        n_clip = int((raw.norm(dim=-1) > cfg.gradient_clip).sum())
        with_grad = sorted(k for k, v in m.named_parameters() if v.grad is not None and v.grad.abs().max() > 0)
        print(f"  train fixture {case}: {n_clip}/24 rows of T norm-clipped; {len(with_grad)} parameters with gradients")
        arrays.update({f"{case}:source": src.numpy(), f"{case}:target": tgt.numpy(),
                       f"{case}:transported": out.transported.detach().numpy(), f"{case}:cost": out.cost.detach().numpy(),
                       f"{case}:w2": np.float32(out.metrics["w2_cost"]),
                       f"{case}:sparsity": np.float32(out.metrics["sparsity_cost"]), f"{case}:rows_clipped": np.int32(n_clip)})
        arrays.update({f"{case}:g:" + k: (v.grad.detach().numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32))
                       for k, v in m.named_parameters()})
        arrays.update({f"{case}:w:" + k: v.detach().numpy() for k, v in m.state_dict().items()})
    save("icnn_train.npz", **arrays)


def gen_icnn_hessian(core, icnn_ref):
    """SingleCellICNN.hessian (2_icnn_core.py:213-241), eval and train mode, on a small potential."""
    cfgm = sys.modules["config"]
    cfg = cfgm.ICNNConfig(input_dim=16, hidden_dims=[16, 16, 8])
    torch.manual_seed(5)
    m = core.SingleCellICNN(cfg)
    with torch.no_grad():
        for mod in m.modules():
            if hasattr(mod, "pos_weights"):
                mod.pos_weights.normal_(0, 0.5)
                mod.scale.fill_(0.3)
        m.final.weight.mul_(3.0)                             # some rows of T above the clip norm in train mode
    g = torch.Generator().manual_seed(23)
    x = torch.randn(6, 16, generator=g)
    sd = {"n." + k: v.detach() for k, v in m.state_dict().items()}
    arrays = {"x": x.numpy()}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        h = m.hessian(x.clone()).detach()
        o = icnn_ref.icnn_hessian(x, sd, "n", 3, train=(mode == "train"), hessian_reg=cfg.hessian_reg)
        check(f"icnn hessian {mode}", o, h, 2e-5)
        arrays["hessian_" + mode] = h.numpy()
        if mode == "eval":
            ev = torch.linalg.eigvalsh(0.5 * (h + h.transpose(-1, -2)))
            print(f"  hessian fixture: min eigenvalue of sym(H) over the batch {ev.min().item():.3e}")
    arrays.update({"w:" + k: v.detach().numpy() for k, v in m.state_dict().items()})
    save("icnn_hessian.npz", **arrays)
    # use_layer_norm = False (ConvexLayer.norm = nn.Identity, :72), softplus activation: potential and transport map
    cfg2 = cfgm.ICNNConfig(input_dim=16, hidden_dims=[16, 16, 8], use_layer_norm=False, activation="softplus")
    torch.manual_seed(6)
    m2 = core.SingleCellICNN(cfg2).eval()
    with torch.no_grad():
        for mod in m2.modules():
            if hasattr(mod, "pos_weights"):
                mod.pos_weights.normal_(0, 0.5)
                mod.scale.fill_(0.3)
    psi = m2(x.clone())[0].detach()
    tmap = m2.gradient(x.clone()).detach()
    sd2 = {"n." + k: v.detach() for k, v in m2.state_dict().items()}
    check("icnn no-LN potential", icnn_ref.icnn_potential(x, sd2, "n", 3, activation="softplus", use_layer_norm=False), psi, 2e-5)
    check("icnn no-LN gradient", icnn_ref.icnn_gradient(x, sd2, "n", 3, activation="softplus", use_layer_norm=False), tmap, 2e-5)
    save("icnn_noln.npz", x=x.numpy(), psi=psi.numpy(), gradient=tmap.numpy(),
         **{"w:" + k: v.detach().numpy() for k, v in m2.state_dict().items()})


def gen_esm_projections():
    esmi = _load("esm_integration", REF + "/triple_flow/3_esm_integration.py")   # projection classes only
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(3, 10, 64, generator=g)
    out = {}
    for name, cls in (("protein", esmi.ProteinProjection), ("gene", esmi.GeneProjection)):
        m = cls(esm_dim=64, output_dim=32).eval()
        with torch.no_grad():
            y = m(x)
        out[name] = (m, y)
    arrays = {"x": x.numpy()}
    for name, (m, y) in out.items():
        arrays["y_" + name] = y.numpy()
        arrays.update(sd_np({name + "." + k: v for k, v in m.state_dict().items()}))
    save("esm_projections.npz", **arrays)



# ------------------------------------------------------------------------------------------------ larger-batch variants
def gen_clip_opt_b128(refclip):
    """old/clip_opt.py at the batch of its caller (run1/full.py:189-198: B = 128), three batches so the cache holds two
    earlier ones: the loss bar of the GPU test is the north-star 1e-3 at this batch."""
    import clip_opt as refopt
    refopt.CLIPEncoder = refclip.CLIPEncoder
    cfg = NS(diffmap_config=sub(48), protein_config=sub(96), projection_dim=32, cache_size=512)
    torch.manual_seed(1)
    m = refopt.OptimizedCLIPModule(cfg).eval()
    g = torch.Generator().manual_seed(17)
    B = 128
    batches = [(torch.randn(B, 48, generator=g), torch.randn(B, 96, generator=g)) for _ in range(3)]
    with torch.no_grad():
        for d, pr in batches[:2]:
            m(d, pr, gather_distributed=False)
        ptr_before = m.cache_ptr
        out = m(*batches[2], gather_distributed=False)
        loss = refopt.optimized_clip_loss(out)
    cache = m.protein_embedding_cache[: m.cache_ptr].clone()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    o = clip_ref.optimized_clip_forward(sd, batches[2][0], batches[2][1], cache)
    check("opt b128 logits", o["logits_per_diffmap_protein"], out["logits_per_diffmap_protein"], 3e-5)
    check("opt b128 loss", clip_ref.optimized_clip_loss(o), loss, 2e-6)
    save("clip_opt_b128.npz", diffmap=batches[2][0].numpy(), protein=batches[2][1].numpy(), cache=cache.numpy(),
         cache_ptr=m.cache_ptr, cache_ptr_before=ptr_before, logits=out["logits_per_diffmap_protein"].numpy(),
         logits_cache=out["logits_per_diffmap_cache"].numpy(), loss=loss.item(), **sd_np(m))


def gen_notebook_b32():
    """The notebook model (current/rna_clip_codes.ipynb cells 24 + 28) at the notebook's batch size (B = 32, :1987) and
    reduced widths (the 71.6 M-parameter original does not fit a fixture; tests/test_gpu_configs.py runs it at its own
    widths against the oracle this function checks)."""
    from oracle import model_ref
    nb = json.load(open(REF + "/current/rna_clip_codes.ipynb"))
    ns = {"torch": torch, "nn": nn, "F": F, "np": np}
    for i in (24, 28):
        exec("".join(nb["cells"][i]["source"]), ns)
    torch.manual_seed(2)
    model = ns["RNARBPCLIPModel"](rna_dim=40, rbp_dim=128, projection_dim=64).eval()
    g = torch.Generator().manual_seed(5)
    B, Lr, Lp = 32, 12, 40
    rna = torch.randn(B, Lr, 40, generator=g)
    rbp = torch.randn(B, Lp, 128, generator=g)
    lr = torch.randint(3, Lr + 1, (B,), generator=g)
    lp = torch.randint(9, Lp + 1, (B,), generator=g)
    lr[0], lp[0] = Lr, Lp
    for i in range(B):
        rna[i, lr[i]:] = float("nan")
        rbp[i, lp[i]:] = float("nan")
    rna.requires_grad_(False)
    ea, eb, loss = model(rna, rbp)
    grads = torch.autograd.grad(loss, list(model.parameters()))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oa, ob, ol = model_ref.rnarbp_clip_forward(sdg, rna, rbp)
    check("notebook b32 rna embed", oa.detach(), ea.detach(), 2e-5)
    check("notebook b32 rbp embed", ob.detach(), eb.detach(), 2e-5)
    check("notebook b32 loss", ol.detach(), loss.detach(), 2e-5)
    ol.backward()
    for (n, _), gr in zip(model.named_parameters(), grads):
        err = (sdg[n].grad - gr).abs().max().item()
        assert err <= 1e-4 * max(1.0, gr.abs().max().item()), (n, err)
    print("  oracle gradients == reference gradients for every parameter")
    arrays = {"g:" + n: gr.numpy() for (n, _), gr in zip(model.named_parameters(), grads)}
    save("notebook_model_b32.npz", rna=rna.numpy(), rbp=rbp.numpy(), rna_embed=ea.detach().numpy(),
         rbp_embed=eb.detach().numpy(), loss=loss.item(), **arrays, **sd_np(model))


# ------------------------------------------------------------------------------------------------ tong/ queue + loss
def gen_queue_loss():
    """tong/utils/data.py:154-184 `MemoryQueue` (true wrap-around FIFO) and tong/utils/losses.py:4-19
    `contrastive_loss` (one-sided, tau = 0.1, queue rows appended to the keys).  data.py imports scanpy / anndata at
    module level (absent here), so the class is taken by source span, like the notebook cells; losses.py imports."""
    import ast
    src = open(REF + "/tong/utils/data.py").read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "MemoryQueue")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "tong/utils/data.py", "exec"), ns)
    losses = _load("tong_losses", REF + "/tong/utils/losses.py")
    size, dim, B = 64, 16, 24
    q = ns["MemoryQueue"](size, dim)
    g = torch.Generator().manual_seed(23)
    arrays, oq, optr = {}, torch.zeros(size, dim), 0
    for step in range(5):                                     # 5 x 24 rows into 64: wraps at step 2 and at step 4
        x, y = torch.randn(B, dim, generator=g), torch.randn(B, dim, generator=g)
        loss = losses.contrastive_loss(x, y, temperature=0.1, queue=q.queue)
        ol = clip_ref.contrastive_loss_queue(x, y, 0.1, oq)
        check(f"queue loss step {step}", ol, loss, 2e-6)
        yn = F.normalize(y, dim=-1)
        full = q.enqueue_dequeue(yn)
        oq, optr = clip_ref.memory_queue_enqueue(oq, optr, yn)
        check(f"queue state step {step}", oq, full, 0.0)
        assert optr == q.ptr, (optr, q.ptr)
        arrays.update({f"x{step}": x.numpy(), f"y{step}": y.numpy(), f"loss{step}": loss.item(),
                       f"queue{step}": full.clone().numpy(), f"ptr{step}": q.ptr})
    loss_noq = losses.contrastive_loss(x, y, temperature=0.1)
    check("queue loss without queue", clip_ref.contrastive_loss_queue(x, y, 0.1, None), loss_noq, 2e-6)
    save("queue_loss.npz", loss_noqueue=loss_noq.item(), **arrays)


# ------------------------------------------------------------------------------------------------ ESMIntegration e2e
def gen_esm_integration():
    """The reference's own `ESMIntegration.get_embeddings` (triple_flow/3_esm_integration.py:90-135) end to end:
    tokenizer call, frozen ESM-2, projection, cache.  Its two `from_pretrained(<name>)` calls (:77-80) are pointed at a
    locally constructed seeded `transformers.EsmModel` (explicit shape, as gen_esm) and a `transformers.EsmTokenizer`
    built from the 33-token alphabet written to a temp file: nothing is fetched."""
    import tempfile
    from transformers import EsmConfig, EsmModel, EsmTokenizer
    from oracle import esm_integration_ref as eref
    cfgm = _load("config", REF + "/triple_flow/1_config.py")
    esmi = _load("esm_integration", REF + "/triple_flow/3_esm_integration.py")
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "vocab.txt"), "w") as f:
        f.write("\n".join(eref.VOCAB) + "\n")
    hf_cfg = EsmConfig(vocab_size=33, hidden_size=96, num_hidden_layers=2, num_attention_heads=4, intermediate_size=384,
                       position_embedding_type="rotary", token_dropout=True, emb_layer_norm_before=False,
                       pad_token_id=1, mask_token_id=32, layer_norm_eps=1e-5, hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0)

    def local_model(name):
        torch.manual_seed(4)
        return EsmModel(hf_cfg, add_pooling_layer=False)
    esmi.AutoModel = NS(from_pretrained=local_model)
    esmi.AutoTokenizer = NS(from_pretrained=lambda name: EsmTokenizer(os.path.join(tmp, "vocab.txt")))
    cfg = cfgm.ESMConfig(model_name="esm2_t33_650M_UR50D", esm_dim=96, protein_dim=32, gene_dim=32,
                         max_sequence_length=24)
    torch.manual_seed(6)
    ref = esmi.ESMIntegration(cfg).eval()
    assert not any(p.requires_grad for p in ref.model.parameters())
    seqs = ["MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQ",          # longer than max_sequence_length - 2: truncated
            "ACDEFGHIKLMNPQRSTVWY", "MKV", "AXBZUO", "ACD<mask>EFxyzGH", "M K T A"]
    P = cfgm.BiologicalDataType.PROTEIN_SEQUENCE
    out_p = ref.get_embeddings(seqs, P)
    again = ref.get_embeddings(seqs, cfgm.BiologicalDataType.PERTURBATION)
    assert again is out_p, "the reference's cache is keyed on the sequences only"
    ref.cache.clear()
    out_g = ref.get_embeddings(seqs, cfgm.BiologicalDataType.PERTURBATION)
    tok = ref.tokenizer(seqs, padding=True, truncation=True, max_length=24, return_tensors="pt")
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    kw = dict(esm_layers=2, esm_heads=4, max_sequence_length=24)
    op, ids, mask = eref.get_embeddings(seqs, sd, protein=True, **kw)
    og, _, _ = eref.get_embeddings(seqs, sd, protein=False, **kw)
    assert torch.equal(ids, tok["input_ids"]) and torch.equal(mask, tok["attention_mask"])
    m = mask[..., None].float()
    check("esm integration protein", op * m, out_p.embeddings * m, 5e-5)
    check("esm integration gene (all keys attended, as the reference does)", og, out_g.embeddings, 5e-5)
    # tokenizer vectors on their own (EsmTokenizer edge cases)
    edge = ["ACDxyEF", "AC DE", "A  xy zz C", "A.-B", "A<null_1>C<pad>D<cls>E<eos>F<unk>G", "<mas k>A", " A ", "xyz", "",
            "MKT" * 20]
    et = ref.tokenizer(edge, padding=True, truncation=True, max_length=16, return_tensors="pt")
    ei, em = eref.tokenize(edge, 16)
    assert torch.equal(ei, et["input_ids"]) and torch.equal(em, et["attention_mask"])
    save("esm_integration.npz", sequences=np.array(seqs), input_ids=ids.numpy(), attention_mask=mask.numpy(),
         protein_embeddings=out_p.embeddings.numpy(), gene_embeddings=out_g.embeddings.numpy(),
         edge_sequences=np.array(edge), edge_input_ids=et["input_ids"].numpy(),
         edge_attention_mask=et["attention_mask"].numpy(), **sd_np(sd))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "proj":
        _load("config", REF + "/triple_flow/1_config.py")
        gen_esm_projections()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "icnn":
        gen_icnn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "trimodal":       # round 4: B = 32 + the reference's gradients
        gen_trimodal()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r03":            # the round-3 additions only
        sys.path[:0] = [REF + "/run1", REF + "/old"]
        import clip as _refclip
        gen_clip_opt_b128(_refclip)
        gen_notebook_b32()
        gen_queue_loss()
        gen_esm_integration()
        sys.exit(0)
    rc = gen_clip_c1()
    gen_clip_opt(rc)
    gen_notebook()
    gen_trimodal()
    gen_tlayer()
    gen_esm()
    gen_icnn()
    gen_esm_projections()
    gen_clip_opt_b128(rc)
    gen_notebook_b32()
    gen_queue_loss()
    gen_esm_integration()
    print("all golden fixtures written and the oracle agrees with the reference on each")
