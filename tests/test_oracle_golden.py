"""CPU: the oracle (oracle/*.py) against the committed golden vectors generated from the reference
(tools/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path
with the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_ref, encoder_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(G, name))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    return z, sd


def t(z, k):
    return torch.from_numpy(z[k])


def test_clip_c1_kat():
    z, sd = load("clip_c1.npz")
    o = clip_ref.rna_protein_clip_forward(sd, t(z, "rna"), t(z, "protein"))
    logits = o["logits_per_rna_protein"]
    assert torch.allclose(logits, t(z, "logits"), atol=2e-5)
    assert torch.allclose(o["rna_embeds"], t(z, "rna_embeds"), atol=1e-6)
    # known answers recorded in SURVEY.md §8c / BASELINE.md §2
    assert abs(float(z["logit_scale_exp"]) - 14.28486) < 1e-4
    assert abs(clip_ref.ce_diag(logits).item() - 5.915865) < 1e-5
    assert abs(clip_ref.clip_loss_symmetric(logits).item() - 5.939782) < 1e-5
    assert abs(clip_ref.ce_diag(logits).item() - float(z["loss_one_sided"])) < 1e-6
    assert torch.allclose(o["rna_embeds"].norm(dim=-1), torch.ones(256), atol=1e-6)


def test_clip_c1_param_grads():
    z, sd = load("clip_c1.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = clip_ref.rna_protein_clip_forward(sd, t(z, "rna"), t(z, "protein"))
    clip_ref.clip_loss_symmetric(o["logits_per_rna_protein"]).backward()
    for k, v in sd.items():
        ref = t(z, "g:" + k)
        assert torch.allclose(v.grad, ref, rtol=1e-4, atol=1e-6), k


def test_clip_diffmap():
    z, sd = load("clip_diffmap.npz")
    o = clip_ref.rna_protein_clip_forward(sd, t(z, "diffmap"), t(z, "protein"), a="diffmap", b="protein",
                                          num_layers=(1, 3), eps=(1e-12, 1e-5))
    assert torch.allclose(o["logits_per_diffmap_protein"], t(z, "logits"), atol=2e-5)


def test_clip_opt_cache_loss():
    z, sd = load("clip_opt.npz")
    o = clip_ref.optimized_clip_forward(sd, t(z, "diffmap"), t(z, "protein"), t(z, "cache"))
    assert torch.allclose(o["logits_per_diffmap_protein"], t(z, "logits"), atol=2e-5)
    assert torch.allclose(o["logits_per_diffmap_cache"], t(z, "logits_cache"), atol=2e-5)
    assert abs(clip_ref.optimized_clip_loss(o).item() - float(z["loss"])) < 1e-5


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_transformer_layer(act):
    z, sd = load(f"tlayer_{act}.npz")
    x = t(z, "x").requires_grad_(True)
    valid = t(z, "valid")
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = encoder_ref.post_ln_layer(x, sdg, "l", 8, valid, act, 1e-12)
    m = valid[..., None]
    assert torch.allclose(y * m, t(z, "y") * m, atol=2e-5)
    (y * t(z, "dy")).sum().backward()
    assert torch.allclose(x.grad, t(z, "dx"), rtol=1e-4, atol=2e-5)
    for k, v in sdg.items():
        assert torch.allclose(v.grad, t(z, "g:" + k), rtol=1e-3, atol=5e-5), k


def test_notebook_model_batch_axis_attention():
    """The notebook's (B,L,D)-into-batch_first=False quirk reproduced by the oracle."""
    z, sd = load("notebook_model.npz")

    def enc(x, prefix):
        valid = ~torch.isnan(x).any(-1)
        xt = torch.nan_to_num(x, 0.0).transpose(0, 1)
        y = encoder_ref.post_ln_encoder(xt, sd, prefix, 3, 8, valid.transpose(0, 1), "relu", 1e-5, 1e-5)
        return y.transpose(0, 1)[:, 0]
    a = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(t(z, "rna"), "rna_encoder"), sd, "rna_projection"))
    b = clip_ref.l2_normalize(clip_ref.optimized_projection_head(enc(t(z, "rbp"), "rbp_encoder"), sd, "rbp_projection"))
    assert torch.allclose(a, t(z, "rna_embed"), atol=2e-5)
    assert torch.allclose(b, t(z, "rbp_embed"), atol=2e-5)
    loss = clip_ref.clip_loss_symmetric((a @ b.t()) * sd["logit_scale"].exp())
    assert abs(loss.item() - float(z["loss"])) < 2e-5


def test_esm_tiny_forward_and_grads():
    z, sd = load("esm_tiny.npz")
    ids, am = t(z, "ids"), t(z, "attention_mask")
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    y = encoder_ref.esm_encoder(ids, am, sdg, 2, 4, 1e-5)
    m = am[..., None].float()
    assert torch.allclose(y * m, t(z, "last_hidden_state") * m, atol=3e-5)
    zg = np.load(os.path.join(G, "esm_tiny_grads.npz"))
    (y * torch.from_numpy(zg["dy"])).sum().backward()
    checked = 0
    for k, v in sdg.items():
        if not v.is_floating_point() or ("g:" + k) not in zg.files or zg["g:" + k].size == 0:
            continue
        assert torch.allclose(v.grad, torch.from_numpy(zg["g:" + k]), rtol=2e-3, atol=2e-4), k
        checked += 1
    assert checked > 30


def test_analytic_properties():
    """Loss at random init ~ ln B; symmetric loss invariant under swapping modalities."""
    g = torch.Generator().manual_seed(0)
    a = clip_ref.l2_normalize(torch.randn(64, 32, generator=g))
    b = clip_ref.l2_normalize(torch.randn(64, 32, generator=g))
    s = (a @ b.t()) * 14.2849
    assert abs(clip_ref.clip_loss_symmetric(s).item() - clip_ref.clip_loss_symmetric(s.t()).item()) < 1e-6
    assert abs(clip_ref.clip_loss_symmetric(s * 0).item() - np.log(64)) < 1e-6


def test_icnn_training_branch():
    """Oracle's train-mode restatement (rescale under no_grad = constant, row-norm clip, cost) incl. the gradients
    through T vs the reference's double backward (golden), both fixture cases."""
    from oracle import icnn_ref
    zf = np.load(os.path.join(G, "icnn_train.npz"))
    for case in ("A", "B"):
        pre = case + ":"
        sd = {"t." + k[len(pre) + 2:]: torch.from_numpy(zf[k]).clone().requires_grad_(True) for k in zf.files
              if k.startswith(pre + "w:")}
        tr, cost, w2, sp = icnn_ref.single_cell_transport_train(torch.from_numpy(zf[pre + "source"]),
                                                                torch.from_numpy(zf[pre + "target"]), sd, "t", 3)
        assert torch.allclose(tr.detach(), torch.from_numpy(zf[pre + "transported"]), atol=2e-5)
        assert abs(cost.item() - float(zf[pre + "cost"])) < 2e-5
        cost.backward()
        for k in zf.files:
            if k.startswith(pre + "g:"):
                g = sd["t." + k[len(pre) + 2:]].grad
                g = torch.zeros_like(torch.from_numpy(zf[k])) if g is None else g
                assert torch.allclose(g, torch.from_numpy(zf[k]), atol=5e-5), k


def test_icnn_hessian():
    """oracle hessian (2_icnn_core.py:213-241) vs the reference's, eval and train mode (tools/make_golden.py)."""
    from oracle import icnn_ref
    zf = np.load(os.path.join(G, "icnn_hessian.npz"))
    sd = {"n." + k[2:]: torch.from_numpy(zf[k]) for k in zf.files if k.startswith("w:")}
    x = torch.from_numpy(zf["x"])
    for mode in ("eval", "train"):
        h = icnn_ref.icnn_hessian(x, sd, "n", 3, train=(mode == "train"))
        assert torch.allclose(h, torch.from_numpy(zf["hessian_" + mode]), rtol=1e-4, atol=2e-5), mode


def test_icnn_without_layer_norm():
    """use_layer_norm = False (ConvexLayer.norm = nn.Identity), softplus activation: potential and transport map."""
    from oracle import icnn_ref
    zf = np.load(os.path.join(G, "icnn_noln.npz"))
    sd = {"n." + k[2:]: torch.from_numpy(zf[k]) for k in zf.files if k.startswith("w:")}
    x = torch.from_numpy(zf["x"])
    kw = dict(activation="softplus", use_layer_norm=False)
    assert torch.allclose(icnn_ref.icnn_potential(x, sd, "n", 3, **kw), torch.from_numpy(zf["psi"]), atol=2e-5)
    assert torch.allclose(icnn_ref.icnn_gradient(x, sd, "n", 3, **kw), torch.from_numpy(zf["gradient"]), atol=2e-5)


def test_icnn_transport_maps():
    """triple_flow ICNN transport maps (eval): oracle vs the reference's autograd-of-autograd outputs."""
    from oracle import icnn_ref
    z, sd = load("icnn_transport.npz")
    for name, src in (("cell_to_pert", "cell"), ("cell_to_protein", "cell"), ("pert_to_protein", "pert")):
        o = icnn_ref.single_cell_transport(t(z, src), sd, name, 3)
        assert torch.allclose(o, t(z, "out_" + name), atol=2e-5), name
    s = encoder_ref._ln(t(z, "cell"), sd, "cell_to_pert.input_norm", 1e-5)
    psi = icnn_ref.icnn_potential(s, sd, "cell_to_pert.transport_net", 3)
    assert torch.allclose(psi, t(z, "psi_cell_to_pert"), atol=2e-5)


def test_trimodal_oracle_vs_reference_fixture():
    """oracle.model_ref.contrastive_model_forward vs the fixture generated from the reference's tri-modal classes
    (current/tf_clip_codes (1).ipynb cell 41, tools/make_golden.py gen_trimodal)."""
    from oracle import model_ref
    z = np.load(os.path.join(G, "trimodal_model.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    tt = lambda k: torch.from_numpy(z[k])
    o = model_ref.contrastive_model_forward(sd, tt("cell_state"), tt("connectivity"), tt("gene_esm"), tt("gene_values"),
                                            tt("protein_emb"))
    for k in ("cell_embed", "pert_embed", "protein_embed"):
        assert (o[k] - tt(k)).abs().max().item() < 3e-5, k
    for k in ("cell_pert_loss", "cell_protein_loss", "pert_protein_loss", "loss"):
        assert abs(o[k].item() - float(z[k])) < 3e-5, k


# ---------------------------------------------------------------------------------------------- round-3 fixtures
def test_clip_opt_b128():
    """old/clip_opt.py at its caller's batch (B = 128), cache holding two earlier batches."""
    z, sd = load("clip_opt_b128.npz")
    o = clip_ref.optimized_clip_forward(sd, t(z, "diffmap"), t(z, "protein"), t(z, "cache"))
    assert torch.allclose(o["logits_per_diffmap_protein"], t(z, "logits"), atol=3e-5)
    assert torch.allclose(o["logits_per_diffmap_cache"], t(z, "logits_cache"), atol=3e-5)
    assert abs(clip_ref.optimized_clip_loss(o).item() - float(z["loss"])) < 1e-5


def test_notebook_model_b32_loss_and_grads():
    """The notebook model at B = 32 with ragged NaN padding: embeddings, loss and EVERY parameter gradient of the
    reference's autograd."""
    from oracle import model_ref
    z, sd = load("notebook_model_b32.npz")
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    a, b, loss = model_ref.rnarbp_clip_forward(sdg, t(z, "rna"), t(z, "rbp"))
    assert torch.allclose(a, t(z, "rna_embed"), atol=2e-5) and torch.allclose(b, t(z, "rbp_embed"), atol=2e-5)
    assert abs(loss.item() - float(z["loss"])) < 2e-5
    loss.backward()
    for k, v in sdg.items():
        ref = t(z, "g:" + k)
        assert (v.grad - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k


def test_memory_queue_and_queue_loss():
    """tong/utils/data.py:154-184 + tong/utils/losses.py:4-19: five batches of 24 rows through a 64-row queue (wraps
    twice): queue contents, pointer and the tau = 0.1 one-sided loss against the queue BEFORE each enqueue."""
    z = np.load(os.path.join(G, "queue_loss.npz"))
    q, ptr = torch.zeros(64, 16), 0
    for step in range(5):
        x, y = t(z, f"x{step}"), t(z, f"y{step}")
        assert abs(clip_ref.contrastive_loss_queue(x, y, 0.1, q).item() - float(z[f"loss{step}"])) < 2e-6
        q, ptr = clip_ref.memory_queue_enqueue(q, ptr, torch.nn.functional.normalize(y, dim=-1))
        assert torch.equal(q, t(z, f"queue{step}")) and ptr == int(z[f"ptr{step}"])
    assert abs(clip_ref.contrastive_loss_queue(x, y, 0.1, None).item() - float(z["loss_noqueue"])) < 2e-6


def test_esm_integration_end_to_end():
    """triple_flow/3_esm_integration.py:90-135 run by the reference itself (tools/make_golden.py gen_esm_integration):
    tokenizer (truncation at max_sequence_length, <unk> runs, in-text special tokens), frozen ESM-2, both projections."""
    from oracle import esm_integration_ref as eref
    z, sd = load("esm_integration.npz")
    seqs = [str(s) for s in z["sequences"]]
    kw = dict(esm_layers=2, esm_heads=4, max_sequence_length=24)
    p, ids, mask = eref.get_embeddings(seqs, sd, protein=True, **kw)
    g, _, _ = eref.get_embeddings(seqs, sd, protein=False, **kw)
    assert torch.equal(ids, t(z, "input_ids")) and torch.equal(mask, t(z, "attention_mask"))
    assert ids.shape[1] == 24 and ids[0, -1].item() == 2                  # the long sequence was cut to 22 residues
    m = mask[..., None].float()
    assert torch.allclose(p * m, t(z, "protein_embeddings") * m, atol=5e-5)
    assert torch.allclose(g, t(z, "gene_embeddings"), atol=5e-5)
    ei, em = eref.tokenize([str(s) for s in z["edge_sequences"]], 16)
    assert torch.equal(ei, t(z, "edge_input_ids")) and torch.equal(em, t(z, "edge_attention_mask"))
