"""CPU: HybridCLIPConfig keeps the reference's public surface (run1/configuration_hybrid_clip.py:93-218)."""
import pytest

import clip_dplm_amd as K


def _cfg(**kw):
    return K.HybridCLIPConfig(rna_config={"hidden_size": 64, "num_hidden_layers": 3}, protein_config={"hidden_size": 96},
                              diffmap_config={}, **kw)


def test_defaults_and_subconfigs():
    c = _cfg()
    assert (c.projection_dim, c.cache_size, c.max_position_embeddings, c.embedding_dim) == (512, 8192, 512, 768)
    assert abs(c.logit_scale_init_value - 2.6592) < 1e-9
    assert c.use_mean_pooling and c.use_hard_negatives and c.use_layer_scale
    assert c.rna_config.hidden_size == 64 and c.rna_config.num_hidden_layers == 3
    assert c.protein_config.hidden_size == 96 and c.diffmap_config.layer_norm_eps == 1e-12
    t = c.architectures["transformer"]
    assert (t.num_layers, t.hidden_size, t.attention_heads, t.intermediate_size, t.hidden_act) == (6, 768, 8, 2048, "gelu")
    assert (c.training.batch_size, c.training.learning_rate, c.training.gradient_clip) == (128, 3e-4, 1.0)


@pytest.mark.parametrize("missing", ["rna_config", "protein_config", "diffmap_config"])
def test_missing_subconfig_raises_like_the_reference(missing):
    kw = {"rna_config": {}, "protein_config": {}, "diffmap_config": {}}
    kw.pop(missing)
    with pytest.raises(ValueError, match=f"`{missing}` cannot be `None`"):
        K.HybridCLIPConfig(**kw)


def test_round_trip_and_experiment_configs():
    c = _cfg(projection_dim=128)
    d = c.to_dict()
    assert d["model_type"] == "hybrid-clip" and d["rna_config"]["hidden_size"] == 64
    assert set(d["architectures"]) == {"mlp", "transformer", "resnet"} and d["training"]["temperature"] == 0.07
    c2 = K.HybridCLIPConfig.from_dict(d)
    assert c2.projection_dim == 128 and c2.rna_config.hidden_size == 64 and c2.architectures["resnet"].num_layers == 4
    c3 = K.HybridCLIPConfig.from_configs(c.rna_config, c.protein_config, c.diffmap_config, projection_dim=32)
    assert c3.projection_dim == 32 and c3.protein_config.hidden_size == 96
    assert c.create_experiment_config("training_sweep", batch_size=64).training.batch_size == 64
    assert c.create_experiment_config("embedding_sweep", embedding_dim=256).embedding_dim == 256
    assert c.training.batch_size == 128        # the original is untouched


def test_wrapper_state_dict_prefix_and_save_load(tmp_path):
    c = _cfg(projection_dim=16)
    c.diffmap_config.hidden_size = 64
    m = K.RNAProteinCLIP(c)
    assert all(k.startswith("rna_protein_clip.") for k in m.state_dict())
    m.save_pretrained(str(tmp_path))
    m2 = K.RNAProteinCLIP.from_pretrained(str(tmp_path))
    for (k, v), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and (v == v2).all()
