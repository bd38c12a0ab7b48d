"""CPU: run every model-level parity test of test_gpu_models.py with tests/ops_emulator.py standing in for the
HIP kernels (host wiring + tolerance calibration in the GPU-less container).  The same functions run against the
real kernels under `-m gpu`."""
import pytest
import torch

import ops_emulator
import test_gpu_models as T

CASES = [
    ("c1", T.test_clip_c1_golden_forward_loss_and_grads, {}),
    ("diffmap", T.test_clip_diffmap_golden, {}),
    ("clip_opt_bf16", T.test_clip_opt_golden_cache_loss, {"precision": "bf16"}),
    ("clip_opt_f32", T.test_clip_opt_golden_cache_loss, {"precision": "f32"}),
    ("tlayer_relu", T.test_transformer_layer_golden, {"act": "relu"}),
    ("tlayer_gelu", T.test_transformer_layer_golden, {"act": "gelu"}),
    ("notebook_f32", T.test_notebook_model_golden, {"precision": "f32"}),
    ("notebook_bf16", T.test_notebook_model_golden, {"precision": "bf16"}),
    ("notebook_slice_f32", T.test_notebook_model_position0_slice_is_exact, {"precision": "f32"}),
    ("notebook_slice_bf16", T.test_notebook_model_position0_slice_is_exact, {"precision": "bf16"}),
    ("gelu120_bf16", T.test_gelu_stack_whose_width_is_not_a_multiple_of_32, {"precision": "bf16"}),
    ("gelu120_f32", T.test_gelu_stack_whose_width_is_not_a_multiple_of_32, {"precision": "f32"}),
    ("esm_tiny", T.test_esm_tiny_golden, {}),
    ("protein_rna", T.test_protein_rna_clip_vs_oracle, {}),
    ("adamw_train", T.test_fused_adamw_training_reduces_loss, {}),
    ("trajectory", T.test_training_trajectory_matches_the_oracle, {}),
    ("icnn", T.test_icnn_transport_golden, {}),
    ("icnn_train_A", T.test_icnn_training_through_transport_map_golden, {"case": "A"}),
    ("icnn_train_B", T.test_icnn_training_through_transport_map_golden, {"case": "B"}),
    ("icnn_hessian", T.test_icnn_hessian_golden, {}),
    ("icnn_noln", T.test_icnn_without_layer_norm_golden, {}),
    ("icnn_train_ragged", T.test_icnn_training_large_ragged_batch_vs_oracle, {}),
    ("esm_proj", T.test_esm_projections_golden, {}),
    ("trimodal_f32", T.test_trimodal_contrastive_model_golden, {"precision": "f32"}),
    ("trimodal_bf16", T.test_trimodal_contrastive_model_golden, {"precision": "bf16"}),
    ("trimodal_slice", T.test_trimodal_position0_slice_is_exact, {}),
    ("trimodal_loss", T.test_trimodal_loss_pairs_kernels_vs_f64, {}),
    ("packed_varlen", T.test_packed_varlen_path_equals_padded_path, {"interleaved": False, "monkeypatch": None}),
    ("packed_varlen_il", T.test_packed_varlen_path_equals_padded_path, {"interleaved": True, "monkeypatch": None}),
    ("dropout_layer_bf16", T.test_dropout_layer_vs_masked_oracle, {"precision": "bf16"}),
    ("dropout_layer_f32", T.test_dropout_layer_vs_masked_oracle, {"precision": "f32"}),
    ("clip_opt_b128", T.test_clip_opt_b128_golden_loss_at_the_north_star_bar, {}),
    ("c1_f32", T.test_clip_c1_exact_f32_linears_reproduce_the_fp32_reference, {}),
    ("notebook_b32", T.test_notebook_model_b32_golden_loss_and_gradients, {}),
    ("esm_integration", T.test_esm_integration_get_embeddings_golden, {}),
    ("fifo_queue", T.test_cache_semantics_fifo_and_queue_loss_golden, {}),
    ("from_config", T.test_protein_rna_clip_from_config_runs, {}),
]


@pytest.mark.parametrize("name,fn,kw", CASES, ids=[c[0] for c in CASES])
def test_emulated(monkeypatch, name, fn, kw):
    ops_emulator.install(monkeypatch)
    if "monkeypatch" in kw:
        kw = dict(kw, monkeypatch=monkeypatch)
    fn(torch.device("cpu"), **kw)
