"""Drop-in mirror of the reference's CLIP modules (same class names, constructor arguments, forward
signatures, output dict keys and state_dict keys), running on the libclipk HIP kernels.

  reference                                   here
  old/clip.py:8-17   CLIPEncoder              CLIPEncoder
  old/clip.py:20-36  ProjectionHead           ProjectionHead
  old/clip.py:38-73  RNAProteinCLIPModule     RNAProteinCLIPModule      (+ .loss(): fused, no B x B logits)
  old/clip.py:75-110 DiffMapProteinCLIPModule DiffMapProteinCLIPModule
  old/clip.py:112-134 RNAProteinCLIP / DiffMapProteinCLIP  (HF wrappers; key prefixes kept)
  old/clip_opt.py:9-44   OptimizedProjectionHead
  old/clip_opt.py:46-128 OptimizedCLIPModule  (FIFO protein-embedding cache, clamp(max=100), all-gather)
  old/clip_opt.py:130-151 optimized_clip_loss

There is no CPU path: calling a module with host tensors raises clip_dplm_amd._ffi.ClipkError.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import functional as KF
from .configuration_hybrid_clip import HybridCLIPConfig
from .loss import clip_loss


class KLinear(nn.Linear):
    """nn.Linear whose forward/backward are the bf16-MFMA gemm_nt / gemm_wgrad kernels (f32 master weights)."""

    precision = "bf16"          # "f32": exact-f32 MFMA (KF.set_linear_precision)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._cache = KF.WeightCache()

    def forward(self, x, act: Optional[str] = None, out_dtype=torch.float32):
        return KF.linear(x, self.weight, self.bias, self._cache, act=act, out_dtype=out_dtype, precision=self.precision)


class KLayerNorm(nn.LayerNorm):
    def forward(self, x, act: Optional[str] = None):
        return KF.layer_norm(x, self.weight, self.bias, self.eps, act=act)


class CLIPEncoder(nn.Module):
    """old/clip.py:8-17."""

    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([KLinear(config.hidden_size, config.hidden_size)
                                     for _ in range(config.num_hidden_layers)])
        self.layernorm = KLayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, x):
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            # ReLU fused into the GEMM epilogue; intermediate activations stay bf16, the last one feeds LN in f32
            x = layer(x, act="relu", out_dtype=torch.bfloat16 if i + 1 < n else torch.float32)
        return self.layernorm(x)


class ProjectionHead(nn.Module):
    """old/clip.py:20-36: Linear -> LayerNorm -> GELU -> Dropout -> Linear -> LayerNorm."""

    def __init__(self, input_dim, output_dim, hidden_dim=None, dropout=0.1):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = input_dim
        self.projection = nn.Sequential(
            KLinear(input_dim, hidden_dim),
            KLayerNorm(hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            KLinear(hidden_dim, output_dim),
            KLayerNorm(output_dim),
        )

    def forward(self, x):
        p = self.projection
        h = p[0](x)
        h = p[1](h, act="gelu")                    # LayerNorm + erf-GELU in one kernel
        h = p[3](h)                                # dropout: identity in eval / p = 0
        h = p[4](h)
        return p[5](h)


class _PairCLIPModule(nn.Module):
    A = "rna"
    B = "protein"

    def __init__(self, config: HybridCLIPConfig):
        super().__init__()
        self.config = config
        a, b = self.A, self.B
        setattr(self, f"{a}_model", CLIPEncoder(getattr(config, f"{a}_config")))
        setattr(self, f"{b}_model", CLIPEncoder(getattr(config, f"{b}_config")))
        setattr(self, f"{a}_projection", ProjectionHead(input_dim=getattr(config, f"{a}_config").hidden_size,
                                                        output_dim=config.projection_dim,
                                                        hidden_dim=config.projection_dim * 2))
        setattr(self, f"{b}_projection", ProjectionHead(input_dim=getattr(config, f"{b}_config").hidden_size,
                                                        output_dim=config.projection_dim,
                                                        hidden_dim=config.projection_dim * 2))
        self.logit_scale = nn.Parameter(torch.ones([]) * config.logit_scale_init_value)
        self.dual_stream = False       # opt-in: the two towers on two HIP streams (functional.parallel_branches)
        self._streams = None

    def embed(self, a_values, b_values):
        a, b = self.A, self.B
        ta = lambda: KF.l2_normalize(getattr(self, f"{a}_projection")(getattr(self, f"{a}_model")(a_values)))
        tb = lambda: KF.l2_normalize(getattr(self, f"{b}_projection")(getattr(self, f"{b}_model")(b_values)))
        if self.dual_stream and a_values.is_cuda:
            # each tower is a chain of microsecond kernels on 256 rows: side by side (in a captured step: two branches of
            # the hipGraph) one hides under the other, forward and backward
            if self._streams is None:
                self._streams = KF.branch_streams(2)
            return KF.parallel_branches(self._streams, (ta, tb), ((a_values,), (b_values,)))
        return ta(), tb()

    def forward(self, a_values, b_values):
        ea, eb = self.embed(a_values, b_values)
        logits = KF.sim_logits(ea, eb, self.logit_scale.exp())
        return {f"logits_per_{self.A}_{self.B}": logits, f"{self.A}_embeds": ea, f"{self.B}_embeds": eb}

    def loss(self, a_values, b_values, symmetric: bool = False, group=None):
        """Training fast path: fused similarity + CE.  symmetric=False is the reference's caller
        (old/ablation.py:16, one-sided CE); symmetric=True is rna_clip_codes.ipynb:1952-1953."""
        ea, eb = self.embed(a_values, b_values)
        return clip_loss(ea, eb, self.logit_scale.exp(), symmetric=symmetric, group=group)


class RNAProteinCLIPModule(_PairCLIPModule):
    """old/clip.py:38-73."""
    A, B = "rna", "protein"


class DiffMapProteinCLIPModule(_PairCLIPModule):
    """old/clip.py:75-110."""
    A, B = "diffmap", "protein"


class _Wrapper(nn.Module):
    """Stand-in for the HF PreTrainedModel wrappers (old/clip.py:112-134): same attribute / key prefix,
    plus minimal save_pretrained / from_pretrained (config.json + pytorch_model.bin state_dict)."""
    config_class = HybridCLIPConfig
    base_model_prefix = ""
    module_class = None

    def __init__(self, config: HybridCLIPConfig):
        super().__init__()
        self.config = config
        setattr(self, self.base_model_prefix, self.module_class(config))

    def forward(self, a_values, b_values):
        return getattr(self, self.base_model_prefix)(a_values, b_values)

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            f.write(self.config.to_json_string())
        torch.save({k: v.cpu() for k, v in self.state_dict().items()}, os.path.join(path, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, path: str):
        cfg = HybridCLIPConfig.from_dict(json.load(open(os.path.join(path, "config.json"))))
        m = cls(cfg)
        m.load_state_dict(torch.load(os.path.join(path, "pytorch_model.bin"), weights_only=True))
        return m


class RNAProteinCLIP(_Wrapper):
    base_model_prefix = "rna_protein_clip"
    module_class = RNAProteinCLIPModule


class DiffMapProteinCLIP(_Wrapper):
    base_model_prefix = "diffmap_protein_clip"
    module_class = DiffMapProteinCLIPModule


# ------------------------------------------------------------------------------------------------ clip_opt
class OptimizedProjectionHead(nn.Module):
    """old/clip_opt.py:9-44: skip(x) + layer_scale * MLP3(x); xavier-uniform weights, zero biases."""

    def __init__(self, input_dim, output_dim, hidden_dim=None, dropout=0.1, xavier: bool = True):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = input_dim * 2
        self.skip = KLinear(input_dim, output_dim)
        self.layer_scale = nn.Parameter(torch.ones(1) * 1e-4)
        self.projection = nn.Sequential(
            KLinear(input_dim, hidden_dim), KLayerNorm(hidden_dim), nn.GELU(), nn.Dropout(dropout),
            KLinear(hidden_dim, hidden_dim), KLayerNorm(hidden_dim), nn.GELU(), nn.Dropout(dropout),
            KLinear(hidden_dim, output_dim), KLayerNorm(output_dim),
        )
        if xavier:
            self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                torch.nn.init.zeros_(m.bias)

    def forward(self, x):
        p = self.projection
        h = p[3](p[1](p[0](x), act="gelu"))
        h = p[7](p[5](p[4](h), act="gelu"))
        h = p[9](p[8](h))
        return KF.SkipScaleFn.apply(self.skip(x), h, self.layer_scale)


class OptimizedCLIPModule(nn.Module):
    """old/clip_opt.py:46-128.  The cache is a non-persistent buffer (moves with .to(), not in state_dict —
    SURVEY App. A-4).  Under torch.distributed the embeddings are all-gathered differentiably (App. A-5).

    cache_semantics — both update the cache BEFORE it is used, as the reference does:
      "reference" (default): old/clip_opt.py:76-81 — a batch that does not fit resets the pointer to 0, and the
                   negatives are `cache[:ptr]`, i.e. after a wrap only the newest rows;
      "fifo":      tong/utils/data.py:154-184 (`MemoryQueue`) — true wrap-around; the negatives are every row written
                   so far, i.e. once full always the newest `cache_size` rows.  Identical to "reference" until the first
                   wrap."""

    def __init__(self, config, cache_semantics: str = "reference"):
        super().__init__()
        if cache_semantics not in ("reference", "fifo"):
            raise ValueError(f"cache_semantics must be 'reference' or 'fifo', got {cache_semantics!r}")
        self.cache_semantics = cache_semantics
        self.cache_filled = 0                   # fifo: rows written so far, capped at cache_size
        self.config = config
        self.register_buffer("protein_embedding_cache", torch.zeros(config.cache_size, config.projection_dim),
                             persistent=False)
        self.cache_ptr = 0
        self.diffmap_model = CLIPEncoder(config.diffmap_config)
        self.protein_model = CLIPEncoder(config.protein_config)
        self.diffmap_projection = OptimizedProjectionHead(input_dim=config.diffmap_config.hidden_size,
                                                          output_dim=config.projection_dim,
                                                          hidden_dim=config.projection_dim * 4)
        self.protein_projection = OptimizedProjectionHead(input_dim=config.protein_config.hidden_size,
                                                          output_dim=config.projection_dim,
                                                          hidden_dim=config.projection_dim * 4)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def update_cache(self, protein_embeds):
        batch_size = protein_embeds.size(0)
        size = self.config.cache_size
        if self.cache_semantics == "fifo":
            if batch_size > size:
                raise ValueError(f"a batch of {batch_size} rows does not fit a cache of {size}")
            e = protein_embeds.detach()
            first = min(batch_size, size - self.cache_ptr)
            self.protein_embedding_cache[self.cache_ptr:self.cache_ptr + first] = e[:first]
            if first < batch_size:
                self.protein_embedding_cache[: batch_size - first] = e[first:]
            self.cache_ptr = (self.cache_ptr + batch_size) % size
            self.cache_filled = min(size, self.cache_filled + batch_size)
            return
        if self.cache_ptr + batch_size > size:
            self.cache_ptr = 0
        self.protein_embedding_cache[self.cache_ptr:self.cache_ptr + batch_size] = protein_embeds.detach()
        self.cache_ptr = (self.cache_ptr + batch_size) % size

    def cache_rows(self):
        """The rows that serve as extra negatives now (None when there are none)."""
        n = self.cache_filled if self.cache_semantics == "fifo" else self.cache_ptr
        return self.protein_embedding_cache[:n].contiguous() if n > 0 else None

    def embed(self, diffmap_values, protein_values):
        ed = self.diffmap_projection(self.diffmap_model(diffmap_values))
        ep = self.protein_projection(self.protein_model(protein_values))
        return KF.l2_normalize(ed), KF.l2_normalize(ep)

    def forward(self, diffmap_values, protein_values, gather_distributed=True):
        ed, ep = self.embed(diffmap_values, protein_values)
        self.update_cache(ep)
        scale = self.logit_scale.exp().clamp(max=100)
        if gather_distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .distributed import all_gather_with_grad
            ed, ep = all_gather_with_grad(ed), all_gather_with_grad(ep)
        out = {
            "logits_per_diffmap_protein": KF.sim_logits(ed, ep, scale),
            "logits_per_diffmap_cache": KF.sim_logits(ed, self.cache_rows(), scale)
            if self.cache_rows() is not None else ed.new_zeros((ed.shape[0], 0)),
            "diffmap_embeds": ed,
            "protein_embeds": ep,
        }
        return out

    def loss(self, diffmap_values, protein_values, group=None):
        """Fused equivalent of optimized_clip_loss(self(diffmap, protein)) without materialised logits."""
        ed, ep = self.embed(diffmap_values, protein_values)
        self.update_cache(ep)
        cache = self.cache_rows()
        return clip_loss(ed, ep, self.logit_scale.exp().clamp(max=100), symmetric=True, cache=cache, group=group)


def optimized_clip_loss(outputs, temperature=0.07):
    """old/clip_opt.py:130-151 on the materialised logits the module returns (API compatibility), computed by the
    clipk_ce_logits kernels (row LSE over [S | S_cache], column LSE over S, fused backward) — no ATen softmax.
    `temperature` is accepted and unused, exactly like the reference; the label-smoothing tensor the reference
    builds and discards is not built (App. A-6)."""
    return KF.cross_entropy_diag(outputs["logits_per_diffmap_protein"], outputs["logits_per_diffmap_cache"],
                                 symmetric=True)
