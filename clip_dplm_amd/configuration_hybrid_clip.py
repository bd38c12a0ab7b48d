"""HybridCLIPConfig — same field names / defaults / to_dict keys as run1/configuration_hybrid_clip.py:93-218.

The reference class cannot be constructed on current `transformers` (SURVEY App. A-1/A-2:
AutoConfig.for_model("custom") is unregistered and `self.architectures = {dict}` collides with the HF field).
This restatement keeps the public surface — kwargs, attributes read by the model constructors
(`{rna,protein,diffmap}_config.{hidden_size,num_hidden_layers,layer_norm_eps}`, `projection_dim`,
`logit_scale_init_value`, `cache_size`), `from_configs`, `to_dict`, `create_experiment_config` — without
depending on the HF registry.
"""
from __future__ import annotations

import copy
import json
from dataclasses import asdict, dataclass
from typing import Any, Dict, Optional


@dataclass
class ModelArchitectureConfig:          # run1/configuration_hybrid_clip.py:68-79
    type: str = "mlp"                   # mlp, transformer, resnet
    num_layers: int = 2
    hidden_size: int = 512
    dropout: float = 0.1
    attention_heads: Optional[int] = 8
    intermediate_size: Optional[int] = 2048
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"
    initializer_range: float = 0.02
    use_cache: bool = True


@dataclass
class TrainingConfig:                   # run1/configuration_hybrid_clip.py:81-91
    batch_size: int = 128
    learning_rate: float = 3e-4
    weight_decay: float = 0.01
    warmup_steps: int = 1000
    max_epochs: int = 100
    gradient_clip: float = 1.0
    label_smoothing: float = 0.1
    temperature: float = 0.07
    use_amp: bool = True


class SubConfig:
    """Attribute bag standing in for AutoConfig.for_model("custom", **kwargs)."""

    model_type = "custom"

    def __init__(self, **kwargs):
        kwargs.setdefault("hidden_size", 512)
        kwargs.setdefault("num_hidden_layers", 2)
        kwargs.setdefault("layer_norm_eps", 1e-12)
        self.__dict__.update(kwargs)

    def to_dict(self) -> Dict[str, Any]:
        d = dict(self.__dict__)
        d["model_type"] = self.model_type
        return d

    def __repr__(self):
        return f"SubConfig({self.__dict__})"


def _as_sub(cfg) -> SubConfig:
    if isinstance(cfg, SubConfig):
        return cfg
    if isinstance(cfg, dict):
        d = dict(cfg)
        d.pop("model_type", None)
        return SubConfig(**d)
    if hasattr(cfg, "to_dict"):
        d = dict(cfg.to_dict())
        d.pop("model_type", None)
        return SubConfig(**d)
    return SubConfig(**{k: v for k, v in vars(cfg).items() if not k.startswith("_")})


class HybridCLIPConfig:
    model_type = "hybrid-clip"
    is_composition = True

    def __init__(self, projection_dim: int = 512, logit_scale_init_value: float = 2.6592, cache_size: int = 8192,
                 max_position_embeddings: int = 512, hidden_dropout_prob: float = 0.1,
                 attention_probs_dropout_prob: float = 0.1, use_hard_negatives: bool = True,
                 hard_negative_weight: float = 0.5, use_layer_scale: bool = True, layer_scale_init_value: float = 1e-4,
                 use_mean_pooling: bool = True, embedding_dim: int = 768,
                 architectures: Optional[Dict[str, ModelArchitectureConfig]] = None,
                 training: Optional[TrainingConfig] = None, **kwargs):
        # same error behaviour as the reference (:117-122)
        if "rna_config" not in kwargs:
            raise ValueError("`rna_config` cannot be `None`.")
        if "protein_config" not in kwargs:
            raise ValueError("`protein_config` cannot be `None`.")
        if "diffmap_config" not in kwargs:
            raise ValueError("`diffmap_config` cannot be `None`.")
        self.rna_config = _as_sub(kwargs.pop("rna_config"))
        self.protein_config = _as_sub(kwargs.pop("protein_config"))
        self.diffmap_config = _as_sub(kwargs.pop("diffmap_config"))
        self.projection_dim = projection_dim
        self.logit_scale_init_value = logit_scale_init_value
        self.cache_size = cache_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.embedding_dim = embedding_dim
        self.use_hard_negatives = use_hard_negatives
        self.hard_negative_weight = hard_negative_weight
        self.use_layer_scale = use_layer_scale
        self.layer_scale_init_value = layer_scale_init_value
        self.use_mean_pooling = use_mean_pooling
        archs = architectures or {
            "mlp": ModelArchitectureConfig(),
            "transformer": ModelArchitectureConfig(type="transformer", num_layers=6, hidden_size=768),
            "resnet": ModelArchitectureConfig(type="resnet", num_layers=4, hidden_size=512),
        }
        self.architectures = {k: (v if isinstance(v, ModelArchitectureConfig) else ModelArchitectureConfig(**v))
                              for k, v in archs.items()}
        self.training = training if isinstance(training, TrainingConfig) else TrainingConfig(**(training or {}))
        self.extra = kwargs

    @classmethod
    def from_configs(cls, rna_config, protein_config, diffmap_config, **kwargs):
        return cls(rna_config=_as_sub(rna_config).to_dict(), protein_config=_as_sub(protein_config).to_dict(),
                   diffmap_config=_as_sub(diffmap_config).to_dict(), **kwargs)

    def to_dict(self) -> Dict[str, Any]:
        out = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if k not in ("extra",)}
        out["rna_config"] = self.rna_config.to_dict()
        out["protein_config"] = self.protein_config.to_dict()
        out["diffmap_config"] = self.diffmap_config.to_dict()
        out["architectures"] = {k: asdict(v) for k, v in self.architectures.items()}
        out["training"] = asdict(self.training)
        out["model_type"] = self.__class__.model_type
        return out

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "HybridCLIPConfig":
        d = dict(d)
        d.pop("model_type", None)
        return cls(**d)

    def create_experiment_config(self, experiment_type: str, **override_kwargs) -> "HybridCLIPConfig":
        """Derived config for one experiment family (same three families and override keys as the reference's
        create_experiment_config, run1/configuration_hybrid_clip.py:195-218); unknown families return a plain copy."""
        derived = copy.deepcopy(self)

        def _embedding_sweep(c):
            for key in ("projection_dim", "embedding_dim"):
                setattr(c, key, override_kwargs.get(key, getattr(self, key)))

        def _architecture_search(c):
            name = override_kwargs.get("architecture_type", "mlp")
            c.architectures[name] = ModelArchitectureConfig(**override_kwargs.get("architecture_config", {}))

        def _training_sweep(c):
            merged = asdict(self.training)
            merged.update(override_kwargs)
            c.training = TrainingConfig(**merged)

        handlers = {"embedding_sweep": _embedding_sweep, "architecture_search": _architecture_search,
                    "training_sweep": _training_sweep}
        if experiment_type in handlers:
            handlers[experiment_type](derived)
        return derived
