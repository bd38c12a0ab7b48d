"""clip_dplm_amd — MI355X-native (gfx950) CLIP-style dual-encoder contrastive path of SrikarK-code/clip-dplm.

Python mirrors of the reference's module API on top of libclipk.so (hand-written HIP kernels behind the C ABI
in include/clipk.h).  There is no CPU fallback: every forward needs device tensors and the built library.
"""
from .configuration_hybrid_clip import HybridCLIPConfig, ModelArchitectureConfig, SubConfig, TrainingConfig
from .encoders import ESM2Encoder, ESM2_SHAPES, TransformerSeqEncoder, pool
from .esm_integration import (BiologicalDataType, ESMConfig, ESMIntegration, ESMOutput, GeneProjection,
                              ProteinProjection, create_esm_integration, get_embeddings_batch)
from .data import MemoryQueue
from .functional import set_linear_precision
from .loss import clip_loss, contrastive_loss
from .modeling_clip import (CLIPEncoder, DiffMapProteinCLIP, DiffMapProteinCLIPModule, OptimizedCLIPModule,
                            OptimizedProjectionHead, ProjectionHead, RNAProteinCLIP, RNAProteinCLIPModule,
                            optimized_clip_loss)
from .modeling_seqclip import (ProteinRNACLIP, RNARBPCLIPEncoder, RNARBPCLIPModel, RNARBPCLIPProjectionHead,
                               create_padding_mask)
from .modeling_trimodal import CellStateEncoder, ContrastiveModel, PerturbationEncoder, TransformerEncoder
from .optim import FlatParams, FusedAdamW, cosine_annealing_lr
from .training import (CosineAnnealingLR, EarlyStopping, GraphedTrainStep, evaluate_model, load_checkpoint, save_checkpoint,
                       train_epoch)
from . import training

__all__ = [
    "HybridCLIPConfig", "ModelArchitectureConfig", "TrainingConfig", "SubConfig",
    "CLIPEncoder", "ProjectionHead", "RNAProteinCLIPModule", "DiffMapProteinCLIPModule", "RNAProteinCLIP",
    "DiffMapProteinCLIP", "OptimizedProjectionHead", "OptimizedCLIPModule", "optimized_clip_loss",
    "RNARBPCLIPProjectionHead", "RNARBPCLIPEncoder", "RNARBPCLIPModel", "create_padding_mask", "ProteinRNACLIP",
    "ContrastiveModel", "CellStateEncoder", "PerturbationEncoder", "TransformerEncoder",
    "ESM2Encoder", "ESM2_SHAPES", "TransformerSeqEncoder", "pool", "clip_loss", "FlatParams", "FusedAdamW",
    "cosine_annealing_lr", "CosineAnnealingLR", "GraphedTrainStep", "EarlyStopping", "train_epoch", "evaluate_model", "save_checkpoint",
    "load_checkpoint", "ESMConfig", "ESMIntegration", "ESMOutput", "BiologicalDataType", "ProteinProjection",
    "GeneProjection", "create_esm_integration", "get_embeddings_batch", "MemoryQueue", "contrastive_loss", "set_linear_precision",
]
