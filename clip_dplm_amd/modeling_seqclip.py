"""Sequence-level CLIP models on the kernel-backed encoders.

  * RNARBPCLIPProjectionHead / RNARBPCLIPEncoder / RNARBPCLIPModel — the notebook model that was actually
    trained (current/rna_clip_codes.ipynb:1901-1954), same names, forward signature `(rna_emb, rbp_emb) ->
    (rna_embed, rbp_embed, loss)`, NaN-padding convention and batch-axis attention quirk.
  * ProteinRNACLIP — BASELINE configs 2-4: ESM-2 protein encoder (third-party arithmetic, see encoders.py) +
    the 6 x 768 "transformer" RNA encoder of run1/configuration_hybrid_clip.py:153-157 + the ProjectionHeads and
    logit_scale of old/clip.py:38-54, returning the dict of old/clip.py:69-73 or the fused loss.
"""
from __future__ import annotations

import os

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import functional as KF
from . import ops
from .encoders import ESM2Encoder, ESM2_SHAPES, TransformerSeqEncoder, pool, pool_packed
from .loss import clip_loss
from .modeling_clip import KLayerNorm, KLinear, OptimizedProjectionHead, ProjectionHead


def create_padding_mask(tensor):
    """rna_clip_codes.ipynb:1841-1857: True = valid, False = NaN padding.  (tiny elementwise: plumbing)"""
    return ~torch.isnan(tensor).any(dim=-1)


class RNARBPCLIPProjectionHead(OptimizedProjectionHead):
    """rna_clip_codes.ipynb:1901-1909 (= OptimizedProjectionHead with hidden 2*input_dim, default torch init)."""

    def __init__(self, input_dim, output_dim):
        super().__init__(input_dim, output_dim, hidden_dim=input_dim * 2, dropout=0.1, xavier=False)


class RNARBPCLIPEncoder(TransformerSeqEncoder):
    """rna_clip_codes.ipynb:1911-1923: 3 x nn.TransformerEncoderLayer(d, nhead=8, ffn=4d) + LayerNorm.
    precision: "f32" (default: the notebook trains it without autocast, ipynb:2061-2089) or "bf16" (bf16-MFMA kernels)."""

    def __init__(self, embed_dim, num_layers=3, dropout: float = 0.1, precision: str = "f32"):
        # position-0 pooling (no averaging over rows): keep the residual stream in f32 (encoders.POSTLN_BF16_RESIDUAL)
        super().__init__(embed_dim=embed_dim, num_layers=num_layers, nhead=8, dim_feedforward=embed_dim * 4,
                         activation="relu", layer_norm_eps=1e-5, final_eps=1e-5, dropout=dropout, residual_dtype="f32",
                         precision=precision)


class RNARBPCLIPModel(nn.Module):
    """rna_clip_codes.ipynb:1925-1954.

    MI355X-first treatment of the model's batch-axis attention: the notebook hands [B, L, D] to
    batch_first=False layers, so attention mixes the B samples AT EACH POSITION, the FFN / LayerNorms are per token, and
    only `enc[:, 0]` is read (ipynb:1944-1949) - positions 1 .. L-1 (up to 2541 of 2542) never reach the embeddings, the
    loss or any parameter gradient.  `_encode` therefore slices to position 0 BEFORE the encoder (exact: same embeds, loss
    and gradients; `RNARBPCLIPEncoder.forward` itself still returns every position it is given), and with B rows left
    the whole model runs in exact f32 by default (`precision="f32"`: the reference trains it in fp32; a bf16-rounded
    weight alone moved the loss by 2e-3).  `precision="bf16"` keeps the bf16-MFMA kernels;
    `slice_first_position=False` encodes every position as the notebook does (tests compare the two)."""

    def __init__(self, rna_dim=120, rbp_dim=1280, projection_dim=512, dropout: float = 0.1, precision: str = "f32",
                 slice_first_position: bool = True):
        super().__init__()
        # `dropout` is the notebook's nn.TransformerEncoderLayer default (0.1): in train() mode the post-LN stack
        # applies it at the layer's four sites with counter-based masks (TransformerSeqEncoder); parity
        # and benchmark runs use eval() or dropout=0.0.
        self.rna_encoder = RNARBPCLIPEncoder(rna_dim, dropout=dropout, precision=precision)
        self.rbp_encoder = RNARBPCLIPEncoder(rbp_dim, dropout=dropout, precision=precision)
        self.rna_projection = RNARBPCLIPProjectionHead(rna_dim, projection_dim)
        self.rbp_projection = RNARBPCLIPProjectionHead(rbp_dim, projection_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.slice_first_position = bool(slice_first_position)
        # opt-in: the two towers on two HIP streams.  Sliced to position 0 each tower is a chain of ~10 us kernels, and the
        # rna tower's (d = 120) do not fill a tenth of the chip: side by side the short chain hides under the long one, in
        # the forward and - autograd replays each node on its forward stream - in the backward; in a captured step they
        # are two branches of the hipGraph
        self.dual_stream = False
        self._streams = None
        KF.set_linear_precision(self, precision)

    def _encode(self, encoder, emb):
        # The notebook hands (B, L, D) to batch_first=False layers with the mask transposed to (L, B): attention
        # mixes the B samples at each position.  Same arithmetic here: positions become the kernel's batch axis.
        if self.slice_first_position:
            emb = emb[:, :1]                                              # position 0 is all `enc[:, 0]` depends on
        valid = create_padding_mask(emb)                                  # [B, L]
        x = torch.nan_to_num(emb, 0.0).transpose(0, 1).contiguous()       # [L, B, D]
        # keys = samples.  The kernels take 1 = valid as bytes: the bool mask reinterpreted, instead of the round trip
        # through nn.TransformerEncoder's padding-mask convention (~valid here, ~mask and a cast inside: three launches)
        y = encoder(x, _valid_u8=valid.transpose(0, 1).contiguous().view(torch.uint8))
        # == enc[:, 0] in the notebook's layout (one position: a view - a select's backward is a zero-fill + a copy launch)
        return y.view(y.shape[1], y.shape[2]) if y.shape[0] == 1 else y[0]

    def forward(self, rna_emb, rbp_emb):
        if self.dual_stream and rna_emb.is_cuda:
            if self._streams is None:
                self._streams = KF.branch_streams(2)
            rna_embed, rbp_embed = KF.parallel_branches(
                self._streams,
                (lambda: KF.l2_normalize(self.rna_projection(self._encode(self.rna_encoder, rna_emb))),
                 lambda: KF.l2_normalize(self.rbp_projection(self._encode(self.rbp_encoder, rbp_emb)))),
                ((rna_emb,), (rbp_emb,)))
        else:
            rna_embed = KF.l2_normalize(self.rna_projection(self._encode(self.rna_encoder, rna_emb)))
            rbp_embed = KF.l2_normalize(self.rbp_projection(self._encode(self.rbp_encoder, rbp_emb)))
        loss = clip_loss(rna_embed, rbp_embed, self.logit_scale.exp(), symmetric=True, group=None)
        return rna_embed, rbp_embed, loss


# mean pooling straight out of the encoders' final LayerNorm (clipk_layernorm_meanpool_*): nothing of size [B, L, d] is
# written for the pooling.  CLIPK_FUSED_POOL=0 keeps the two-kernel path (LayerNorm, then pool) for A/B runs.
FUSED_MEAN_POOL = os.environ.get("CLIPK_FUSED_POOL", "1") != "0"


class ProteinRNACLIP(nn.Module):
    """BASELINE configs 2-4: protein ids -> ESM-2, RNA features -> 6-layer post-LN transformer; masked-mean
    pooling (`use_mean_pooling`), ProjectionHead (hidden 2P), L2 normalise, exp(logit_scale) similarity,
    symmetric InfoNCE."""

    def __init__(self, esm: str = "esm2_t12_35M_UR50D", rna_dim: int = 768, rna_layers: int = 6, rna_heads: int = 8,
                 rna_ffn: int = 2048, rna_act: str = "gelu", projection_dim: int = 512,
                 logit_scale_init_value: float = 2.6592, layer_norm_eps: float = 1e-12, pooling: str = "mean",
                 initializer_range: float = 0.02, freeze_protein_encoder: bool = False, rna_dropout: float = 0.0):
        super().__init__()
        self.protein_model = ESM2Encoder.from_name(esm, initializer_range=initializer_range)
        self.rna_model = TransformerSeqEncoder(rna_dim, rna_layers, rna_heads, rna_ffn, rna_act, layer_norm_eps,
                                               dropout=rna_dropout)
        for m in self.rna_model.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, initializer_range)
                nn.init.zeros_(m.bias)
        for l in self.rna_model.layers:
            nn.init.normal_(l.self_attn.in_proj_weight, 0.0, initializer_range)
        self.rna_projection = ProjectionHead(rna_dim, projection_dim, hidden_dim=projection_dim * 2)
        self.protein_projection = ProjectionHead(self.protein_model.hidden_size, projection_dim,
                                                 hidden_dim=projection_dim * 2)
        self.logit_scale = nn.Parameter(torch.ones([]) * logit_scale_init_value)
        self.pooling = pooling
        self.dual_stream = False           # opt-in: enqueue the two towers on separate HIP streams
        self.micro_batches = 1             # with dual_stream: split the batch into this many stream pairs
        self._streams = None
        if freeze_protein_encoder:                       # triple_flow/3_esm_integration.py:83-84
            for p in self.protein_model.parameters():
                p.requires_grad_(False)

    @classmethod
    def from_config(cls, config, esm: str = "esm2_t12_35M_UR50D", architecture: str = "transformer",
                    freeze_protein_encoder: bool = False, **overrides):
        """Build the model a `HybridCLIPConfig` describes (run1/configuration_hybrid_clip.py:93-166): the RNA tower is
        `config.architectures[architecture]` (a `ModelArchitectureConfig`, :68-79; the "transformer" default of
        :153-157 is BASELINE's 6 x 768 / 8 heads / 2048 / gelu / eps 1e-12), the heads project to
        `config.projection_dim`, the temperature starts at `config.logit_scale_init_value` and `config.use_mean_pooling`
        (:109, declared and never read by the reference, App. A-18) selects masked-mean pooling, else position 0.  The
        protein tower is the ESM-2 shape `esm` (the config has no ESM section).  `arch.dropout` (0.1) is
        nn.TransformerEncoderLayer's dropout and acts in train() mode only."""
        arch = config.architectures[architecture]
        if arch.type != "transformer":
            raise ValueError(f"architectures[{architecture!r}].type = {arch.type!r}: the sequence tower is a transformer "
                             f"(the 'mlp' family is RNAProteinCLIPModule / DiffMapProteinCLIPModule)")
        kw = dict(esm=esm, rna_dim=arch.hidden_size, rna_layers=arch.num_layers, rna_heads=arch.attention_heads or 8,
                  rna_ffn=arch.intermediate_size or 4 * arch.hidden_size, rna_act=arch.hidden_act,
                  projection_dim=config.projection_dim, logit_scale_init_value=config.logit_scale_init_value,
                  layer_norm_eps=arch.layer_norm_eps, pooling="mean" if config.use_mean_pooling else "first",
                  initializer_range=arch.initializer_range, freeze_protein_encoder=freeze_protein_encoder,
                  rna_dropout=arch.dropout)
        kw.update(overrides)
        model = cls(**kw)
        model.config = config
        return model

    def _embed_rna(self, rna_values, rna_mask):
        kpm = None if rna_mask is None else ~rna_mask.bool()
        if self.pooling == "mean" and FUSED_MEAN_POOL and ops.meanpool_fused_supported(rna_values.shape[-1]):
            # final LayerNorm + masked mean in one kernel
            return KF.l2_normalize(self.rna_projection(self.rna_model.forward_pooled(rna_values, kpm)))
        hr = self.rna_model(rna_values, src_key_padding_mask=kpm)
        return KF.l2_normalize(self.rna_projection(pool(hr, rna_mask, self.pooling)))

    def _embed_protein(self, protein_ids, protein_mask):
        if self.pooling == "mean" and FUSED_MEAN_POOL and ops.meanpool_fused_supported(self.protein_model.hidden_size):
            return KF.l2_normalize(self.protein_projection(self.protein_model.forward_pooled(protein_ids, protein_mask)))
        hp = self.protein_model(protein_ids, attention_mask=protein_mask)
        return KF.l2_normalize(self.protein_projection(pool(hp, protein_mask, self.pooling)))

    def embed(self, rna_values, protein_ids, rna_mask=None, protein_mask=None):
        """rna_values [B, Lr, rna_dim] f32; protein_ids [B, Lp] int64; masks [B, L] with 1 = valid.

        The two towers are independent until the loss, so they are enqueued on two HIP streams: the memory-bound
        kernels of one tower (LayerNorm, attention, GEMM epilogues) overlap the MFMA-bound phases of the other, in the
        forward and — because autograd replays each node on its forward stream — in the backward."""
        if not (self.dual_stream and rna_values.is_cuda):
            nmb = max(1, int(self.micro_batches))
            if nmb == 1:
                return self._embed_rna(rna_values, rna_mask), self._embed_protein(protein_ids, protein_mask)
            # samples are independent up to the loss: encode them in `micro_batches` chunks, so that a layer's
            # activations (755 MB of qkv per ESM layer at B = 1024) fit the 256 MiB Infinity Cache between the
            # kernel that writes them and the kernel that reads them
            B = rna_values.shape[0]
            cuts = [(i * B) // nmb for i in range(nmb + 1)]
            sl = lambda t, i: None if t is None else t[cuts[i]:cuts[i + 1]]
            er = torch.cat([self._embed_rna(sl(rna_values, i), sl(rna_mask, i)) for i in range(nmb)], 0)
            ep = torch.cat([self._embed_protein(sl(protein_ids, i), sl(protein_mask, i)) for i in range(nmb)], 0)
            return er, ep
        main = torch.cuda.current_stream()
        nmb = max(1, int(self.micro_batches))
        if self._streams is None or len(self._streams) != 2 * nmb:
            self._streams = tuple(torch.cuda.Stream() for _ in range(2 * nmb))
        B = rna_values.shape[0]
        cuts = [(i * B) // nmb for i in range(nmb + 1)]
        ers, eps = [], []
        for i in range(nmb):                      # samples are independent up to the loss: micro-batches on own streams
            lo, hi = cuts[i], cuts[i + 1]
            s1, s2 = self._streams[2 * i], self._streams[2 * i + 1]
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                ers.append(self._embed_rna(rna_values[lo:hi], None if rna_mask is None else rna_mask[lo:hi]))
            with torch.cuda.stream(s2):
                eps.append(self._embed_protein(protein_ids[lo:hi], None if protein_mask is None else protein_mask[lo:hi]))
        for s in self._streams:
            main.wait_stream(s)
        for t in ers + eps:
            t.record_stream(main)
        # the backward replays each tower on its stream and the kernels add parameter gradients straight into .grad views
        # there: JoinStreamsFn makes the calling stream wait for the tower streams when the backward pass ends (the
        # autograd engine itself only syncs the streams of AccumulateGrad leaves)
        joined = KF.JoinStreamsFn.apply(tuple(self._streams), *(ers + eps))
        ers, eps = list(joined[:nmb]), list(joined[nmb:])
        er = ers[0] if nmb == 1 else torch.cat(ers, 0)
        ep = eps[0] if nmb == 1 else torch.cat(eps, 0)
        return er, ep

    def embed_packed(self, rna_packed, rna_cu, rna_max_len, protein_ids_packed, protein_cu, protein_max_len):
        """Variable-length batches without padding: rna_packed f32 [T_r, rna_dim], protein_ids_packed
        int64 [T_p], cu_seqlens int32 [B+1] each (data.collate_packed builds them).  Same embeddings as embed() on the
        padded batch with masks; the padded rows are never computed."""
        hr = self.rna_model.forward_packed(rna_packed, rna_cu, rna_max_len)
        hp = self.protein_model.forward_packed(protein_ids_packed, protein_cu, protein_max_len)
        er = KF.l2_normalize(self.rna_projection(pool_packed(hr, rna_cu, self.pooling)))
        ep = KF.l2_normalize(self.protein_projection(pool_packed(hp, protein_cu, self.pooling)))
        return er, ep

    def loss_packed(self, rna_packed, rna_cu, rna_max_len, protein_ids_packed, protein_cu, protein_max_len, group=None,
                    symmetric: bool = True):
        er, ep = self.embed_packed(rna_packed, rna_cu, rna_max_len, protein_ids_packed, protein_cu, protein_max_len)
        return clip_loss(er, ep, self.logit_scale.exp(), symmetric=symmetric, group=group)

    def forward(self, rna_values, protein_ids, rna_mask=None, protein_mask=None):
        er, ep = self.embed(rna_values, protein_ids, rna_mask, protein_mask)
        return {"logits_per_rna_protein": KF.sim_logits(er, ep, self.logit_scale.exp()),
                "rna_embeds": er, "protein_embeds": ep}

    def loss(self, rna_values, protein_ids, rna_mask=None, protein_mask=None, group=None, symmetric: bool = True):
        er, ep = self.embed(rna_values, protein_ids, rna_mask, protein_mask)
        return clip_loss(er, ep, self.logit_scale.exp(), symmetric=symmetric, group=group)
