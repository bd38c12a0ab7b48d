"""Input format of the notebook pipeline: per-sequence [L_i, D] float arrays, NaN-padded into
[B, L_max, D] batches whose padding is recovered as a mask.  Host-side mirror of
current/rna_clip_codes.ipynb:1800-1857 (RNARBPDataset, collate_fn, create_padding_mask): same names / behaviour.
Pure data plumbing (no arithmetic on the path), so it is ordinary PyTorch host code."""
from __future__ import annotations

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset

from .modeling_seqclip import create_padding_mask  # noqa: F401  (re-export, same name as the notebook)


class RNARBPDataset(Dataset):
    """rna_clip_codes.ipynb:1806-1822: lists/arrays of per-sample [seq_len, emb_dim] embeddings."""

    def __init__(self, rna_embeddings, rbp_embeddings):
        self.rna_embeddings = rna_embeddings
        self.rbp_embeddings = rbp_embeddings

    def __len__(self):
        return len(self.rna_embeddings)

    def __getitem__(self, idx):
        return (torch.from_numpy(np.asarray(self.rna_embeddings[idx])).float(),
                torch.from_numpy(np.asarray(self.rbp_embeddings[idx])).float())


def collate_fn(batch):
    """rna_clip_codes.ipynb:1824-1838: pad both modalities to the batch maximum with NaN (masked later)."""
    rna_embs, rbp_embs = zip(*batch)
    return (pad_sequence(rna_embs, batch_first=True, padding_value=float("nan")),
            pad_sequence(rbp_embs, batch_first=True, padding_value=float("nan")))


def pack_sequences(seqs, device=None):
    """List of per-sample [L_i, ...] tensors -> (packed [sum L_i, ...], cu_seqlens int32 [B+1], max_len): the
    variable-length batch format of the packed attention path (no NaN padding, no mask)."""
    lens = torch.tensor([int(t.shape[0]) for t in seqs], dtype=torch.int32)
    cu = torch.zeros(len(seqs) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    packed = torch.cat([torch.as_tensor(t) for t in seqs], 0)
    if device is not None:
        packed, cu = packed.to(device), cu.to(device)
    return packed, cu, int(lens.max())


def collate_packed(batch):
    """Drop-in alternative to collate_fn for models with a packed path (ProteinRNACLIP.loss_packed): each modality
    becomes (packed, cu_seqlens, max_len) instead of a NaN-padded [B, L_max, D] tensor.  rna_clip_codes.ipynb:2340
    logs batches padded to L = 2542 for sequences as short as 30 tokens: that padding never reaches a kernel here."""
    a, b = zip(*batch)
    return pack_sequences(a), pack_sequences(b)


def unpad(x, valid):
    """Padded [B, L, ...] + validity mask [B, L] (True / 1 = real token, right-padded) -> packed triple."""
    lens = valid.long().sum(1)
    return pack_sequences([x[i, : int(lens[i])] for i in range(x.shape[0])])


class MemoryQueue:
    """tong/utils/data.py:154-184: fixed-size FIFO of past embeddings for contrastive negatives, with TRUE wrap-around
    (a batch that crosses the end continues at row 0; `old/clip_opt.py:76-81` resets the pointer to 0 instead and
    forgets the older rows — `OptimizedCLIPModule(cache_semantics=...)` offers both).  Same attributes (`size`, `dim`,
    `ptr`, `queue`) and the same `enqueue_dequeue(embeddings) -> queue` as the reference: the WHOLE queue is returned,
    rows that were never written are zeros.  `device=` places the buffer (the reference keeps it on the CPU)."""

    def __init__(self, size, dim, device=None):
        self.size = size
        self.dim = dim
        self.ptr = 0
        self.queue = torch.zeros(size, dim, device=device)

    @torch.no_grad()
    def enqueue_dequeue(self, embeddings):
        n = embeddings.shape[0]
        if self.queue is None:                                   # (reference :168-170)
            self.queue = embeddings.detach()
            return self.queue
        e = embeddings.detach().to(self.queue.device, self.queue.dtype)
        if n > self.size:
            raise ValueError(f"a batch of {n} rows does not fit a queue of {self.size}")   # the reference's slice assignment raises too
        if self.ptr + n > self.size:
            first = self.size - self.ptr
            self.queue[self.ptr:] = e[:first]
            self.queue[: n - first] = e[first:]
            self.ptr = n - first
        else:
            self.queue[self.ptr:self.ptr + n] = e
            self.ptr = (self.ptr + n) % self.size
        return self.queue
