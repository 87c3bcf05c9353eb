"""
Embedding step in front of the k-NN (what chromadb does client-side before the search,
reference call sites store.py:236-238 and :314-316).

ChromaDB's default embedder (all-MiniLM-L6-v2, 384-d, ONNX) is fetched from the network on
first use and is unavailable offline, so the default here is a deterministic lexical
feature-hashing embedder of the same width: unsigned hashed counts of lower-cased word
tokens plus their character trigrams.  Non-negative features keep cosine in [0, 1] like a
sentence embedder does for related text, and lexical overlap reproduces the ordering the
reference's own store tests assert (tests/test_reference_scenarios.py).  Any callable
`list[str] -> [n, d] float32` can be injected instead (e.g. a local MiniLM on the GPU).
"""

from __future__ import annotations

import re
import zlib
from typing import Protocol, Sequence

import numpy as np

_TOKEN = re.compile(r"\w+", re.UNICODE)


class EmbeddingFunction(Protocol):
    def __call__(self, texts: Sequence[str]) -> np.ndarray: ...


class HashingEmbeddingFunction:
    """Deterministic across processes and machines (crc32, no Python hash randomisation)."""

    def __init__(self, dim: int = 384, trigram_weight: float = 0.35):
        if dim < 8:
            raise ValueError("dim must be >= 8")
        self.dim = int(dim)
        self.trigram_weight = float(trigram_weight)
        self._cache: dict = {}
        self._cache_max = 1 << 18   # tokens remembered (a few tens of MB at most)

    def _bucket(self, feature: str) -> int:
        return zlib.crc32(feature.encode("utf-8")) % self.dim

    def _token_features(self, tok: str):
        """(buckets, weights) of one lower-cased token: its word feature, then its character trigrams, in that order.
        Cached: metric vocabularies repeat their tokens thousands of times, and hashing (encode + crc32 per feature, a
        Python-level loop) was most of a query's host time (bench.py `dropin_call`)."""
        hit = self._cache.get(tok)
        if hit is None:
            padded = f"^{tok}$"
            buckets = [self._bucket("w:" + tok)] + [self._bucket("t:" + padded[i : i + 3]) for i in range(len(padded) - 2)]
            weights = np.full(len(buckets), self.trigram_weight, dtype=np.float32)
            weights[0] = 1.0
            hit = (np.asarray(buckets, dtype=np.intp), weights)
            if len(self._cache) < self._cache_max:
                self._cache[tok] = hit
        return hit

    def embed_one(self, text: str) -> np.ndarray:
        return self([text])[0]

    def __call__(self, texts: Sequence[str]) -> np.ndarray:
        out = np.zeros((len(texts), self.dim), dtype=np.float32)
        idx, wts, rows = [], [], []
        for r, text in enumerate(texts):
            for tok in _TOKEN.findall(text.lower()):
                b, w = self._token_features(tok)
                idx.append(b)
                wts.append(w)
                rows.append((r, len(b)))
        if idx:
            # one unbuffered, in-order float32 accumulation for the whole batch: each bucket receives its additions in text
            # order, i.e. the same bits as the feature-by-feature loop this replaces
            flat = np.concatenate(idx) + np.repeat(np.asarray([r for r, _ in rows], dtype=np.intp) * self.dim, [m for _, m in rows])
            np.add.at(out.reshape(-1), flat, np.concatenate(wts))
        return out


class LocalTransformerEmbeddingFunction:
    """Sentence embeddings from a transformer encoder stored in a LOCAL directory (nothing is downloaded).

    The counterpart of chromadb's default embedder for installations that have the all-MiniLM-L6-v2 files (or any
    BERT-style encoder in Hugging Face layout: config.json, tokenizer files, weights) on disk: tokenise, run the
    encoder with torch (on the GPU when `device` is a cuda device, so text -> vector -> top-k stays on one device),
    mean-pool the last hidden states over the attention mask and L2-normalise — the pooling the sentence-transformers
    MiniLM family is trained with.  Returns float32 [n, hidden_size].

    Loading uses `local_files_only=True` and safetensors / `weights_only` checkpoints only; a missing directory is an
    error, never a fetch.
    """

    def __init__(self, model_dir: str, device: str = "cpu", max_length: int = 256, batch_size: int = 64):
        import os

        if not os.path.isdir(model_dir):
            raise FileNotFoundError(f"no model directory at {model_dir!r} (this embedder never downloads)")
        import torch
        from transformers import AutoModel, AutoTokenizer

        self._torch = torch
        self.device = torch.device(device)
        self.max_length = int(max_length)
        self.batch_size = int(batch_size)
        self.tokenizer = AutoTokenizer.from_pretrained(model_dir, local_files_only=True)
        self.model = AutoModel.from_pretrained(model_dir, local_files_only=True).to(self.device).eval()
        self.dim = int(self.model.config.hidden_size)

    def embed_tensor(self, texts: Sequence[str]):
        """[n, dim] float32 tensor on `self.device` (what a device-resident pipeline feeds to search_tensors)."""
        torch = self._torch
        out = []
        with torch.inference_mode():
            for i in range(0, len(texts), self.batch_size):
                enc = self.tokenizer(list(texts[i : i + self.batch_size]), padding=True, truncation=True, max_length=self.max_length,
                                     return_tensors="pt").to(self.device)
                hidden = self.model(**enc).last_hidden_state.float()
                mask = enc["attention_mask"].unsqueeze(-1).float()
                pooled = (hidden * mask).sum(1) / mask.sum(1).clamp(min=1e-9)
                out.append(torch.nn.functional.normalize(pooled, p=2, dim=1))
        if not out:
            return torch.zeros((0, self.dim), dtype=torch.float32, device=self.device)
        return torch.cat(out, 0)

    def __call__(self, texts: Sequence[str]) -> np.ndarray:
        return self.embed_tensor(texts).cpu().numpy().astype(np.float32)
