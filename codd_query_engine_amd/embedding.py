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

    def _bucket(self, feature: str) -> int:
        return zlib.crc32(feature.encode("utf-8")) % self.dim

    def embed_one(self, text: str) -> np.ndarray:
        v = np.zeros(self.dim, dtype=np.float32)
        for tok in _TOKEN.findall(text.lower()):
            v[self._bucket("w:" + tok)] += 1.0
            padded = f"^{tok}$"
            for i in range(len(padded) - 2):
                v[self._bucket("t:" + padded[i : i + 3])] += self.trigram_weight
        return v

    def __call__(self, texts: Sequence[str]) -> np.ndarray:
        if len(texts) == 0:
            return np.zeros((0, self.dim), dtype=np.float32)
        return np.stack([self.embed_one(t) for t in texts]).astype(np.float32)
