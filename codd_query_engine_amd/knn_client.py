"""
KnnClient / Collection — the chromadb-shaped object the store is constructed with.

The reference injects a `chromadb` client into MetricsSemanticMetadataStore
(store.py:43-57) and calls exactly: client.get_or_create_collection (store.py:60-69),
client.heartbeat (codd_jobs/metrics_semantic_indexer_main.py:215,330), collection.upsert
(store.py:236-238), collection.get (store.py:260), collection.query (store.py:314-316).
This module answers those calls with Chroma-shaped dicts, keeps ids / documents / metadata
on the host (string bookkeeping, not arithmetic) and sends every vector operation through
the C ABI of the HIP engine (knn_index.DeviceKnnIndex).  There is no CPU search path: the
default engine factory raises if the library or the GPU is missing.

`engine_factory(dim) -> engine` exists so that host-logic tests can stand the façade on a
checker engine of their own (tests/_oracle_engine.py); products never pass it.

Persistence (`KnnClient(path=...)`, the role of `SemanticStoreConfig.chromadb_path`, declared and
never read in the reference, codd_lib/codd_lib/config/semantic_store_config.py:13): the indexer job
and the service are different processes that share state only through the Chroma server; here they
share a directory.  Layout per collection:

    <path>/<collection>/CURRENT            name of the live generation (replaced atomically, last)
    <path>/<collection>/gen-<n>/manifest.json   name, metadata, dim, padded_dim, dtype, count
                               /rows.bin        stored rows exactly as they sit in HBM
                               /ids.json, metadatas.jsonl, documents.jsonl

`persist()` writes a new generation and flips CURRENT; `reload()` picks up a newer generation.
"""

from __future__ import annotations

import json
import os
import shutil
import time
from typing import Any, Callable, Optional, Sequence

import numpy as np

from .embedding import EmbeddingFunction, HashingEmbeddingFunction

_SUPPORTED_SPACES = ("cosine",)
FORMAT_VERSION = 1  # of the on-disk layout


def _default_engine_factory(device: str, dtype: str) -> Callable[[int], Any]:
    def make(dim: int):
        from .knn_index import DeviceKnnIndex  # raises NativeLibraryError without .so / GPU

        return DeviceKnnIndex(dim, dtype=dtype, device=device)

    return make


class Collection:
    """One named set of (id, document, metadata, embedding) rows.

    Row slots are assigned in first-insertion order and never move, so "ties -> lower row"
    means "ties -> the id inserted first".
    """

    def __init__(self, name: str, metadata: Optional[dict], embedding_function: EmbeddingFunction,
                 engine_factory: Callable[[int], Any]):
        self.name = name
        self.metadata = dict(metadata or {})
        self._embed = embedding_function
        self._engine_factory = engine_factory
        self._engine = None
        self._ids: list[str] = []
        self._slot_of: dict[str, int] = {}
        self._documents: list[Optional[str]] = []
        self._metadatas: list[Optional[dict]] = []
        self._dirty = False
        self._generation = 0

    # ------------------------------------------------------------------ helpers
    def _engine_for(self, dim: int):
        if self._engine is None:
            self._engine = self._engine_factory(dim)
        elif self._engine.dim != dim:
            raise ValueError(f"embedding dimension {dim} does not match collection dimension {self._engine.dim}")
        return self._engine

    @staticmethod
    def _as_matrix(embeddings) -> np.ndarray:
        m = np.asarray(embeddings, dtype=np.float32)
        if m.ndim == 1:
            m = m[None, :]
        if m.ndim != 2:
            raise ValueError("embeddings must be [n, d]")
        return np.ascontiguousarray(m)

    def count(self) -> int:
        return len(self._ids)

    # ------------------------------------------------------------------ writes
    def upsert(self, ids: Sequence[str], embeddings=None, metadatas: Optional[Sequence[Optional[dict]]] = None,
               documents: Optional[Sequence[Optional[str]]] = None) -> None:
        """Insert-or-replace by id (chromadb Collection.upsert; store.py:236-238)."""
        ids = list(ids)
        n = len(ids)
        if n == 0:
            return
        if any(not isinstance(i, str) or not i for i in ids):
            raise ValueError("ids must be non-empty strings")
        if len(set(ids)) != n:
            raise ValueError("duplicate ids in one upsert")
        for name, seq in (("metadatas", metadatas), ("documents", documents)):
            if seq is not None and len(seq) != n:
                raise ValueError(f"{name} has {len(seq)} entries for {n} ids")
        if embeddings is None:
            if documents is None or any(d is None for d in documents):
                raise ValueError("upsert needs embeddings or documents to embed")
            vecs = self._as_matrix(self._embed(list(documents)))
        else:
            vecs = self._as_matrix(embeddings)
        if vecs.shape[0] != n:
            raise ValueError(f"{vecs.shape[0]} embeddings for {n} ids")
        engine = self._engine_for(vecs.shape[1])

        slots = np.empty(n, dtype=np.int64)
        next_slot = len(self._ids)
        fresh = []
        for i, doc_id in enumerate(ids):
            slot = self._slot_of.get(doc_id)
            if slot is None:
                slot = next_slot
                next_slot += 1
                fresh.append(doc_id)
            slots[i] = slot
        engine.upsert(slots, vecs)  # device first: host bookkeeping only changes if it succeeded
        for doc_id in fresh:
            self._slot_of[doc_id] = len(self._ids)
            self._ids.append(doc_id)
            self._documents.append(None)
            self._metadatas.append(None)
        for i, slot in enumerate(slots.tolist()):
            if documents is not None:
                self._documents[slot] = documents[i]
            if metadatas is not None:
                self._metadatas[slot] = dict(metadatas[i]) if metadatas[i] is not None else None
        self._dirty = True

    def add(self, ids: Sequence[str], embeddings=None, metadatas=None, documents=None) -> None:
        """chromadb Collection.add: like upsert, but ids already present are left untouched."""
        keep = [i for i, d in enumerate(ids) if d not in self._slot_of]
        if not keep:
            return
        pick = lambda seq: None if seq is None else [seq[i] for i in keep]  # noqa: E731
        emb = None if embeddings is None else self._as_matrix(embeddings)[keep]
        self.upsert([ids[i] for i in keep], embeddings=emb, metadatas=pick(metadatas), documents=pick(documents))

    # ------------------------------------------------------------------ reads
    def get(self, ids: Optional[Sequence[str]] = None, limit: Optional[int] = None, offset: int = 0,
            include: Sequence[str] = ("metadatas", "documents")) -> dict:
        """chromadb Collection.get: FLAT lists; unknown ids are skipped (store.py:260-261)."""
        if ids is None:
            slots = list(range(len(self._ids)))[offset : (None if limit is None else offset + limit)]
        else:
            slots = [self._slot_of[i] for i in ids if i in self._slot_of]
        return {
            "ids": [self._ids[s] for s in slots],
            "metadatas": [self._metadatas[s] for s in slots] if "metadatas" in include else None,
            "documents": [self._documents[s] for s in slots] if "documents" in include else None,
            "embeddings": None,
        }

    def query(self, query_texts: Optional[Sequence[str]] = None, query_embeddings=None, n_results: int = 10,
              include: Sequence[str] = ("metadatas", "documents", "distances")) -> dict:
        """chromadb Collection.query: NESTED lists, one inner list per query, ascending
        distance, min(n_results, count) hits each (store.py:314-329)."""
        if (query_texts is None) == (query_embeddings is None):
            raise ValueError("give exactly one of query_texts / query_embeddings")
        if n_results < 1:
            raise ValueError("n_results must be >= 1")
        if query_embeddings is None:
            if isinstance(query_texts, str):
                query_texts = [query_texts]
            q = self._as_matrix(self._embed(list(query_texts)))
        else:
            q = self._as_matrix(query_embeddings)
        B = q.shape[0]
        empty = {"ids": [[] for _ in range(B)], "distances": [[] for _ in range(B)] if "distances" in include else None,
                 "metadatas": [[] for _ in range(B)] if "metadatas" in include else None,
                 "documents": [[] for _ in range(B)] if "documents" in include else None, "embeddings": None}
        if self._engine is None or len(self._ids) == 0 or B == 0:
            return empty
        if q.shape[1] != self._engine.dim:
            raise ValueError(f"query dimension {q.shape[1]} does not match collection dimension {self._engine.dim}")
        k = min(int(n_results), len(self._ids))
        dist, rows = self._engine.search(q, k)
        out = empty
        for b in range(B):
            hit = [(int(r), float(d)) for r, d in zip(rows[b].tolist(), dist[b].tolist()) if r >= 0]
            out["ids"][b] = [self._ids[r] for r, _ in hit]
            if out["distances"] is not None:
                out["distances"][b] = [d for _, d in hit]
            if out["metadatas"] is not None:
                out["metadatas"][b] = [self._metadatas[r] for r, _ in hit]
            if out["documents"] is not None:
                out["documents"][b] = [self._documents[r] for r, _ in hit]
        return out


    # ------------------------------------------------------------------ persistence
    def _write_generation(self, directory: str) -> None:
        """Write gen-<n+1> under `directory`, then flip CURRENT (atomic rename)."""
        gen = self._generation + 1
        name = f"gen-{gen:08d}"
        tmp = os.path.join(directory, f".{name}.tmp-{os.getpid()}")
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        n = len(self._ids)
        manifest = {"format_version": FORMAT_VERSION, "name": self.name, "metadata": self.metadata, "count": n,
                    "dim": None, "padded_dim": None, "dtype": None}
        if self._engine is not None and n:
            rows = self._engine.read_rows(0, n)
            manifest.update(dim=self._engine.dim, padded_dim=int(rows.shape[1]), dtype=getattr(self._engine, "dtype", "f32"))
            # whether every stored row is a unit vector: both MFMA filters assume it, and a reader of this generation has no
            # other way to know that some rows were written with normalize = 0 (its own norm check of rows.bin comes on top)
            stat = getattr(self._engine, "stat", None)
            manifest["all_normalized"] = bool(stat("all_normalized")) if callable(stat) else True
            rows.tofile(os.path.join(tmp, "rows.bin"))
        with open(os.path.join(tmp, "ids.json"), "w") as f:
            json.dump(self._ids, f, ensure_ascii=False)
        for fname, seq in (("metadatas.jsonl", self._metadatas), ("documents.jsonl", self._documents)):
            with open(os.path.join(tmp, fname), "w") as f:
                for item in seq:
                    f.write(json.dumps(item, ensure_ascii=False) + "\n")
        with open(os.path.join(tmp, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, ensure_ascii=False)
        final = os.path.join(directory, name)
        shutil.rmtree(final, ignore_errors=True)
        os.replace(tmp, final)
        cur_tmp = os.path.join(directory, f".CURRENT.tmp-{os.getpid()}")
        with open(cur_tmp, "w") as f:
            f.write(name)
        os.replace(cur_tmp, os.path.join(directory, "CURRENT"))
        for old in os.listdir(directory):  # keep the previous generation for a reader that is mid-load
            if old.startswith("gen-") and old not in (name, f"gen-{gen - 1:08d}"):
                shutil.rmtree(os.path.join(directory, old), ignore_errors=True)
        self._generation = gen
        self._dirty = False

    @staticmethod
    def _current_generation(directory: str) -> Optional[str]:
        try:
            with open(os.path.join(directory, "CURRENT")) as f:
                return f.read().strip() or None
        except OSError:
            return None

    def _load_generation(self, directory: str, gen_name: str) -> None:
        gdir = os.path.join(directory, gen_name)
        with open(os.path.join(gdir, "manifest.json")) as f:
            manifest = json.load(f)
        if manifest.get("format_version") != FORMAT_VERSION:
            raise ValueError(f"{gdir}: unsupported index format {manifest.get('format_version')}")
        with open(os.path.join(gdir, "ids.json")) as f:
            ids = json.load(f)
        read_lines = lambda fname: [json.loads(line) for line in open(os.path.join(gdir, fname))]  # noqa: E731
        metadatas, documents = read_lines("metadatas.jsonl"), read_lines("documents.jsonl")
        n = manifest["count"]
        if not (len(ids) == len(metadatas) == len(documents) == n):
            raise ValueError(f"{gdir}: sidecar lengths disagree with the manifest")
        engine = None
        if n and manifest["dim"]:
            dtype = manifest["dtype"]
            rows = np.fromfile(os.path.join(gdir, "rows.bin"), dtype=np.float32 if dtype == "f32" else np.uint16)
            rows = rows.reshape(n, manifest["padded_dim"])
            engine = self._engine_factory(manifest["dim"])
            if getattr(engine, "dtype", dtype) != dtype:
                raise ValueError(f"{gdir}: stored dtype {dtype} does not match the client's dtype {engine.dtype}")
            engine.load_rows(rows, 0)
            if manifest.get("all_normalized") is False and callable(getattr(engine, "set_option", None)):
                engine.set_option("all_normalized", 0)
        old = self._engine
        self._engine = engine
        if old is not None and hasattr(old, "close"):
            old.close()
        self.metadata = dict(manifest.get("metadata") or {})
        self._ids, self._metadatas, self._documents = ids, metadatas, documents
        self._slot_of = {doc_id: i for i, doc_id in enumerate(ids)}
        self._generation = int(gen_name.split("-")[1])
        self._dirty = False


class KnnClient:
    """Stands where `chromadb.HttpClient(host, port)` / `EphemeralClient()` stand.

    Args:
        device: GPU that holds the row stores ("cuda:0").
        dtype: storage dtype of the rows: "f32" | "bf16" | "f16".
        embedding_function: list[str] -> [n,d] float32; default HashingEmbeddingFunction(384).
        engine_factory: test seam, see module docstring.
    """

    def __init__(self, device: str = "cuda:0", dtype: str = "f32", embedding_function: Optional[EmbeddingFunction] = None,
                 engine_factory: Optional[Callable[[int], Any]] = None, path: Optional[str] = None):
        self.device = device
        self.dtype = dtype
        self.path = path
        self._embed = embedding_function or HashingEmbeddingFunction()
        self._engine_factory = engine_factory or _default_engine_factory(device, dtype)
        self._collections: dict[str, Collection] = {}
        if path is not None:
            os.makedirs(path, exist_ok=True)
            for entry in sorted(os.listdir(path)):
                cdir = os.path.join(path, entry)
                gen = Collection._current_generation(cdir) if os.path.isdir(cdir) else None
                if gen:
                    col = Collection(entry, None, self._embed, self._engine_factory)
                    col._load_generation(cdir, gen)
                    self._collections[col.name] = col

    # ------------------------------------------------------------------ persistence
    def _dir_of(self, name: str) -> str:
        if os.sep in name or name.startswith("."):
            raise ValueError(f"collection name {name!r} cannot be used as a directory name")
        return os.path.join(self.path, name)

    def persist(self) -> int:
        """Write every changed collection to `path` (new generation + CURRENT flip). Returns how many."""
        if self.path is None:
            return 0
        written = 0
        for col in self._collections.values():
            if col._dirty or col._generation == 0:
                cdir = self._dir_of(col.name)
                os.makedirs(cdir, exist_ok=True)
                col._write_generation(cdir)
                written += 1
        return written

    def reload(self) -> int:
        """Pick up generations another process (the indexer job) has published since we loaded."""
        if self.path is None:
            return 0
        loaded = 0
        for entry in sorted(os.listdir(self.path)):
            cdir = os.path.join(self.path, entry)
            gen = Collection._current_generation(cdir) if os.path.isdir(cdir) else None
            if not gen:
                continue
            col = self._collections.get(entry)
            if col is None:
                col = Collection(entry, None, self._embed, self._engine_factory)
                self._collections[entry] = col
            if int(gen.split("-")[1]) > col._generation:
                col._load_generation(cdir, gen)
                loaded += 1
        return loaded

    def heartbeat(self) -> int:
        """Liveness probe (indexer_main.py:215,330): nanoseconds since the epoch, like chromadb."""
        return time.time_ns()

    def get_or_create_collection(self, name: str, metadata: Optional[dict] = None,
                                 embedding_function: Optional[EmbeddingFunction] = None) -> Collection:
        if not isinstance(name, str) or not name:
            raise ValueError("collection name must be a non-empty string")
        existing = self._collections.get(name)
        if existing is not None:
            return existing
        space = (metadata or {}).get("hnsw:space", "cosine")
        if space not in _SUPPORTED_SPACES:
            raise ValueError(f"hnsw:space={space!r} is not on this path (supported: {_SUPPORTED_SPACES})")
        col = Collection(name, metadata, embedding_function or self._embed, self._engine_factory)
        self._collections[name] = col
        return col

    def create_collection(self, name: str, metadata: Optional[dict] = None, embedding_function=None) -> Collection:
        if name in self._collections:
            raise ValueError(f"Collection {name} already exists")
        return self.get_or_create_collection(name, metadata, embedding_function)

    def get_collection(self, name: str) -> Collection:
        try:
            return self._collections[name]
        except KeyError:
            raise ValueError(f"Collection {name} does not exist") from None

    def list_collections(self) -> list[str]:
        return list(self._collections)

    def delete_collection(self, name: str) -> None:
        col = self._collections.pop(name, None)
        if col is None:
            raise ValueError(f"Collection {name} does not exist")
        eng = col._engine
        if eng is not None and hasattr(eng, "close"):
            eng.close()
        if self.path is not None:
            shutil.rmtree(self._dir_of(name), ignore_errors=True)
