"""Host logic of the chromadb-shaped façade (knn_client.py) on the checker engine: the Chroma
conventions the reference relies on (store.py:236-238, :260-261, :314-329) and the façade's own
argument checking.  No GPU."""

import numpy as np
import pytest

from codd_query_engine_amd import HashingEmbeddingFunction, KnnClient
from tests._oracle_engine import OracleEngine


@pytest.fixture()
def client():
    return KnnClient(engine_factory=lambda dim: OracleEngine(dim))


def test_get_or_create_is_idempotent_and_checks_space(client):
    a = client.get_or_create_collection("c", metadata={"hnsw:space": "cosine", "hnsw:M": 16})
    assert client.get_or_create_collection("c") is a
    assert a.metadata["hnsw:M"] == 16 and client.list_collections() == ["c"]
    with pytest.raises(ValueError):
        client.get_or_create_collection("l2", metadata={"hnsw:space": "l2"})
    with pytest.raises(ValueError):
        client.create_collection("c")
    with pytest.raises(ValueError):
        client.get_collection("missing")
    assert isinstance(client.heartbeat(), int) and client.heartbeat() > 1_600_000_000 * 10**9
    client.delete_collection("c")
    assert client.list_collections() == []
    with pytest.raises(ValueError):
        client.delete_collection("c")


def test_query_shapes_follow_chroma(client):
    col = client.get_or_create_collection("c")
    empty = col.query(query_texts=["anything"], n_results=3)
    assert empty["ids"] == [[]] and empty["distances"] == [[]] and empty["metadatas"] == [[]]
    col.upsert(ids=["a", "b", "c"], documents=["cpu usage", "memory usage", "disk latency"],
               metadatas=[{"k": 1}, {"k": 2}, None])
    out = col.query(query_texts=["cpu", "disk latency"], n_results=10)   # more than stored: min(n, count)
    assert [len(x) for x in out["ids"]] == [3, 3]
    assert out["ids"][0][0] == "a" and out["ids"][1][0] == "c"
    assert all(out["distances"][b] == sorted(out["distances"][b]) for b in range(2))
    assert out["metadatas"][1][0] is None and out["documents"][0][0] == "cpu usage"
    assert isinstance(out["distances"][0][0], float)
    only_ids = col.query(query_texts="cpu", n_results=1, include=())
    assert only_ids["ids"] == [["a"]] and only_ids["distances"] is None and only_ids["metadatas"] is None
    got = col.get(ids=["c", "nope", "a"])
    assert got["ids"] == ["c", "a"] and got["metadatas"] == [None, {"k": 1}]          # flat lists, unknown ids skipped
    assert col.get(limit=2, offset=1)["ids"] == ["b", "c"]


def test_upsert_replaces_in_place_and_add_skips_existing(client):
    col = client.get_or_create_collection("c")
    col.upsert(ids=["a", "b"], documents=["alpha", "beta"], metadatas=[{"v": 1}, {"v": 1}])
    col.upsert(ids=["b", "c"], documents=["beta two", "gamma"], metadatas=[{"v": 2}, {"v": 2}])
    assert col.count() == 3 and col.get(ids=["b"])["metadatas"] == [{"v": 2}]
    assert col.get()["ids"] == ["a", "b", "c"]                                          # slots never move
    col.add(ids=["a", "d"], documents=["ALPHA CHANGED", "delta"])
    assert col.get(ids=["a"])["documents"] == ["alpha"] and col.count() == 4


def test_argument_checks(client):
    col = client.get_or_create_collection("c")
    with pytest.raises(ValueError):
        col.upsert(ids=["a", "a"], documents=["x", "y"])
    with pytest.raises(ValueError):
        col.upsert(ids=["a"], documents=["x"], metadatas=[{}, {}])
    with pytest.raises(ValueError):
        col.upsert(ids=["a"])                                       # neither embeddings nor documents
    with pytest.raises(ValueError):
        col.upsert(ids=[""], documents=["x"])
    col.upsert(ids=["a"], embeddings=np.ones((1, 8), dtype=np.float32))
    with pytest.raises(ValueError):
        col.upsert(ids=["b"], embeddings=np.ones((1, 16), dtype=np.float32))             # dimension is fixed by the first row
    with pytest.raises(ValueError):
        col.query(query_embeddings=np.ones((1, 16), dtype=np.float32))
    with pytest.raises(ValueError):
        col.query(query_texts=["x"], query_embeddings=np.ones((1, 8)))
    with pytest.raises(ValueError):
        col.query(query_texts=["x"], n_results=0)
    assert col.count() == 1                                                              # failed calls changed nothing


def test_explicit_embeddings_and_tie_order(client):
    col = client.get_or_create_collection("c")
    v = np.eye(4, dtype=np.float32)
    col.upsert(ids=["x", "y", "z", "y2"], embeddings=np.stack([v[0], v[1], v[2], v[1]]))
    out = col.query(query_embeddings=v[1][None, :], n_results=4)
    assert out["ids"][0][:2] == ["y", "y2"]                        # exact tie: the id inserted first wins
    assert out["distances"][0][0] == 0.0 and out["distances"][0][2] == 1.0


def test_hashing_embedder_is_deterministic_and_lexical():
    e = HashingEmbeddingFunction(384)
    a, b = e(["CPU utilization percentage", "cpu   UTILIZATION percentage"]), e(["Memory utilization in bytes"])
    assert a.shape == (2, 384) and a.dtype == np.float32 and (a >= 0).all()
    assert np.array_equal(a[0], a[1])                              # case and whitespace do not matter
    cos = lambda x, y: float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y)))  # noqa: E731
    q = e(["CPU utilization"])[0]
    assert cos(q, a[0]) > cos(q, b[0]) > 0
    assert e([]).shape == (0, 384)
    assert np.array_equal(HashingEmbeddingFunction(384)(["x y z"]), e(["x y z"]))  # no per-process hash salt
    with pytest.raises(ValueError):
        HashingEmbeddingFunction(4)
