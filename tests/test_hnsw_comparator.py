"""oracle/hnsw_cpu.c — the like-for-like ALGORITHM comparator of bench.py's cpu_baseline leg: a from-scratch HNSW with
the reference's parameters (store.py:63-68: cosine, M = 16, construction_ef = 200, search_ef = 100).  It is a checker /
baseline, never a product path: nothing under codd_query_engine_amd/ imports it."""

import os
import subprocess

import numpy as np

from oracle import hnsw_cpu as h
from oracle import knn_oracle as o


def test_parameters_are_the_reference_collection_metadata():
    assert h.REFERENCE_PARAMS == {"M": 16, "construction_ef": 200, "search_ef": 100}


def test_small_index_is_exhaustive_and_agrees_with_the_exact_oracle():
    """Below construction_ef rows every insertion sees every row: the graph search is exhaustive and must return the exact
    neighbours (ids equal; distances equal up to summation order)."""
    rng = np.random.default_rng(3)
    rows = o.normalize_rows(rng.standard_normal((120, 384)).astype(np.float32))
    q = o.normalize_rows(rng.standard_normal((16, 384)).astype(np.float32))
    ix = h.HnswIndex(rows)
    dist, ids = ix.search(q, 10)
    d_ref, i_ref = o.search(rows, "f32", q, 10)
    assert np.array_equal(ids, i_ref)
    assert np.abs(dist - d_ref).max() < 2e-6
    ix.close()


def test_recall_on_clustered_rows_and_on_the_worst_case():
    """Embedding-like (clustered) rows: recall@10 >= 0.95 at the reference's search_ef.  Isotropic Gaussian rows are the
    worst case of any graph index (every row is almost equally far from every other): recall is poor there, which is what
    bench.py reports next to the exact engine's 1.0."""
    rng = np.random.default_rng(4)
    n, d = 20_000, 128
    centres = rng.standard_normal((100, d)).astype(np.float32)
    rows = o.normalize_rows((centres[rng.integers(0, 100, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32))
    q = o.normalize_rows((centres[rng.integers(0, 100, 64)] + 0.3 * rng.standard_normal((64, d))).astype(np.float32))
    ix = h.HnswIndex(rows)
    assert ix.max_level() >= 2
    _, ids = ix.search(q, 10)
    _, exact = o.search(rows, "f32", q, 10)
    assert h.recall_at_k(ids, exact) >= 0.95
    _, ids_low = ix.search(q, 10, search_ef=10)
    assert h.recall_at_k(ids_low, exact) <= h.recall_at_k(ids, exact)
    ix.close()
    iso = o.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
    qi = o.normalize_rows(rng.standard_normal((64, d)).astype(np.float32))
    ix = h.HnswIndex(iso)
    _, ids = ix.search(qi, 10)
    _, exact = o.search(iso, "f32", qi, 10)
    r = h.recall_at_k(ids, exact)
    assert 0.2 < r < 1.0
    ix.close()


def test_the_package_never_imports_the_comparator():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["grep", "-rlE", "hnsw_cpu|from oracle|import oracle", os.path.join(root, "codd_query_engine_amd"), "--include=*.py"],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", out
