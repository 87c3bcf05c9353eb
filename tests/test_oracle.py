"""The CPU oracle against its committed golden vectors and against itself
(C canonical restatement vs independent numpy fp64 brute force)."""

import os

import numpy as np
import pytest

from oracle import knn_oracle as o

CASES = ["rand_f32", "rand_bf16", "rand_f16", "odd_dim_f32", "ties_zero_f32", "near_ties_f32", "few_rows_f32", "k100_f32"]


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "knn_golden.npz"))


def load_case(golden, name):
    keys = ["raw", "q_raw", "rows", "keys", "dist", "ids", "score64", "ids64", "k", "dtype"]
    c = {k: golden[f"{name}/{k}"] for k in keys}
    c["k"] = int(c["k"])
    c["dtype"] = str(c["dtype"])
    return c


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(golden, name):
    c = load_case(golden, name)
    rows = o.to_storage(o.normalize_rows(c["raw"]), c["dtype"])
    assert np.array_equal(rows, c["rows"]), "ingest (normalise + round) drifted"
    qn = o.normalize_rows(c["q_raw"])
    keys = o.search_keys(rows, c["dtype"], qn, c["k"])
    assert np.array_equal(keys, c["keys"])
    dist, ids = o.unpack_keys(keys)
    assert np.array_equal(ids, c["ids"])
    assert np.array_equal(dist, c["dist"])  # bit-exact fp32, inf padded


@pytest.mark.parametrize("name", CASES)
def test_canonical_scores_within_fp32_tolerance_of_fp64(golden, name):
    c = load_case(golden, name)
    tol = 2e-6 if c["dtype"] == "f32" else 2e-6  # storage rounding is in the rows both sides see
    valid = c["ids"] >= 0
    score32 = 1.0 - c["dist"].astype(np.float64)
    assert np.abs(score32[valid] - c["score64"][valid]).max() <= tol
    # the rankings may only differ between rows whose fp64 scores are closer than tol
    differ = valid & (c["ids"] != c["ids64"])
    if differ.any():
        wide = o.widen(c["rows"], c["dtype"])
        qn = o.normalize_rows(c["q_raw"])
        for b, j in zip(*np.nonzero(differ)):
            s_a = float(qn[b].astype(np.float64) @ wide[c["ids"][b, j]].astype(np.float64))
            s_b = float(qn[b].astype(np.float64) @ wide[c["ids64"][b, j]].astype(np.float64))
            assert abs(s_a - s_b) <= tol


def test_tie_rule_lower_row_first(golden):
    c = load_case(golden, "ties_zero_f32")
    # rows 3, 10, 77, 200 are identical and the query points at them: they lead, in row order
    assert c["ids"][0, :4].tolist() == [3, 10, 77, 200]
    assert np.all(c["dist"][0, :4] == c["dist"][0, 0])
    # zero query: every score is +0 -> distance exactly 1, rows in index order
    assert c["ids"][1].tolist() == list(range(12))
    assert np.all(c["dist"][1] == 1.0)


def test_zero_rows_score_zero(golden):
    c = load_case(golden, "ties_zero_f32")
    assert not c["rows"][50].any() and not c["rows"][51].any()
    qn = o.normalize_rows(c["q_raw"])
    assert o.canon_dot(qn[2], c["rows"][50]) == 0.0


def test_fewer_rows_than_k_pads(golden):
    c = load_case(golden, "few_rows_f32")
    assert (c["ids"][:, :6] >= 0).all() and (c["ids"][:, 6:] == -1).all()
    assert np.isinf(c["dist"][:, 6:]).all()


def test_canonical_dot_matches_numpy_emulation():
    """Independent restatement of the canonical order in numpy (fp64-emulated fmaf)."""
    rng = np.random.default_rng(5)
    for dtype, E in (("f32", 4), ("bf16", 8)):
        d = 192
        q = o.normalize_rows(rng.standard_normal((1, d)).astype(np.float32))[0]
        c = o.widen(o.to_storage(o.normalize_rows(rng.standard_normal((1, d)).astype(np.float32)), dtype), dtype)[0]
        acc = np.zeros(64, dtype=np.float32)
        for j in range(d // E):
            lane = j % 64
            for e in range(E):
                acc[lane] = np.float32(np.float64(q[j * E + e]) * np.float64(c[j * E + e]) + np.float64(acc[lane]))
        for stride in (32, 16, 8, 4, 2, 1):
            acc = (acc + acc[np.arange(64) ^ stride]).astype(np.float32)
        assert abs(float(acc[0]) - o.canon_dot(q, c, dtype)) <= 1.2e-7  # double rounding may cost an ulp


def test_merge_keys_is_topk_of_union():
    rng = np.random.default_rng(9)
    rows = o.normalize_rows(rng.standard_normal((900, 64)).astype(np.float32))
    qn = o.normalize_rows(rng.standard_normal((4, 64)).astype(np.float32))
    whole = o.search_keys(rows, "f32", qn, 10)
    parts = [o.search_keys(rows[a:b], "f32", qn, 10, row_base=a) for a, b in ((0, 300), (300, 301), (301, 900))]
    merged = o.merge_keys(np.concatenate(parts, axis=1), 10)
    assert np.array_equal(merged, whole)


def test_fast_baseline_agrees_with_canonical():
    rng = np.random.default_rng(11)
    rows = o.normalize_rows(rng.standard_normal((5000, 128)).astype(np.float32))
    qn = o.normalize_rows(rng.standard_normal((3, 128)).astype(np.float32))
    d_fast, r_fast = o.search_fast_f32(rows, qn, 10)
    d_can, r_can = o.search(rows, "f32", qn, 10)
    assert np.abs(d_fast - d_can).max() <= 2e-6
    assert (r_fast == r_can).mean() > 0.9
