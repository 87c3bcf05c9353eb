"""
oracle/gen_knn_golden.py — writes tests/golden/knn_golden.npz: small seeded k-NN cases
(inputs AND expected outputs) from the CPU oracle, so that (a) the oracle itself is pinned
against regressions and (b) the GPU parity tests have fixtures that do not depend on numpy's
RNG stream staying stable.  The reference holds no numeric vectors for this call
(SURVEY.md §8c): these are build-generated, "parity unpinned" against ChromaDB.

Cases: random unit rows; exact duplicates (ties -> lower row); near-ties one ulp apart;
zero rows (score 0 -> distance 1); fewer rows than k; a d that is not a multiple of 64.
TEST INFRASTRUCTURE.  Run: python oracle/gen_knn_golden.py
"""

from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import knn_oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "knn_golden.npz")


def case(rng, n, d, B, k, dtype, mutate=None):
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q_raw = rng.standard_normal((B, d)).astype(np.float32)
    if mutate:
        mutate(raw, q_raw)
    rows = o.to_storage(o.normalize_rows(raw), dtype)
    qn = o.normalize_rows(q_raw)
    keys = o.search_keys(rows, dtype, qn, k)
    dist, ids = o.unpack_keys(keys)
    sc64, ids64 = o.search_f64(o.widen(rows, dtype), qn, k)
    return {"raw": raw, "q_raw": q_raw, "rows": rows, "keys": keys, "dist": dist, "ids": ids,
            "score64": sc64, "ids64": ids64, "k": np.int64(k), "dtype": np.array(dtype)}


def main():
    rng = np.random.default_rng(20261004)
    cases = {}
    cases["rand_f32"] = case(rng, 1536, 64, 8, 10, "f32")
    cases["rand_bf16"] = case(rng, 1024, 128, 5, 10, "bf16")
    cases["rand_f16"] = case(rng, 512, 64, 3, 7, "f16")
    cases["odd_dim_f32"] = case(rng, 300, 100, 4, 16, "f32")

    def dup(raw, q):
        raw[10] = raw[3]; raw[200] = raw[3]; raw[77] = raw[3]      # exact duplicates of row 3
        q[0] = raw[3] * 2.5                                          # query pointing at them
        raw[50] = 0.0; raw[51] = 0.0                                 # zero rows
        q[1] = 0.0                                                   # zero query: every score 0
    cases["ties_zero_f32"] = case(rng, 256, 64, 3, 12, "f32", dup)

    def near(raw, q):
        base = raw[5].copy()
        for j in range(1, 9):                                        # 8 rows differing in one element by tiny steps
            raw[5 + j] = base
            raw[5 + j, 0] = np.nextafter(base[0], np.float32(10.0), dtype=np.float32) if j % 2 else base[0]
            raw[5 + j, 1] = base[1] * np.float32(1.0 + j * 1e-7)
        q[0] = base
    cases["near_ties_f32"] = case(rng, 128, 64, 2, 10, "f32", near)
    cases["few_rows_f32"] = case(rng, 6, 64, 2, 10, "f32")
    cases["k100_f32"] = case(rng, 700, 64, 2, 100, "f32")

    flat = {f"{name}/{key}": val for name, c in cases.items() for key, val in c.items()}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **flat)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT) // 1024, "KiB", list(cases))


if __name__ == "__main__":
    main()
