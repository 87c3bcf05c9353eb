#!/usr/bin/env python3
"""Development tool: which stage of the per-block int8 bound loses a neighbour (per_block = 0 / 1 / 2 / 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

for (n, d, B, seed) in ((70_000, 768, 256, 70_000 + 256 + 768), (140_000, 768, 256, 140_000 + 256 + 768), (40_000, 640, 200, 40_000 + 200 + 640)):
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = DeviceKnnIndex(d)
    ix.upsert(np.arange(n, dtype=np.int64), raw)
    for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
        ix.set_option(key, 1)
    ix.set_option("filter", 0)
    d0, r0 = ix.search(q, 10)
    ix.set_option("filter", 1)
    for pb in (0, 7, 7):
        ix.set_option("per_block", pb)
        h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
        dd, rr = ix.search(q, 10)
        bad = np.argwhere(rr != r0)
        print(f"n={n} d={d} B={B} per_block={pb}: hits/q {(ix.stat('filter_hits') - h0) / B:.1f} surv/q {(ix.stat('filter_survivors') - s0) / B:.1f} mismatching entries {len(bad)}"
              + (f" first: query {bad[0][0]} rank {bad[0][1]} got row {rr[bad[0][0], bad[0][1]]} (tile {rr[bad[0][0], bad[0][1]] // 256}) want row {r0[bad[0][0], bad[0][1]]} (tile {r0[bad[0][0], bad[0][1]] // 256}, block-in-tile {(r0[bad[0][0], bad[0][1]] % 256) // 32})" if len(bad) else ""), flush=True)
        if len(bad):
            qs = sorted(set(int(b[0]) for b in bad))
            print("   queries:", qs[:20], " wanted rows' tiles:", sorted(set(int(r0[b[0], b[1]]) // 256 for b in bad))[:20], " blocks-in-tile:", sorted(set((int(r0[b[0], b[1]]) % 256) // 32 for b in bad)))
    ix.close()
