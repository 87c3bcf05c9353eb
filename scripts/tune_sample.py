#!/usr/bin/env python3
"""Development tool: end-to-end step time (B=256, 10M x 768) against the sampling knobs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, B, k = 768, 256, 10
g = torch.Generator(device="cuda").manual_seed(1)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    ix.upsert_device(c0, torch.randn((min(250_000, rows - c0), d), generator=g, device="cuda"))
q = torch.randn((B, d), generator=g, device="cuda")
for div in (20, 40, 80, 160, 320, 640):
    ix.set_option("sample_div", div)
    for _ in range(3):
        ix.search_tensors(q, k)
    torch.cuda.synchronize()
    h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
    t0 = time.perf_counter()
    for _ in range(10):
        ix.search_tensors(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    print(f"sample_div {div:4d}: {dt:.3f} ms/step  hits/query {(ix.stat('filter_hits') - h0) / 10 / B:.0f}  survivors/query {(ix.stat('filter_survivors') - s0) / 10 / B:.1f}  fallback {ix.stat('fallback_queries')}", flush=True)
