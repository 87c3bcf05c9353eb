#!/usr/bin/env python3
"""Development tool: exact scan vs MFMA filter on small corpora (where should the switch sit?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

d, k = 768, 10
g = torch.Generator(device="cuda").manual_seed(1)
for rows in (6_000, 12_000, 24_000, 48_000, 100_000, 250_000, 500_000, 1_000_000):
    ix = DeviceKnnIndex(d)
    ix.upsert_device(0, torch.randn((rows, d), generator=g, device="cuda"))
    line = f"rows {rows:8d}:"
    for B in (1, 8, 32, 256):
        q = torch.randn((B, d), generator=g, device="cuda")
        res = {}
        for mode in ("scan", "filter"):
            ix.set_option("filter", 1 if mode == "filter" else 0)
            ix.set_option("filter_min_rows", 1); ix.set_option("filter_min_rows_small", 1); ix.set_option("filter_min_batch", 1)
            for _ in range(3):
                ix.search_tensors(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                d_, r_ = ix.search_tensors(q, k)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / 20 * 1e3
        line += f"  B={B}: scan {res['scan']:.3f} / filter {res['filter']:.3f} ms"
    print(line, flush=True)
    ix.close()
