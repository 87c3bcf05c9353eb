#!/usr/bin/env python3
"""p50 latency of small batches (development tool): python scripts/lat_small.py ROWS [DIM]
Host wall clock of search_tensors + D2H of the results, per batch size, with the single-launch kernel and with the chain."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[1]); d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
g = torch.Generator(device="cuda").manual_seed(1)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    m = min(250_000, rows - c0)
    ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
out = {"rows": rows, "dim": d}
for B in (1,):
    qs = [torch.randn((B, d), generator=g, device="cuda") for _ in range(8)]
    for mode, mx in (("single_launch", 1), ("chain", 0)):
        ix.set_option("small_batch_max", mx)
        ix.set_option("profile", 256)
        lat = []
        for i in range(43):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dd, rr = ix.search_tensors(qs[i % 8], 10)
            rr.cpu(); dd.cpu()
            if i >= 3:
                lat.append((time.perf_counter() - t0) * 1e3)
        ev = ix.stat("events:filter")
        out[f"B{B}_{mode}"] = {"p50_ms": round(statistics.median(lat), 4), "min_ms": round(min(lat), 4),
                                "kernel_ms": round(ix.stat("time_ns:filter") * 1e-6 / max(ev, 1), 4)}
    ix.set_option("small_batch_max", 1)
    a = ix.search_tensors(qs[0], 10)
    ix.set_option("filter", 0)
    b = ix.search_tensors(qs[0], 10)
    ix.set_option("filter", 1)
    out[f"B{B}_exact"] = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
out["small_batch_passes"] = ix.stat("small_batch_passes")
out["fallback_queries"] = ix.stat("fallback_queries")
print(json.dumps(out))
