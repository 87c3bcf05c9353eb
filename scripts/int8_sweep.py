#!/usr/bin/env python3
"""sample_div8 sweep for the int8 filter (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, k = 768, 10
g = torch.Generator(device="cuda").manual_seed(1)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    ix.upsert_device(c0, torch.randn((min(250_000, rows - c0), d), generator=g, device="cuda"))
for B in (1, 32, 256):
    q = torch.randn((B, d), generator=g, device="cuda")
    for div8 in (20,):
        ix.set_option("sample_div8", div8)
        ix.search_tensors(q, k); torch.cuda.synchronize()
        h0, s0, f0 = ix.stat("filter_hits"), ix.stat("filter_survivors"), ix.stat("fallback_queries")
        n = 10
        ix.set_option("profile", n * 6 + 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ix.search_tensors(q, k)
        e1.record(); torch.cuda.synchronize()
        kt = {name: ix.stat(f"time_ns:{name}") * 1e-6 / max(ix.stat(f"events:{name}"), 1) for name in ("filter", "sample", "finalize")}
        ix.set_option("profile", 0)
        print(f"B {B:3d} div8 {div8:2d}: step {e0.elapsed_time(e1) / n:.3f} ms  filter {kt['filter']:.3f} sample {kt['sample']:.3f} finalize {kt['finalize']:.3f} "
              f"hits/q {(ix.stat('filter_hits') - h0) / n / B:.0f} surv/q {(ix.stat('filter_survivors') - s0) / n / B:.0f} fb {ix.stat('fallback_queries') - f0} passes8 {ix.stat('shadow8_passes')}", flush=True)
