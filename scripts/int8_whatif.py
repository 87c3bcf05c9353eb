#!/usr/bin/env python3
"""(needs an experiment build: build_variant("exp", {"CODD_EXPERIMENTS": 1}) and CODD_KNN_LIB=...: "exp_slack_pct" does not exist in the shipped library)
What would the int8 filter cost at large batches if its thresholds were tighter? (development tool; the scaled
slack is UNSOUND, timing only)"""
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
for B in (32, 64, 256):
    q = torch.randn((B, d), generator=g, device="cuda")
    for mode, pct, div8 in ((0, 100, 20), (1, 100, 20), (1, 50, 20), (1, 50, 10), (1, 35, 10)):
        ix.set_option("shadow8", mode); ix.set_option("shadow8_max_batch", 256); ix.set_option("exp_slack_pct", pct); ix.set_option("sample_div8", div8)
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
        print(f"B {B:3d} int8={mode} slack {pct:3d}% div8 {div8:2d}: step {e0.elapsed_time(e1) / n:.3f} ms  filter {kt['filter']:.3f} sample {kt['sample']:.3f} finalize {kt['finalize']:.3f} "
              f"hits/q {(ix.stat('filter_hits') - h0) / n / B:.0f} surv/q {(ix.stat('filter_survivors') - s0) / n / B:.0f} fb {ix.stat('fallback_queries') - f0}", flush=True)
