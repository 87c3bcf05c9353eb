#!/usr/bin/env python3
"""resident_q on/off at 768-d (two of six query slices resident) — development tool."""
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
for B in (128, 256):
    q = torch.randn((B, d), generator=g, device="cuda")
    outs = {}
    for rep in range(2):
        for res in (0, 1):
            ix.set_option("resident_q", res)
            outs[res] = ix.search_tensors(q, k); torch.cuda.synchronize()
            n = 10
            ix.set_option("profile", n * 6 + 8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ix.search_tensors(q, k)
            e1.record(); torch.cuda.synchronize()
            kt = {name: ix.stat(f"time_ns:{name}") * 1e-6 / max(ix.stat(f"events:{name}"), 1) for name in ("filter", "sample")}
            ix.set_option("profile", 0)
            print(f"B {B:3d} resident_q={res}: step {e0.elapsed_time(e1) / n:.3f} ms  filter {kt['filter']:.3f} sample {kt['sample']:.3f}", flush=True)
    print("   equal:", bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])), flush=True)
