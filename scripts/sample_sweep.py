#!/usr/bin/env python3
"""Sample-size sweep of the int8 filter path (development tool): step time and its parts for sample_rounds8 x sample_div8.
    python scripts/sample_sweep.py [rows] [batch]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d, k = 768, 10
g = torch.Generator(device="cuda").manual_seed(1)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    m = min(250_000, rows - c0)
    ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
qs = [torch.randn((B, d), generator=g, device="cuda") for _ in range(8)]
ix.set_option("shadow8_cooldown", 0)
print(f"rows {rows} B {B}")
print(f"{'rounds':>6s} {'div8':>5s} {'step ms':>8s} {'filter':>7s} {'sample':>7s} {'final':>7s} {'hits/q':>8s} {'surv/q':>7s}")
for rounds in (1, 2, 3):
    for div8 in (10, 20, 40, 80):
        ix.set_option("sample_rounds8", rounds); ix.set_option("sample_div8", div8)
        for q in qs[:2]:
            ix.search_tensors(q, k)
        torch.cuda.synchronize()
        h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
        ix.set_option("profile", 256)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for i in range(24):
            ix.search_tensors(qs[i % 8], k)
        t1.record(); torch.cuda.synchronize()
        part = {n: ix.stat(f"time_ns:{n}") * 1e-6 / max(ix.stat(f"events:{n}"), 1) for n in ("filter", "sample", "finalize")}
        hits, surv = (ix.stat("filter_hits") - h0) / 24 / B, (ix.stat("filter_survivors") - s0) / 24 / B
        print(f"{rounds:6d} {div8:5d} {t0.elapsed_time(t1) / 24:8.4f} {part['filter']:7.4f} {part['sample']:7.4f} {part['finalize']:7.4f} {hits:8.0f} {surv:7.0f}", flush=True)
