#!/usr/bin/env python3
"""A few extra operating points for the record (development tool): other widths, dtypes and k on one MI355X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

POINTS = [(10_000_000, 384, "f32", 256, 10), (10_000_000, 512, "f32", 256, 10), (10_000_000, 256, "f32", 256, 10), (10_000_000, 384, "f32", 64, 10), (10_000_000, 384, "f32", 1, 10), (12_500_000, 1024, "f16", 256, 10), (12_500_000, 1024, "f16", 1, 10),
          (10_000_000, 768, "f32", 256, 100), (10_000_000, 768, "bf16", 256, 10)]
for rows, d, dtype, B, k in POINTS:
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(d, dtype)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        ix.upsert_device(c0, torch.randn((min(250_000, rows - c0), d), generator=g, device="cuda"))
    q = torch.randn((B, d), generator=g, device="cuda")
    ix.set_option("resident_q", int(os.environ.get("CODD_RESIDENT_Q", "1")))
    for _ in range(3):
        ix.search_tensors(q, k)
    torch.cuda.synchronize()
    n = 10
    ix.set_option("profile", n * 6 + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ix.search_tensors(q, k)
    e1.record(); torch.cuda.synchronize()
    step = e0.elapsed_time(e1) / n
    f = ix.stat("time_ns:filter") * 1e-6 / max(ix.stat("events:filter"), 1)
    print(f"{rows:>9d} x {d:4d} {dtype:4s} B {B:3d} k {k:3d}: step {step:.3f} ms = {B / step * 1e3:9.0f} queries/s  filter kernel {f:.3f} ms  "
          f"int8 passes {ix.stat('shadow8_passes')} fallback {ix.stat('fallback_queries')}", flush=True)
    ix.close(); del ix; torch.cuda.empty_cache()
