#!/usr/bin/env python3
"""Is the int8 tile kernel power-limited?  (development tool; run with a no-hits experiment build:
    CODD_KNN_LIB=.../libcodd_knn_nohits.so python scripts/power_check.py [rows])
Times the filter launch on a random corpus and on corpora whose int8 bytes barely toggle (one-hot rows; sparse queries):
same instruction stream, same bytes moved, different switching activity in the matrix pipe."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d, B, k = 768, 256, 10
g = torch.Generator(device="cuda").manual_seed(1)
for corpus, queries in (("random", "random"), ("one-hot", "random"), ("random", "one-hot"), ("one-hot", "one-hot"), ("random", "random")):
    ix = DeviceKnnIndex(d)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        m = min(250_000, rows - c0)
        if corpus == "random":
            x = torch.randn((m, d), generator=g, device="cuda")
        else:
            x = torch.zeros((m, d), device="cuda")
            x[torch.arange(m, device="cuda"), torch.randint(0, d, (m,), generator=g, device="cuda")] = 1.0
        ix.upsert_device(c0, x)
    if queries == "random":
        q = torch.randn((B, d), generator=g, device="cuda")
    else:
        q = torch.zeros((B, d), device="cuda")
        q[torch.arange(B, device="cuda"), torch.randint(0, d, (B,), generator=g, device="cuda")] = 1.0
    ix.set_option("shadow8_cooldown", 0)
    ix.set_option("i8v2", 1)
    for _ in range(3):
        ix.search_tensors(q, k)
    torch.cuda.synchronize()
    best = []
    for rnd in range(5):
        ix.set_option("profile", 64)
        for _ in range(10):
            ix.search_tensors(q, k)
        torch.cuda.synchronize()
        best.append(ix.stat("time_ns:filter") * 1e-6 / max(ix.stat("events:filter"), 1))
    best.sort()
    print(f"corpus {corpus:8s} queries {queries:8s}: filter launch {best[0]:.4f} (min) {best[2]:.4f} (median) ms, i8v2 passes {ix.stat('i8v2_passes')}", flush=True)
    ix.close()
