#!/usr/bin/env python3
"""Development tool: per-step counters of the 256-centre clustered case (which filter ran, hits, time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows, d, B, k = 4_000_000, 768, 256, 10
centres, noise = int(sys.argv[1]) if len(sys.argv) > 1 else 256, float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
gc = torch.Generator(device="cuda").manual_seed(7)
c = torch.nn.functional.normalize(torch.randn((centres, d), generator=gc, device="cuda"), dim=1)
g = torch.Generator(device="cuda").manual_seed(11)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    m = min(250_000, rows - c0)
    which = torch.randint(0, centres, (m,), generator=g, device="cuda")
    ix.upsert_device(c0, (c[which] + noise * torch.randn((m, d), generator=g, device="cuda") / d ** 0.5).contiguous())
which = torch.randint(0, centres, (B,), generator=g, device="cuda")
q = c[which] + noise * torch.randn((B, d), generator=g, device="cuda") / d ** 0.5
names = ["filter_passes", "shadow8_passes", "i8v2_passes", "shadow8_cooldowns", "shadow8_builds", "shadow16_builds", "filter_hits", "filter_survivors", "fallback_queries", "shadow8_eps_r_micro"]
prev = {n: ix.stat(n) for n in names}
for step in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.search_tensors(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    cur = {n: ix.stat(n) for n in names}
    print(f"step {step}: {dt:7.3f} ms  " + "  ".join(f"{n}={cur[n] - prev[n]}" for n in names if n != "shadow8_eps_r_micro") + f"  eps_r_micro={cur['shadow8_eps_r_micro']}", flush=True)
    prev = cur
