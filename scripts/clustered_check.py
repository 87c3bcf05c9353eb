#!/usr/bin/env python3
"""Development tool: the filter leg on CLUSTERED data (dense neighbourhoods near the top of the score
distribution): hits, survivors, fallbacks and step time, checked against the exact scan."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows, d, B, k = 4_000_000, 768, 256, 10
for centres, noise in ((4096, 0.5), (4096, 0.2), (256, 0.3), (64, 0.1)):
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
    ix.set_option("shadow8", int(os.environ.get("CODD_SHADOW8", "1")))
    for _ in range(4):  # (the engine reads its counters back asynchronously: let it settle on a filter, and build the lazy shadows, before timing)
        ix.search_tensors(q, k)
        torch.cuda.synchronize()
    h0, s0, f0 = ix.stat("filter_hits"), ix.stat("filter_survivors"), ix.stat("fallback_queries")
    t0 = time.perf_counter()
    for _ in range(5):
        dd, rr = ix.search_tensors(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    hits, surv, fb = (ix.stat("filter_hits") - h0) / 5 / B, (ix.stat("filter_survivors") - s0) / 5 / B, (ix.stat("fallback_queries") - f0) / 5
    ix.set_option("filter", 0)
    de, re_ = ix.search_tensors(q[:32], k)
    ok = bool(torch.equal(rr[:32], re_) and torch.equal(dd[:32], de))
    print(f"centres {centres:5d} noise {noise}: {dt:8.3f} ms/step  hits/query {hits:8.0f}  survivors/query {surv:7.1f}  fallback queries/step {fb:5.1f}  exact={ok}", flush=True)
    ix.close()
