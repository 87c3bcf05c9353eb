#!/usr/bin/env python3
"""Randomised soak: the MFMA filter leg against the exact scan (both on the GPU), bit for bit, over random
shapes, dtypes, k and data kinds for a wall-clock budget.  Development tool; prints one line per failure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from codd_query_engine_amd.knn_index import DeviceKnnIndex

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tile_only = len(sys.argv) > 3 and sys.argv[3] == "tile"  # shapes that take i8_tile_kernel only (65..256 queries, rows of >= 384 elements)
t_end, cases, fails, fallbacks, passes8 = time.time() + budget, 0, 0, 0, 0
t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:  # (a silent run looks hung to the job runner)
        print(f"... {cases} cases, {fails} mismatches so far", flush=True)
        t_note = time.time() + 60
    d = int(rng.choice([64, 128, 192, 256, 320, 384, 512, 768, 1024]))
    dtype = str(rng.choice(["f32", "bf16", "f16"]))
    n = int(rng.integers(6_000, 250_000))
    B = int(rng.choice([1, 2, 7, 8, 9, 31, 32, 33, 64, 100, 128, 129, 255, 256, 257, 300]))
    k = int(rng.choice([1, 5, 10, 10, 10, 33, 64, 65, 100]))
    if tile_only:
        d = int(rng.choice([384, 512, 640, 768, 768, 896, 1000, 1024, 1152, 1536, 2048]))
        if d > 1024:
            dtype = str(rng.choice(["bf16", "f16"]))  # (f32 rows go up to 1024 elements)
        B = int(rng.integers(65, 257))
        n = int(rng.integers(6_000, 400_000))
    kind = str(rng.choice(["random", "clustered", "dupes"]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, d), generator=g, device="cuda")
    if kind == "clustered":
        c = torch.randn((int(rng.integers(4, 200)), d), generator=g, device="cuda")
        x = c[torch.randint(0, c.shape[0], (n,), generator=g, device="cuda")] + float(rng.choice([0.05, 0.3, 1.0])) * x
    elif kind == "dupes":
        x[torch.randint(0, n, (n // 3,), generator=g, device="cuda")] = x[int(rng.integers(n))].clone()
    q = torch.randn((B, d), generator=g, device="cuda")
    if kind != "random":
        q[: B // 2] = x[torch.randint(0, n, (B // 2,), generator=g, device="cuda")] + 0.01 * q[: B // 2]
    ix = DeviceKnnIndex(d, dtype)
    ix.upsert_device(0, x.contiguous())
    for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
        ix.set_option(key, 1)
    ix.set_option("shadow8_max_batch", 256 if tile_only else int(rng.choice([8, 64, 256, 256])))  # which batches take the int8 filter
    ix.set_option("i8v2", 2 if tile_only else int(rng.choice([1, 2])))       # the tile kernel from 640 / from 384 elements on
    if tile_only:
        ix.set_option("shadow8_cooldown", 0)
    if rng.random() < 0.3:  # an overwrite and an append after a first search: the int8 shadow must follow
        ix.search_tensors(q[:1], 1)
        x2 = torch.randn((int(rng.integers(1, 3000)), d), generator=g, device="cuda")
        ix.upsert_device(int(rng.integers(0, n)), q[:1].contiguous())
        ix.upsert_device(n, x2)
    df, rf = ix.search_tensors(q, k)
    if rng.random() < 0.5:  # the same index searched again right away (stale LDS / DMA / counter state must not leak)
        q = torch.randn((B, d), generator=g, device="cuda")
        df, rf = ix.search_tensors(q, k)
    # the same search twice more: the candidate lists (their SIZE: order is up to the atomics) must not change between identical searches — a race in
    # the filter kernels' schedule shows there long before it costs a true neighbour
    hh = []
    for _ in range(2):
        h0 = ix.stat("filter_hits")
        ix.search_tensors(q, k)
        hh.append(ix.stat("filter_hits") - h0)
    # (only where nothing legitimately changes between two searches: candidate lists that overflow are cut off at a point the atomics' order decides, and an index
    #  that cools down from the int8 filter to the 2-byte one after measuring its survivors takes another path the second time)
    if hh[0] != hh[1] and (kind == "random" or tile_only):
        fails += 1
        print(f"NON-DETERMINISTIC candidate lists n={n} d={d} dtype={dtype} B={B} k={k} kind={kind}: {hh}", flush=True)
    used_filter = ix.stat("filter_passes") > 0
    tile_passes = globals().get("tile_passes", 0) + ix.stat("i8v2_passes")
    f16_passes = globals().get("f16_passes", 0) + ix.stat("f16_tile_passes")
    fallbacks += ix.stat("fallback_queries")
    passes8 += ix.stat("shadow8_passes")
    ix.set_option("filter", 0)
    de, re_ = ix.search_tensors(q, k)
    ok = bool(torch.equal(rf, re_) and torch.equal(df, de))
    cases += 1
    if not ok:
        fails += 1
        bad = (rf != re_).any(dim=1).nonzero().flatten().tolist()[:5]
        print(f"MISMATCH n={n} d={d} dtype={dtype} B={B} k={k} kind={kind} filter={used_filter} queries={bad}", flush=True)
    ix.close()
print(f"soak: {cases} cases, {fails} mismatches, {fallbacks} fallback queries, {passes8} int8 passes ({globals().get('tile_passes', 0)} through the tile kernel), {globals().get('f16_passes', 0)} fp16 tile passes in {budget:.0f} s")
sys.exit(1 if fails else 0)
