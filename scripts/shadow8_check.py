#!/usr/bin/env python3
"""int8 shadow vs bf16 shadow for small batches: p50 latency, per-kernel time, candidates (development tool).

    python scripts/shadow8_check.py [rows] [dim]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    k = 10
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(d)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        ix.upsert_device(c0, torch.randn((min(250_000, rows - c0), d), generator=g, device="cuda"))
    torch.cuda.synchronize()
    for B in (1, 8, 32, 64):
        q = torch.randn((B, d), generator=g, device="cuda")
        res = {}
        for mode in (0, 1):
            ix.set_option("shadow8", mode)
            ix.set_option("shadow8_max_batch", 64)
            t0 = time.perf_counter()
            out = ix.search_tensors(q, k)
            torch.cuda.synchronize()
            first = (time.perf_counter() - t0) * 1e3
            h0, s0, f0 = ix.stat("filter_hits"), ix.stat("filter_survivors"), ix.stat("fallback_queries")
            n = 30
            ix.set_option("profile", n * 6 + 8)
            lat = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dd, rr = ix.search_tensors(q, k)
                rr.cpu()
                lat.append((time.perf_counter() - t0) * 1e3)
            lat.sort()
            kt = {name: ix.stat(f"time_ns:{name}") * 1e-6 / max(ix.stat(f"events:{name}"), 1) for name in ("filter", "sample", "finalize")}
            ix.set_option("profile", 0)
            res[mode] = out
            print(f"B {B:3d} shadow8={mode}: p50 {lat[len(lat) // 2]:.3f} ms (first call {first:.1f} ms)  filter {kt['filter']:.3f} sample {kt['sample']:.3f} "
                  f"finalize {kt['finalize']:.3f}  hits/q {(ix.stat('filter_hits') - h0) / n / B:.0f} surv/q {(ix.stat('filter_survivors') - s0) / n / B:.1f} "
                  f"fb {ix.stat('fallback_queries') - f0}", flush=True)
        print("   equal:", bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
