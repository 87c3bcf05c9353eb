#!/usr/bin/env python3
"""Per-kernel time of one B=256 search step as a function of the shard's row count (development tool).

    python scripts/rows_sweep.py [dim] [batch]

What a rank sees when the 10M-row corpus is split over 1/2/4/8/16 GPUs: filter / sample / finalize kernel
averages from the library's HIP-event counters, the whole step (HIP events around 20 back-to-back searches) and the
shadow bytes per second of the filter kernel.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    k = 10
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((B, d), generator=g, device="cuda")
    print(f"{'rows':>10s} {'step ms':>8s} {'filter':>8s} {'sample':>8s} {'final':>8s} {'scan':>8s} {'TB/s':>6s} {'hits/q':>8s} {'surv/q':>7s} fb", flush=True)
    for rows in (312_500, 625_000, 1_250_000, 2_500_000, 5_000_000, 10_000_000):
        ix = DeviceKnnIndex(d)
        ix.reserve(rows)
        for c0 in range(0, rows, 250_000):
            m = min(250_000, rows - c0)
            ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
        for _ in range(3):
            ix.search_tensors(q, k)
        torch.cuda.synchronize()
        n = 20
        h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
        ix.set_option("profile", n * 6 + 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ix.search_tensors(q, k)
        e1.record()
        torch.cuda.synchronize()
        step = e0.elapsed_time(e1) / n
        t = {}
        for name in ("filter", "sample", "finalize", "scan"):
            ev = ix.stat(f"events:{name}")
            t[name] = ix.stat(f"time_ns:{name}") * 1e-6 / max(ev, 1)
        hits = (ix.stat("filter_hits") - h0) / n / B
        surv = (ix.stat("filter_survivors") - s0) / n / B
        tbs = rows * d * 2 / (t["filter"] * 1e-3) / 1e12 if t["filter"] else 0.0
        print(f"{rows:10d} {step:8.3f} {t['filter']:8.3f} {t['sample']:8.3f} {t['finalize']:8.3f} {t['scan']:8.3f} {tbs:6.2f} {hits:8.1f} {surv:7.1f} {ix.stat('fallback_queries')}", flush=True)
        ix.close()
        del ix
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
