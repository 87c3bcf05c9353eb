#!/usr/bin/env python3
"""A/B of the two int8 filter kernels in ONE process, interleaved rounds (development tool).

    python scripts/ab_i8.py [rows] [dim] [batch]

Per round and per kernel (i8v2 = 0: gemm_filter_kernel<., 8, int8>; 1: i8_tile_kernel): HIP-event time of the filter and
sample launches, hits and survivors; results are compared with the exact scan once."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    k = 10
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(d)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        m = min(250_000, rows - c0)
        ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
    q = torch.randn((B, d), generator=g, device="cuda")
    ix.set_option("shadow8_cooldown", 0)
    ix.set_option("filter", 0)
    d_ref, i_ref = ix.search_tensors(q, k)
    ix.set_option("filter", 1)
    out = {}
    for v in (0, 1):
        ix.set_option("i8v2", 2 * v)
        dist, idx = ix.search_tensors(q, k)
        torch.cuda.synchronize()
        out[f"ok{v}"] = bool(torch.equal(idx, i_ref) and torch.equal(dist, d_ref))
    res = {0: [], 1: []}
    for rnd in range(5):
        for v in (0, 1):
            ix.set_option("i8v2", 2 * v)  # 0: first-generation kernels; 2: the tile kernel wherever it applies
            ix.set_option("profile", 64)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                ix.search_tensors(q, k)
            t1.record()
            torch.cuda.synchronize()
            res[v].append({
                "step_ms": t0.elapsed_time(t1) / 10,
                "filter_ms": ix.stat("time_ns:filter") * 1e-6 / max(ix.stat("events:filter"), 1),
                "sample_ms": ix.stat("time_ns:sample") * 1e-6 / max(ix.stat("events:sample"), 1),
                "finalize_ms": ix.stat("time_ns:finalize") * 1e-6 / max(ix.stat("events:finalize"), 1),
            })
    for v in (0, 1):
        f = sorted(r["filter_ms"] for r in res[v]); s = sorted(r["sample_ms"] for r in res[v]); st = sorted(r["step_ms"] for r in res[v])
        out[f"v{v}"] = {"filter_ms_min": round(f[0], 4), "filter_ms_med": round(f[len(f) // 2], 4), "sample_ms_med": round(s[len(s) // 2], 4),
                        "step_ms_med": round(st[len(st) // 2], 4)}
    out["rows"], out["dim"], out["batch"] = rows, d, B
    out["fallback"] = ix.stat("fallback_queries")
    out["GBps_v1"] = round(rows * ((d + 127) // 128 * 128) / out["v1"]["filter_ms_med"] / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
