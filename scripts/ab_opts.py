#!/usr/bin/env python3
"""Interleaved in-process A/B of engine OPTION SETS on one index (development tool; guide rule 24).

    python scripts/ab_opts.py ROWS [--dim D] [--batch B] [--rounds R] "i8_pair=0" "i8_pair=1" "i8_pair=1,sample_div8=30" ...

Every option set is a comma-separated list of key=value (codd_knn_set_option).  Per round and set: 10 searches; HIP-event times
of the filter / sample / finalize launches and the step; hits and survivors per query; every set is checked once against the
exact scan (ids and distances bit-equal)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("rows", type=int)
    p.add_argument("sets", nargs="+")
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--rounds", type=int, default=5)
    a = p.parse_args()
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    k = 10
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(a.dim)
    ix.reserve(a.rows)
    for c0 in range(0, a.rows, 250_000):
        m = min(250_000, a.rows - c0)
        ix.upsert_device(c0, torch.randn((m, a.dim), generator=g, device="cuda"))
    qs = [torch.randn((a.batch, a.dim), generator=g, device="cuda") for _ in range(4)]
    ix.set_option("shadow8_cooldown", 0)
    ix.set_option("filter", 0)
    ref = ix.search_tensors(qs[0], k)
    ix.set_option("filter", 1)
    sets = [[kv.split("=") for kv in s.split(",") if kv] for s in a.sets]

    def apply(s):
        for key, val in s:
            ix.set_option(key, int(val))

    out = {"rows": a.rows, "dim": a.dim, "batch": a.batch, "sets": {}}
    res = {i: [] for i in range(len(sets))}
    for i, s in enumerate(sets):
        apply(s)
        dist, idx = ix.search_tensors(qs[0], k)
        torch.cuda.synchronize()
        out["sets"][a.sets[i]] = {"exact": bool(torch.equal(idx, ref[1]) and torch.equal(dist, ref[0]))}
    for rnd in range(a.rounds):
        for i, s in enumerate(sets):
            apply(s)
            ix.set_option("profile", 64)
            h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for j in range(10):
                ix.search_tensors(qs[j % len(qs)], k)
            t1.record()
            torch.cuda.synchronize()
            r = {"step_ms": t0.elapsed_time(t1) / 10}
            for name in ("filter", "sample", "finalize"):
                r[name + "_ms"] = ix.stat(f"time_ns:{name}") * 1e-6 / max(ix.stat(f"events:{name}"), 1)
            r["hits_q"] = (ix.stat("filter_hits") - h0) / (10 * a.batch)
            r["surv_q"] = (ix.stat("filter_survivors") - s0) / (10 * a.batch)
            res[i].append(r)
    for i in range(len(sets)):
        d = out["sets"][a.sets[i]]
        for key in ("step_ms", "filter_ms", "sample_ms", "finalize_ms"):
            v = sorted(r[key] for r in res[i])
            d[key] = {"min": round(v[0], 4), "med": round(v[len(v) // 2], 4)}
        d["hits_q"] = round(res[i][-1]["hits_q"], 1)
        d["surv_q"] = round(res[i][-1]["surv_q"], 1)
    out["fallback"] = ix.stat("fallback_queries")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
