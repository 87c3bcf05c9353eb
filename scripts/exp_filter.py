#!/usr/bin/env python3
"""Kernel A/B harness for the MFMA filter (development tool, not part of the product).

    python scripts/exp_filter.py build            # here: compile the variants (hipcc, no GPU)
    python scripts/exp_filter.py run [rows]       # on the GPU box: time every variant, interleaved rounds

Each variant is the same source with different -D switches (csrc/filter_gemm.h); every timing
process also checks the variant's results against the exact scan (bit-equal) unless the variant is
a diagnostic that breaks results on purpose.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "base": {},
    # add {"name": {"CODD_...": value}} entries here; CODD_EXP_INT8=1 in the environment also times the int8 filter
}


def lib_path(name):
    return os.path.join(ROOT, "codd_query_engine_amd", "csrc", f"libcodd_knn_{name}.so")


def build():
    from codd_query_engine_amd import build as b

    for name, defs in VARIANTS.items():
        print("building", name, defs, flush=True)
        b.build_variant(name, defs)


def child(rows, rounds):
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    d, B, k = 768, 256, 10
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(d)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        m = min(250_000, rows - c0)
        ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
    q = torch.randn((B, d), generator=g, device="cuda")
    ix.set_option("shadow8", 0)  # the first set of timings is the bf16 filter; CODD_EXP_INT8=1 adds the int8 one
    dist, idx = ix.search_tensors(q, k)
    torch.cuda.synchronize()
    ix.set_option("filter", 0)
    d_ref, i_ref = ix.search_tensors(q, k)
    ix.set_option("filter", 1)
    ok = bool(torch.equal(idx, i_ref) and torch.equal(dist, d_ref))
    ix.set_option("profile", rounds * 4 + 8)
    for _ in range(rounds):
        ix.search_tensors(q, k)
    torch.cuda.synchronize()
    out = {"ok": ok, "fallback": ix.stat("fallback_queries")}
    for name in ("filter", "sample", "finalize"):
        ev = ix.stat(f"events:{name}")
        out[name] = ix.stat(f"time_ns:{name}") * 1e-6 / max(ev, 1)
    if os.environ.get("CODD_EXP_INT8"):
        ix.set_option("shadow8", 1)
        ix.set_option("shadow8_max_batch", 256)
        ix.search_tensors(q, k)
        torch.cuda.synchronize()
        ix.set_option("profile", rounds * 4 + 8)
        for _ in range(rounds):
            ix.search_tensors(q, k)
        torch.cuda.synchronize()
        ev = ix.stat("events:filter")
        out["filter_int8"] = ix.stat("time_ns:filter") * 1e-6 / max(ev, 1)
        out["passes8"] = ix.stat("shadow8_passes")
    print(json.dumps(out))


def run(rows):
    names = [n for n in VARIANTS if os.path.exists(lib_path(n))]
    results = {n: [] for n in names}
    for rnd in range(3):
        for n in names:
            env = dict(os.environ, CODD_KNN_LIB=lib_path(n))
            p = subprocess.run([sys.executable, __file__, "child", str(rows), "10"], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", p.stderr[-400:], flush=True)
                continue
            results[n].append(json.loads(line[-1]))
    for n in names:
        r = results[n]
        if r:
            if "filter_int8" in r[0]:
                m8 = sorted(x["filter_int8"] for x in r)
                print(f"{n:12s} int8 filter ms min {m8[0]:.3f} med {m8[len(m8)//2]:.3f} passes8={r[0]['passes8']}", flush=True)
            ms = sorted(x["filter"] for x in r)
            print(f"{n:12s} filter ms min {ms[0]:.3f} med {ms[len(ms)//2]:.3f}  sample {r[0]['sample']:.3f} finalize {r[0]['finalize']:.3f} ok={all(x['ok'] for x in r)} fb={r[0]['fallback']}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000)
