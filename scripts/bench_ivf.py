#!/usr/bin/env python3
"""One-shard rehearsal of BASELINE config 5 (100M x 1024 fp16 over 8 GPUs -> 12.5M x 1024 fp16 per GPU):
coarse-IVF + exact scores against this engine's own flat search on the same shard.

    python scripts/bench_ivf.py [--rows 12500000 --dim 1024 --nlist 2048 --nprobe 8]

Prints one JSON object: build time, list statistics, recall@10 (IVF vs exact flat), p50 latency and
queries/s of both at B = 1, 32 and 256.  Clustered synthetic data (SURVEY.md §8d row 5 in spirit).
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=12_500_000)
    p.add_argument("--dim", type=int, default=1024)
    p.add_argument("--centres", type=int, default=4096)
    p.add_argument("--nlist", type=int, default=2048)
    p.add_argument("--nprobe", type=int, default=8)
    p.add_argument("--iters", type=int, default=6)
    p.add_argument("--dtype", default="f16")
    p.add_argument("--noise", type=float, default=0.5, help="norm of the noise added to a unit centre (cos to own centre = 1/sqrt(1+noise^2))")
    a = p.parse_args()
    import torch

    from codd_query_engine_amd import ivf
    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    dev = "cuda:0"
    gc = torch.Generator(device=dev).manual_seed(7)
    centres = torch.nn.functional.normalize(torch.randn((a.centres, a.dim), generator=gc, device=dev), dim=1)

    def draw(n, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        which = torch.randint(0, a.centres, (n,), generator=g, device=dev)
        return centres[which] + a.noise * torch.randn((n, a.dim), generator=g, device=dev) / a.dim ** 0.5

    ix = DeviceKnnIndex(a.dim, a.dtype, dev)
    ix.reserve(a.rows)
    t0 = time.perf_counter()
    for c0 in range(0, a.rows, 250_000):
        m = min(250_000, a.rows - c0)
        ix.upsert_device(c0, draw(m, 100 + c0 // 250_000).contiguous())
    torch.cuda.synchronize()
    t_ingest = time.perf_counter() - t0
    t0 = time.perf_counter()
    stats = ivf.build_ivf(ix, a.nlist, iters=a.iters)
    t_build = time.perf_counter() - t0

    q = draw(256, 9999)
    k = 10
    _, truth = ix.search_tensors(q, k)
    out = {"rows": a.rows, "dim": a.dim, "dtype": a.dtype, "ingest_s": t_ingest, "ivf_build_s": t_build, "ivf": stats, "nprobe": a.nprobe, "noise": a.noise}
    for nprobe in sorted({1, a.nprobe, 4 * a.nprobe}):
        _, got = ivf.search_ivf(ix, q, k, nprobe)
        out[f"recall@10_nprobe{nprobe}"] = (got.unsqueeze(2) == truth.unsqueeze(1)).any(dim=2).float().mean().item()

    def timed(fn, B, iters=30):
        qq = q[:B].contiguous()
        for _ in range(3):
            fn(qq)
        torch.cuda.synchronize()
        lat = []
        for _ in range(iters):
            t = time.perf_counter()
            d, r = fn(qq)
            r.cpu()
            lat.append((time.perf_counter() - t) * 1e3)
        p50 = statistics.median(lat)
        return {"p50_ms": p50, "qps": B / p50 * 1e3}

    for B in (1, 32, 256):
        out[f"ivf_B{B}"] = timed(lambda qq: ivf.search_ivf(ix, qq, k, a.nprobe), B)
        out[f"flat_B{B}"] = timed(lambda qq: ix.search_tensors(qq, k), B)
    # batch 256: the per-pair scan against the list-sharing scan (a probed list read once per 4 of its queries), two probe depths
    for nprobe in (1, a.nprobe, 4 * a.nprobe):
        for share in (0, 1):
            ix.set_option("ivf_share", share)
            out[f"ivf_B256_nprobe{nprobe}_share{share}"] = timed(lambda qq: ivf.search_ivf(ix, qq, k, nprobe), 256)
    ix.set_option("ivf_share", 1)
    elem = 4 if a.dtype == "f32" else 2
    out["bytes_flat_pass_shadow"] = a.rows * a.dim * 2
    out["bytes_ivf_per_query"] = a.nprobe * (a.rows / a.nlist) * a.dim * elem
    print(json.dumps(out))
    ix.close()


if __name__ == "__main__":
    main()
