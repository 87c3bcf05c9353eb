#!/usr/bin/env python3
"""Does a search capture into a HIP graph, and what does replay save at B = 1? (development tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    d, k = 768, 10
    for rows in (100_000, 1_000_000, 10_000_000):
        g = torch.Generator(device="cuda").manual_seed(1)
        ix = DeviceKnnIndex(d)
        ix.reserve(rows)
        for c0 in range(0, rows, 250_000):
            ix.upsert_device(c0, torch.randn((min(250_000, rows - c0), d), generator=g, device="cuda"))
        for B in (1, 256):
            q = torch.randn((B, d), generator=g, device="cuda")
            ref = ix.search_tensors(q, k)
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            static_q = q.clone()
            with torch.cuda.stream(s):
                for _ in range(3):
                    ix.search_tensors(static_q, k)
            s.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                out = ix.search_tensors(static_q, k)
            torch.cuda.synchronize()

            def timed(fn, n=200):
                lat = []
                for _ in range(n):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    o = fn()
                    o[1].cpu()
                    lat.append(time.perf_counter() - t0)
                lat.sort()
                return lat[len(lat) // 2] * 1e3

            def eager():
                return ix.search_tensors(q, k)

            def replay():
                static_q.copy_(q)
                graph.replay()
                return out

            t_e, t_g = timed(eager), timed(replay)
            ok = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            print(f"rows {rows:9d} B {B:3d}: eager p50 {t_e:.3f} ms   graph p50 {t_g:.3f} ms   ok={ok}", flush=True)
        ix.close()


if __name__ == "__main__":
    main()
