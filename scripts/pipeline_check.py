#!/usr/bin/env python3
"""Does issuing consecutive batches on alternating HIP streams hide the launch gaps? (development tool)

    python scripts/pipeline_check.py [rows] [depth ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    depths = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
    d, B, k = 768, 256, 10
    g = torch.Generator(device="cuda").manual_seed(1)
    ix = DeviceKnnIndex(d)
    ix.reserve(rows)
    for c0 in range(0, rows, 250_000):
        m = min(250_000, rows - c0)
        ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
    qs = [torch.randn((B, d), generator=g, device="cuda") for _ in range(4)]
    ref = [ix.search_tensors(q, k) for q in qs]
    torch.cuda.synchronize()
    n = 200
    for depth in depths:
        streams = [torch.cuda.Stream() for _ in range(depth)]
        outs = [None] * n
        for rep in range(2):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cur = torch.cuda.current_stream()
            for i in range(n):
                s = streams[i % depth]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs[i] = ix.search_tensors(qs[i % 4], k)
            for s in streams:
                cur.wait_stream(s)
            e1.record()
            torch.cuda.synchronize()
        ok = all(torch.equal(outs[i][0], ref[i % 4][0]) and torch.equal(outs[i][1], ref[i % 4][1]) for i in range(n))
        print(f"depth {depth}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/step  ok={ok}  workspaces={ix.stat('workspaces')}", flush=True)


if __name__ == "__main__":
    main()
