#!/usr/bin/env python3
"""bench.py over the BASELINE.json configurations that fit one MI355X, one JSON line each (run on the GPU box).

    python scripts/bench_configs.py > gpurun_out/configs.jsonl

configs[1]: 1M x 768 f32, B = 1 (p50 latency is the figure of merit);  configs[2]: 10M x 768 f32, B = 256 (the headline);
configs[3]: what ONE of 8 ranks holds of the 10M x 768 bf16 corpus (1.25M rows), B = 1 and B = 256, through the RCCL code path.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ("configs[1] 1M x 768 f32, B=1", ["--rows", "1000000", "--batch", "1", "--steps", "50"]),
    ("configs[2] 10M x 768 f32, B=256", ["--steps", "10"]),
    ("configs[2] 10M x 768 f32, B=1", ["--batch", "1", "--steps", "20"]),
    ("configs[3] one of 8 shards: 1.25M x 768 bf16, B=256", ["--rows", "1250000", "--dtype", "bf16", "--force-dist", "--steps", "50"]),
    ("configs[3] one of 8 shards: 1.25M x 768 bf16, B=1", ["--rows", "1250000", "--dtype", "bf16", "--batch", "1", "--force-dist", "--steps", "50"]),
]

for name, extra in RUNS:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--warmup", "3", *extra], capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        print(json.dumps({"run": name, "failed": p.stderr[-500:]}), flush=True)
        continue
    line = json.loads(lines[-1])
    r = line["roofline"]
    print(json.dumps({"run": name, "qps": line["value"], "ms_per_step": line["ms_per_step"], "p50_latency_ms_batch1": line["p50_latency_ms_batch1"],
                      "results_valid": line["results_valid"], "dominant_kernel": r["kernel"], "kernel_avg_ms": r["avg_launch_ms"],
                      "achieved_GBps": r["achieved"], "frac_of_8TBps": r["frac"], "batches_in_flight": line["config"].get("batches_in_flight"),
                      "all_kernels": r["all_kernels"]}), flush=True)
