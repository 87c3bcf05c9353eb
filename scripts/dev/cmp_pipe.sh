#!/bin/bash
# development aid: the headline bench with 1, 2 and 3 batches in flight, alternating (scripts/dev/cmp_pipe.sh)
mkdir -p gpurun_out/pipe
for i in 1 2; do for d in 1 2 3; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-iters 3 --pipeline $d > gpurun_out/pipe/d${d}_$i.json 2>/dev/null
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/pipe/*.json")):
    try:
        j=json.load(open(f)); k=j["roofline"]["all_kernels"]; print(f, round(j["value"]), round(j["ms_per_step"],4), {n:round(v["avg_ms"],4) for n,v in k.items()}, j["results_valid"])
    except Exception as e: print(f, "ERR", e)
PY
