#!/bin/bash
# development aid: bench.py alternating between this tree's library and round 2's final one on ONE box (boxes differ by 5 %).
# The other library is not kept in the tree: git worktree add /tmp/r2wt <round-2 commit> && (cd /tmp/r2wt && python -m codd_query_engine_amd.build) &&
#   cp /tmp/r2wt/codd_query_engine_amd/csrc/libcodd_knn.so codd_query_engine_amd/csrc/libcodd_knn_r2final.so   (untracked: *.so is git-ignored, but it ships with gpurun)
mkdir -p gpurun_out/r3p
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-iters 3 > gpurun_out/r3p/r3_$i.json 2>/dev/null
  CODD_KNN_LIB=$PWD/codd_query_engine_amd/csrc/libcodd_knn_r2final.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-iters 3 > gpurun_out/r3p/r2_$i.json 2>gpurun_out/r3p/r2_$i.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3p/*.json")):
    try:
        j=json.load(open(f)); k=j["roofline"]["all_kernels"]; print(f, round(j["value"]), round(j["ms_per_step"],4), {n:round(v["avg_ms"],4) for n,v in k.items()}, j["results_valid"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r3p/r2_1.err
