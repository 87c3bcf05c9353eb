#!/bin/bash
# development aid: bench.py --steps 20 --warmup 5 alternating between libraries on ONE box (boxes differ by 5 %):
#   scripts/dev/cmp_libs.sh ROUNDS name=path.so [name=path.so ...]      ("cur" = the in-tree library)
rounds=$1; shift
mkdir -p gpurun_out/cmp
for i in $(seq 1 $rounds); do
  for spec in cur "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = cur ]; then unset CODD_KNN_LIB; else export CODD_KNN_LIB=$PWD/$lib; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-iters 3 > gpurun_out/cmp/${name}_$i.json 2>/dev/null
  done
done
unset CODD_KNN_LIB
python - <<PY
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/cmp/*.json")):
    try:
        j=json.load(open(f)); k=j["roofline"]["all_kernels"]; name=f.split("/")[-1].rsplit("_",1)[0]
        acc[name].append((j["value"], k["filter"]["avg_ms"], j["results_valid"]))
    except Exception as e: print(f, "ERR", e)
for n,v in acc.items():
    print(n, "q/s", [round(x[0]) for x in v], "filter ms", [round(x[1],4) for x in v], "mean", round(sum(x[1] for x in v)/len(v),4), all(x[2] for x in v))
PY
