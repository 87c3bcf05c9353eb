#!/bin/bash
# The round's profile artefacts in one GPU call (development tool).  usage: scripts/profile_round.sh <tag>
# Writes under gpurun_out/prof_<tag>/; the summaries are then copied by hand into profiles/<round>/.
set -e
tag=${1:-r3}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# 1. the headline line with the CPU comparators beside it
python3 bench.py > $out/bench_full.json 2> $out/bench_full.err
# 2. the same command under rocprofv3 (kernel trace + stats): the per-kernel durations the bench line has to agree with
rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- python3 bench.py --no-cpu-baseline > $out/bench_paired.json 2> $out/bench_paired.err
# 3. counter passes (each its own run)
scripts/pmc_round.sh $tag
# 4. what one of 8 ranks sees: 1.25M rows through the RCCL code path, one stream
rocprofv3 --kernel-trace -d $out/dist --output-format csv -- python3 bench.py --rows 1250000 --force-dist --steps 40 --warmup 5 --no-cpu-baseline --pipeline 1 > $out/bench_dist.json 2> $out/bench_dist.err
python3 scripts/gap_report.py $out/dist > $out/step_timeline_1p25m.txt
python3 bench.py --rows 1250000 --force-dist --steps 40 --warmup 5 --no-cpu-baseline --pipeline 1 > $out/bench_dist_plain.json 2> $out/bench_dist_plain.err
# 5. clustered corpora
python3 scripts/clustered_check.py > $out/clustered.txt 2>&1
echo done
