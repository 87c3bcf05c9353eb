#!/bin/bash
# rocprofv3 counter passes over the headline bench (development tool).  usage: scripts/pmc_round.sh <tag>
# Each pass is its own run with --kernel-trace only (gpurun refuses --pmc together with the tracing domains).
set -e
tag=${1:-r3}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --latency-iters 3"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $out/p1 --output-format csv -- $B > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA --kernel-trace -d $out/p2 --output-format csv -- $B > $out/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace -d $out/p3 --output-format csv -- $B > $out/p3.log 2>&1
python3 scripts/pmc_summary.py $out/p1 $out/p2 $out/p3 > $out/summary.txt
