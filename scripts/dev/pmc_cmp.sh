#!/bin/bash
# development aid: the same counter pass over the headline bench with two libraries (scripts/dev/pmc_cmp.sh OTHER_LIB.so)
out=gpurun_out/pmc_cmp; mkdir -p $out; export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --latency-iters 3"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace -d $out/new --output-format csv -- $B > $out/new.log 2>&1
export CODD_KNN_LIB=$1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace -d $out/old --output-format csv -- $B > $out/old.log 2>&1
unset CODD_KNN_LIB
python3 scripts/pmc_summary.py $out/new | grep "i8_tile_kernel<0" > $out/summary.txt
echo ---- >> $out/summary.txt
python3 scripts/pmc_summary.py $out/old | grep "i8_tile_kernel<0" >> $out/summary.txt
cat $out/summary.txt
