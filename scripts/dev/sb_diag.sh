#!/bin/bash
# development aid: where small_batch_kernel's time goes — lat_small.py with the diagnostic variants (-DCODD_SB_EXP_*; results of those are garbage)
mkdir -p gpurun_out/sbdiag
for v in cur sb_nooffer sb_nostream sb_nofinal; do
  if [ $v = cur ]; then unset CODD_KNN_LIB; else export CODD_KNN_LIB=$PWD/codd_query_engine_amd/csrc/libcodd_knn_$v.so; fi
  for rows in 1000000 250000; do
    echo "$v $rows $(python scripts/lat_small.py $rows 2>/dev/null | tail -1)" | tee -a gpurun_out/sbdiag/out.txt
  done
done
