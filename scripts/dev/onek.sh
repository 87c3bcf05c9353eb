#!/bin/bash
# scripts/dev/onek.sh MODE S3 NQB RES [extra -D...]: one instantiation of i8_tile_kernel -> /tmp/onek/k.s + its resource line
mkdir -p /tmp/onek
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -mllvm -pragma-unroll-threshold=65536 \
  -Icodd_query_engine_amd/csrc -Iinclude --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -DONEK_MODE=$1 -DONEK_S3=$2 -DONEK_NQB=$3 -DONEK_RES=$4 "${@:5}" \
  -o /tmp/onek/k_$1_$2_$3_$4.s scripts/dev/one_kernel.hip 2>&1 | grep -A12 "i8_tile" | grep -i " VGPRs:\|AGPRs\|scratch\|VGPRs Spill\|occupancy\|error" | tr '\n' ' '
echo
