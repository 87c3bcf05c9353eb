// Development aid: compiles ONE instantiation of i8_tile_kernel (seconds instead of the library's minutes) so that its ISA and
// resource usage can be read while the schedule is being edited:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=65536 -Icodd_query_engine_amd/csrc -Iinclude \
//         -DONEK_MODE=0 -DONEK_S3=3 -DONEK_NQB=16 -DONEK_RES=false --cuda-device-only -S -o /tmp/onek/k.s scripts/dev/one_kernel.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef CODD_EXPERIMENTS
#define CODD_EXPERIMENTS 0
#endif
#include "filter_gemm.h"
#include "filter_i8.h"
#ifndef ONEK_MODE
#define ONEK_MODE 0
#define ONEK_S3 3
#define ONEK_NQB 16
#define ONEK_RES false
#endif
#ifndef ONEK_F16
#define ONEK_F16 false
#endif
template __global__ void codd::i8_tile_kernel<ONEK_MODE, ONEK_S3, ONEK_NQB, ONEK_RES, ONEK_F16>(const uint4*, const uint4*, int64_t, int, int64_t, int64_t, const float*, codd::u64*,
                                                                                      codd::u64*, unsigned*, int, unsigned*, const float*, const float*, const float2*, float);
