// Development tool: sustained issue rate of the int8 MFMA shapes on gfx950 (2 waves per SIMD, every CU busy, independent
// accumulators, operands in registers): what the matrix pipe delivers under the chip's power limit with nothing else going on.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const int* __restrict__ seed, int* __restrict__ out, int iters) {
    i32x4 a, b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = seed[(threadIdx.x * 4 + i) & 1023];
        for (int j = 0; j < 4; ++j) b[j][i] = seed[(threadIdx.x * 16 + j * 4 + i + 77) & 1023];
    }
    if (SHAPE == 16) {
        i32x4 c[32];
        for (int i = 0; i < 32; ++i) c[i] = i32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[i & 3], c[i], 0, 0, 0);
        }
        i32x4 s = c[0];
        for (int i = 1; i < 32; ++i) s += c[i];
        out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
        i32x16 c[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[i & 3], c[i], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;
    int *seed, *out;
    hipMalloc(&seed, 4096);
    hipMalloc(&out, 512 * 512 * 4);
    int h[1024];
    srand(1);
    // operand bytes: 0 = uniform random, 1 = zero, N > 1 = Gaussian with standard deviation N (clamped to +-127): what a
    // quantised unit vector looks like (sigma ~ 33 at 768 dimensions)
    for (int i = 0; i < 1024; ++i) {
        if (zero == 0) h[i] = (int)((unsigned)rand() * 2654435761u);
        else if (zero == 1) h[i] = 0;
        else {
            unsigned w = 0;
            for (int b = 0; b < 4; ++b) {
                double u = 0;
                for (int t = 0; t < 12; ++t) u += rand() / (double)RAND_MAX;
                int v = (int)((u - 6.0) * zero);
                v = v > 127 ? 127 : (v < -127 ? -127 : v);
                w |= (unsigned)(v & 255) << (8 * b);
            }
            h[i] = (int)w;
        }
    }
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int shape : {16, 32, 16}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (shape == 16) k<16><<<512, 512>>>(seed, out, iters);
            else k<32><<<512, 512>>>(seed, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: 32 x (16x16x64x2) = 16 x (32x32x32x2) = 1,048,576 ops
            const double ops = 512.0 * 8 * iters * 1048576.0;
            printf("shape %dx: %8.3f ms  %7.1f TOP/s  (%s operands)\n", shape, ms, ops / ms * 1e-9, zero == 1 ? "zero" : (zero ? "gaussian" : "uniform"));
        }
    }
    return 0;
}
