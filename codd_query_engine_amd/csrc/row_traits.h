// row_traits.h — how a 16-byte chunk of a stored row widens to fp32 (exact), per storage dtype.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace codd {

constexpr int DT_F32 = 0;   // == CODD_KNN_DTYPE_F32
constexpr int DT_BF16 = 1;  // == CODD_KNN_DTYPE_BF16
constexpr int DT_F16 = 2;   // == CODD_KNN_DTYPE_F16

template <int DT>
struct RowTraits;
template <>
struct RowTraits<DT_F32> {
    static constexpr int E = 4;  // elements per 16-byte chunk
    static __device__ __forceinline__ void widen(const uint4& c, float* w) {
        w[0] = __uint_as_float(c.x); w[1] = __uint_as_float(c.y);
        w[2] = __uint_as_float(c.z); w[3] = __uint_as_float(c.w);
    }
};
template <>
struct RowTraits<DT_BF16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void widen(const uint4& c, float* w) {
        w[0] = __uint_as_float(c.x << 16); w[1] = __uint_as_float(c.x & 0xffff0000u);
        w[2] = __uint_as_float(c.y << 16); w[3] = __uint_as_float(c.y & 0xffff0000u);
        w[4] = __uint_as_float(c.z << 16); w[5] = __uint_as_float(c.z & 0xffff0000u);
        w[6] = __uint_as_float(c.w << 16); w[7] = __uint_as_float(c.w & 0xffff0000u);
    }
};
template <>
struct RowTraits<DT_F16> {
    static constexpr int E = 8;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void widen(const uint4& c, float* w) {
        const h2 a = __builtin_bit_cast(h2, c.x), b = __builtin_bit_cast(h2, c.y);
        const h2 d = __builtin_bit_cast(h2, c.z), e = __builtin_bit_cast(h2, c.w);
        w[0] = (float)a[0]; w[1] = (float)a[1]; w[2] = (float)b[0]; w[3] = (float)b[1];
        w[4] = (float)d[0]; w[5] = (float)d[1]; w[6] = (float)e[0]; w[7] = (float)e[1];
    }
};

// fp32 -> storage element, round to nearest even (bit-identical to oracle/knn_oracle.c)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t f32_to_f16_rne(float f) {
    const _Float16 h = (_Float16)f;  // v_cvt_f16_f32, RNE in the default mode
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

}  // namespace codd
