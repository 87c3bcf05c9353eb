// wave_topk.h — wave64 primitives shared by every kernel of the path: packed keys, the
// canonical butterfly, and a top-k list kept sorted ACROSS the lanes of one wavefront.
// gfx950 only: wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace codd {

typedef unsigned long long u64;
constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---- packed keys: larger = better (higher score, then LOWER row). 0 = empty. ------------
__device__ __forceinline__ uint32_t ord_f32(float s) {
    if (s != s) s = -INFINITY;  // NaN ranks as -inf
    const uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, uint32_t row) {
    return ((u64)ord_f32(score) << 32) | (u64)(0xffffffffu - row);
}
__device__ __forceinline__ float key_score(u64 key) { return unord_f32((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t key_row(u64 key) { return 0xffffffffu - (uint32_t)(key & 0xffffffffull); }

// ---- cross-lane moves of 64-bit values -----------------------------------------------------
__device__ __forceinline__ u64 shfl_up1_u64(u64 v) {
    const uint32_t lo = __shfl_up((uint32_t)v, 1), hi = __shfl_up((uint32_t)(v >> 32), 1);
    return ((u64)hi << 32) | lo;
}
// src must be wave-uniform
__device__ __forceinline__ u64 readlane_u64(u64 v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// ---- canonical butterfly: strides 32,16,8,4,2,1, then +0.0f -----------------
__device__ __forceinline__ float butterfly_sum(float a) {
    a += __shfl_xor(a, 32);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 8);
    a += __shfl_xor(a, 4);
    a += __shfl_xor(a, 2);
    a += __shfl_xor(a, 1);
    return a + 0.0f;
}

// Same tree for FOUR rows at once (7 cross-lane moves instead of 24): on return every lane of
// the 16-lane group g = lane>>4 holds the canonical sum of row g.  Bitwise identical to
// butterfly_sum() per row because IEEE addition commutes and the pairing per level is the same.
__device__ __forceinline__ float butterfly_sum4(float a0, float a1, float a2, float a3, int lane) {
    const bool hi = (lane & 32) != 0;
    const float s0 = hi ? a0 : a2, s1 = hi ? a1 : a3;  // sent to lane^32
    const float k0 = hi ? a2 : a0, k1 = hi ? a3 : a1;  // kept
    const float x0 = k0 + __shfl_xor(s0, 32);
    const float x1 = k1 + __shfl_xor(s1, 32);
    const bool h2 = (lane & 16) != 0;
    const float s = h2 ? x0 : x1, kp = h2 ? x1 : x0;
    float y = kp + __shfl_xor(s, 16);
    y += __shfl_xor(y, 8);
    y += __shfl_xor(y, 4);
    y += __shfl_xor(y, 2);
    y += __shfl_xor(y, 1);
    return y + 0.0f;
}

// ---- top-k list distributed over the lanes of one wave -------------------------------------
// rank r lives in slot r/64 of lane r%64, descending.  SLOTS = 1 (k <= 64) or 2 (k <= 128).
// `thr` (the key at rank k-1) is wave-uniform: a candidate enters only if key > thr.
template <int SLOTS>
struct WaveTopK {
    u64 v[SLOTS];
    u64 thr;

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) v[s] = 0ull;
        thr = 0ull;
    }

    // key: wave-uniform, strictly greater than thr.  k: wave-uniform.
    __device__ __forceinline__ void insert(u64 key, int k, int lane) {
        int pos = __popcll(__ballot(v[0] > key));
        if (SLOTS == 2) pos += __popcll(__ballot(v[1] > key));
        const u64 up0 = shfl_up1_u64(v[0]);
        if (SLOTS == 2) {
            const u64 carry = readlane_u64(v[0], 63);
            const u64 up1 = shfl_up1_u64(v[1]);
            if (pos < 64) {
                v[1] = (lane == 0) ? carry : up1;
                v[0] = (lane < pos) ? v[0] : ((lane == pos) ? key : up0);
            } else {
                const int p1 = pos - 64;
                v[1] = (lane < p1) ? v[1] : ((lane == p1) ? key : up1);
            }
        } else {
            v[0] = (lane < pos) ? v[0] : ((lane == pos) ? key : up0);
        }
        const int kr = k - 1;
        thr = (SLOTS == 2 && kr >= 64) ? readlane_u64(v[1], kr - 64) : readlane_u64(v[0], kr);
    }

    // offer one wave-uniform candidate
    __device__ __forceinline__ void offer(u64 key, int k, int lane) {
        if (key > thr) insert(key, k, lane);
    }

    // offer 64 lane-private candidates (0 = none)
    __device__ __forceinline__ void offer_lanes(u64 cand, int k, int lane) {
        u64 mask = __ballot(cand > thr);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const u64 key = readlane_u64(cand, src);
            if (key > thr) insert(key, k, lane);
        }
    }

    // key at rank r (r = slot*64 + lane) for this lane
    __device__ __forceinline__ u64 at_slot(int s) const { return v[s]; }
};

}  // namespace codd
