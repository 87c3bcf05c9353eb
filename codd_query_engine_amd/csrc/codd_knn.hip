// codd_knn.hip — HIP kernels (gfx950 / CDNA4, wave64) and the C ABI of include/codd_knn.h.
//
// What runs here is the arithmetic half of ChromaDB on Codd's search_relevant_metrics path
// (reference call sites: codd_dal/metrics/metrics_semantic_metadata_store.py:60-69 create,
// :236-238 upsert, :314-316 query; scoring :336).  Kernels:
//
//   normalize_rows_kernel   ingest + query prep: c <- c/|c| (canonical sum of squares, IEEE sqrt/div)
//   scan_topk_kernel        exact streaming scan: one wave owns 4 rows per step, 16-B/lane coalesced
//                           loads, fmaf chains in canonical order, wave-distributed top-k lists
//   merge_keys_kernel       integer top-k of packed keys (per-block partials, shard partials)
//
// HBM-bound byte streaming: no LDS staging of the corpus (each row is consumed by exactly one
// wave, so an LDS round trip would be pure overhead — guide §5 "GEMV / M <= 16" row), many
// 16-B loads in flight per lane, results leave as 8-byte keys.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

#include "codd_knn.h"
#include "wave_topk.h"

using namespace codd;

// =============================================================================================
// device code
// =============================================================================================

namespace {

constexpr int DT_F32 = CODD_KNN_DTYPE_F32;
constexpr int DT_BF16 = CODD_KNN_DTYPE_BF16;
constexpr int DT_F16 = CODD_KNN_DTYPE_F16;

template <int DT>
struct RowTraits;
template <>
struct RowTraits<DT_F32> {
    static constexpr int E = 4;      // elements per 16-byte chunk
    static constexpr int ESIZE = 4;  // bytes per element
    static __device__ __forceinline__ void widen(const uint4& c, float* w) {
        w[0] = __uint_as_float(c.x); w[1] = __uint_as_float(c.y);
        w[2] = __uint_as_float(c.z); w[3] = __uint_as_float(c.w);
    }
};
template <>
struct RowTraits<DT_BF16> {
    static constexpr int E = 8;
    static constexpr int ESIZE = 2;
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
    static constexpr int ESIZE = 2;
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

// ---------------------------------------------------------------------------------------------
// normalize_rows_kernel: one wave per input vector.  in: n x d fp32 (row stride d).
// out row = slots ? slots[r] : first_slot + r, width dpad, storage dtype DT.
// Sum of squares in the canonical order with E = 4 (the input is fp32), then IEEE sqrt and
// IEEE division per element; a zero / non-finite norm stores an all-zero row.
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, int64_t n, int d, int dpad,
                                                             int normalize, const int64_t* __restrict__ slots,
                                                             int64_t first_slot, void* __restrict__ out_) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* x = in + r * (int64_t)d;
    const int nch = dpad >> 2;
    float scale_div = 1.0f;
    bool zero_row = false;
    if (normalize) {
        float acc = 0.0f;
        for (int j = lane; j < nch; j += kWave) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = j * 4 + e;
                const float v = i < d ? x[i] : 0.0f;
                acc = __builtin_fmaf(v, v, acc);
            }
        }
        const float n2 = butterfly_sum(acc);
        const float nrm = __builtin_sqrtf(n2);
        zero_row = !(nrm > 0.0f) || !(nrm < INFINITY);
        scale_div = nrm;
    }
    const int64_t orow = slots ? slots[r] : first_slot + r;
    for (int j = lane; j < nch; j += kWave) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = j * 4 + e;
            float t = i < d ? x[i] : 0.0f;
            if (normalize) t = zero_row ? 0.0f : t / scale_div;
            v[e] = t;
        }
        if (DT == DT_F32) {
            float4* o = reinterpret_cast<float4*>(out_) + orow * (int64_t)nch + j;
            *o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = DT == DT_BF16 ? f32_to_bf16_rne(v[e]) : f32_to_f16_rne(v[e]);
            uint2* o = reinterpret_cast<uint2*>(out_) + orow * (int64_t)nch + j;
            *o = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scan_topk_kernel<DT, NB, NITER, SLOTS>: exact canonical-score scan of the whole row store for
// up to NB queries at once.
//   - a wave owns row group g = 4 consecutive rows per step; NITER 16-byte loads per row per lane
//     (chunk j = lane + 64*it), i.e. 4*NITER loads in flight per lane before the first use;
//   - the NB query fragments live in registers for the whole kernel;
//   - per (row, query): one fmaf chain per lane in canonical order, then the 4-row butterfly;
//   - per (wave, query): a top-k list distributed over the lanes (wave_topk.h);
//   - per block: the 4 wave lists are merged through LDS and written as k packed keys to
//     partial[q][block][0..k).
// ---------------------------------------------------------------------------------------------
template <int DT, int NB, int NITER, int SLOTS>
__global__ __launch_bounds__(256) void scan_topk_kernel(const void* __restrict__ rows_, int64_t n, int dpad,
                                                        const float* __restrict__ qn, int nq, int k,
                                                        uint32_t row_base, u64* __restrict__ partial,
                                                        int64_t partial_stride_q) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int nchunks = dpad / E;

    float qf[NB][NITER][E];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int j = lane + kWave * it;
#pragma unroll
            for (int e = 0; e < E; ++e)
                qf[b][it][e] = (b < nq && j < nchunks) ? qn[(int64_t)b * dpad + (int64_t)j * E + e] : 0.0f;
        }

    WaveTopK<SLOTS> L[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) L[b].init();

    const uint4* base = reinterpret_cast<const uint4*>(rows_);
    const int64_t ngroups = (n + 3) >> 2;
    const int64_t W = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += W) {
        float w[4][NITER][E];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int64_t row = g * 4 + r;
            row = row < n ? row : n - 1;
            const uint4* p = base + row * (int64_t)nchunks + lane;
#pragma unroll
            for (int it = 0; it < NITER; ++it) {
                uint4 c = make_uint4(0u, 0u, 0u, 0u);
                if (lane + kWave * it < nchunks) c = p[kWave * it];
                RT::widen(c, w[r][it]);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int it = 0; it < NITER; ++it)
#pragma unroll
                    for (int e = 0; e < E; ++e) acc = __builtin_fmaf(qf[b][it][e], w[r][it][e], acc);
                a[r] = acc;
            }
            const float y = butterfly_sum4(a[0], a[1], a[2], a[3], lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), 16 * r));
                const int64_t row = g * 4 + r;
                if (row < n) L[b].offer(make_key(s, row_base + (uint32_t)row), k, lane);
            }
        }
    }

    // block merge through LDS: [wave][b][slot][lane]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* lds = reinterpret_cast<u64*>(smem_raw);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) lds[((wave * NB + b) * SLOTS + s) * kWave + lane] = L[b].v[s];
    __syncthreads();
    for (int b = wave; b < nq; b += 4) {
        WaveTopK<SLOTS> M;
        M.init();
        for (int wv = 0; wv < 4; ++wv)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                u64 cand = lds[((wv * NB + b) * SLOTS + s) * kWave + lane];
                if (s * kWave + lane >= k) cand = 0ull;
                M.offer_lanes(cand, k, lane);
            }
        u64* dst = partial + (int64_t)b * partial_stride_q + (int64_t)blockIdx.x * k;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int rank = s * kWave + lane;
            if (rank < k) dst[rank] = M.v[s];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// merge_keys_kernel: block b reduces in[b][0..m) to its k largest keys (descending) and writes
// keys and/or (distance, row).  Used for the per-block partials of a scan and for the
// all-gathered shard partials.
// ---------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ __launch_bounds__(256) void merge_keys_kernel(const u64* __restrict__ in, int64_t m, int k,
                                                         u64* __restrict__ out_keys, float* __restrict__ out_dist,
                                                         int64_t* __restrict__ out_rows) {
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const u64* src = in + (int64_t)blockIdx.x * m;
    WaveTopK<SLOTS> L;
    L.init();
    for (int64_t i0 = (int64_t)wave * kWave; i0 < m; i0 += 256) {
        const int64_t i = i0 + lane;
        const u64 cand = i < m ? src[i] : 0ull;
        L.offer_lanes(cand, k, lane);
    }
    __shared__ u64 lds[4 * SLOTS * kWave];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) lds[(wave * SLOTS + s) * kWave + lane] = L.v[s];
    __syncthreads();
    if (wave != 0) return;
    for (int wv = 1; wv < 4; ++wv)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            u64 cand = lds[(wv * SLOTS + s) * kWave + lane];
            if (s * kWave + lane >= k) cand = 0ull;
            L.offer_lanes(cand, k, lane);
        }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int rank = s * kWave + lane;
        if (rank < k) {
            const u64 key = L.v[s];
            const int64_t o = (int64_t)blockIdx.x * k + rank;
            if (out_keys) out_keys[o] = key;
            if (out_dist) out_dist[o] = key ? 1.0f - key_score(key) : INFINITY;
            if (out_rows) out_rows[o] = key ? (int64_t)key_row(key) : (int64_t)-1;
        }
    }
}

}  // namespace

// =============================================================================================
// host side: the index object and the C ABI
// =============================================================================================

struct codd_knn_index {
    int device = 0;
    int dim = 0;
    int dpad = 0;
    int dtype = 0;
    int metric = 0;
    int num_cus = 256;
    int scan_blocks_per_cu = 4;
    int64_t capacity = 0;  // row slots allocated
    int64_t count = 0;     // highest written slot + 1
    void* rows = nullptr;  // [capacity][dpad] storage dtype

    // workspaces (grown on demand, never inside a captured region after warm-up)
    float* qn = nullptr;       // [B][dpad] normalised queries
    int64_t qn_cap = 0;        // in queries
    u64* partial = nullptr;    // [B][blocks][k]
    int64_t partial_cap = 0;   // in keys
    u64* keys_tmp = nullptr;   // [B][k] for codd_knn_search
    int64_t keys_tmp_cap = 0;

    int64_t stat_searches = 0;
    int64_t stat_scan_launches = 0;
    int64_t stat_last_scan_blocks = 0;

    // optional HIP-event timing of the dominant kernel (bench.py's roofline figure):
    // one (start, stop) pair per scan launch, recorded on the launch stream, read after a sync
    bool profile = false;
    std::vector<hipEvent_t> ev;  // 2 * pairs
    int ev_used = 0;             // pairs recorded since the last reset
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return e__ == hipErrorOutOfMemory ? CODD_KNN_ENOMEM : CODD_KNN_EDEVICE;          \
        }                                                                                    \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            changed = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

size_t elem_size(int dtype) { return dtype == DT_F32 ? 4 : 2; }
int elems_per_chunk(int dtype) { return dtype == DT_F32 ? 4 : 8; }

int ensure_rows(codd_knn_index* ix, int64_t need) {
    if (need <= ix->capacity) return CODD_KNN_OK;
    int64_t cap = ix->capacity > 0 ? ix->capacity : 1024;
    while (cap < need) cap += cap / 2 + 1024;
    if (need > cap) cap = need;
    void* fresh = nullptr;
    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    HIP_TRY(hipMalloc(&fresh, (size_t)cap * row_bytes));
    if (ix->rows && ix->count > 0) {
        hipError_t e = hipMemcpy(fresh, ix->rows, (size_t)ix->count * row_bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(CODD_KNN_EDEVICE, "row copy on growth failed: %s", hipGetErrorString(e));
        }
    }
    if (ix->rows) (void)hipFree(ix->rows);
    ix->rows = fresh;
    ix->capacity = cap;
    return CODD_KNN_OK;
}

template <typename T>
int ensure_buf(T** buf, int64_t* cap, int64_t need) {
    if (need <= *cap) return CODD_KNN_OK;
    if (*buf) {
        HIP_TRY(hipDeviceSynchronize());  // a previous search may still read it
        (void)hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
    }
    HIP_TRY(hipMalloc((void**)buf, (size_t)need * sizeof(T)));
    *cap = need;
    return CODD_KNN_OK;
}

int launch_normalize(int dtype, const float* in, int64_t n, int d, int dpad, int normalize, const int64_t* slots,
                     int64_t first_slot, void* out, hipStream_t st) {
    if (n <= 0) return CODD_KNN_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    switch (dtype) {
        case DT_F32:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_F32>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out);
            break;
        case DT_BF16:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_BF16>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out);
            break;
        case DT_F16:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_F16>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out);
            break;
        default:
            return fail(CODD_KNN_EINVAL, "unknown dtype%s");
    }
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

template <int DT, int NB, int NITER>
void launch_scan_slots(int slots, dim3 grid, size_t lds, hipStream_t st, const void* rows, int64_t n, int dpad,
                       const float* qn, int nq, int k, uint32_t row_base, u64* partial, int64_t stride_q) {
    if (slots == 1)
        hipLaunchKernelGGL((scan_topk_kernel<DT, NB, NITER, 1>), grid, dim3(256), lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q);
    else
        hipLaunchKernelGGL((scan_topk_kernel<DT, NB, NITER, 2>), grid, dim3(256), lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q);
}

template <int DT, int NB>
int launch_scan_niter(int niter, int slots, dim3 grid, size_t lds, hipStream_t st, const void* rows, int64_t n, int dpad,
                      const float* qn, int nq, int k, uint32_t row_base, u64* partial, int64_t stride_q) {
    switch (niter) {
        case 1: launch_scan_slots<DT, NB, 1>(slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q); break;
        case 2: launch_scan_slots<DT, NB, 2>(slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q); break;
        case 3: launch_scan_slots<DT, NB, 3>(slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q); break;
        case 4: launch_scan_slots<DT, NB, 4>(slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q); break;
        default: return fail(CODD_KNN_ENOTSUP, "row too wide for the scan kernel%s");
    }
    return CODD_KNN_OK;
}

template <int DT>
int launch_scan_nb(int nb, int niter, int slots, dim3 grid, hipStream_t st, const void* rows, int64_t n, int dpad,
                   const float* qn, int nq, int k, uint32_t row_base, u64* partial, int64_t stride_q) {
    const size_t lds = (size_t)4 * nb * slots * kWave * sizeof(u64);
    switch (nb) {
        case 1: return launch_scan_niter<DT, 1>(niter, slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q);
        case 4: return launch_scan_niter<DT, 4>(niter, slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q);
        case 8: return launch_scan_niter<DT, 8>(niter, slots, grid, lds, st, rows, n, dpad, qn, nq, k, row_base, partial, stride_q);
        default: return fail(CODD_KNN_EINVAL, "bad query group%s");
    }
}

int launch_merge(const u64* in, int B, int64_t m, int k, u64* out_keys, float* out_dist, int64_t* out_rows, hipStream_t st) {
    if (k <= 64)
        hipLaunchKernelGGL(merge_keys_kernel<1>, dim3(B), dim3(256), 0, st, in, m, k, out_keys, out_dist, out_rows);
    else
        hipLaunchKernelGGL(merge_keys_kernel<2>, dim3(B), dim3(256), 0, st, in, m, k, out_keys, out_dist, out_rows);
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

// the whole shard-local search: normalise queries, scan in groups of <= 8 queries, merge.
int search_impl(codd_knn_index* ix, const float* dev_queries, int B, int k, uint32_t row_base, u64* out_keys,
                float* out_dist, int64_t* out_rows, hipStream_t st) {
    if (!ix) return fail(CODD_KNN_EINVAL, "null index%s");
    if (!dev_queries) return fail(CODD_KNN_EINVAL, "null queries%s");
    if (B < 1 || B > CODD_KNN_MAX_BATCH) return fail(CODD_KNN_EINVAL, "B out of range [1,1024]%s");
    if (k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "k out of range [1,128]%s");
    DeviceGuard guard(ix->device);
    ix->stat_searches++;
    const int64_t n = ix->count;
    const int E = elems_per_chunk(ix->dtype);
    const int nchunks = ix->dpad / E;
    const int niter = (nchunks + kWave - 1) / kWave;
    if (niter > 4) return fail(CODD_KNN_ENOTSUP, "dim too large for this dtype (f32 <= 1024, bf16/f16 <= 2048)%s");
    const int slots = k <= 64 ? 1 : 2;

    // grid: enough waves to cover the row groups, capped at a few blocks per CU (grid-stride)
    const int64_t ngroups = (n + 3) / 4;
    int64_t blocks = (ngroups + 3) / 4;
    const int64_t cap_blocks = (int64_t)ix->num_cus * ix->scan_blocks_per_cu;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (blocks < 1) blocks = 1;
    ix->stat_last_scan_blocks = blocks;

    int rc;
    if ((rc = ensure_buf(&ix->qn, &ix->qn_cap, (int64_t)B * ix->dpad)) != 0) return rc;
    const int64_t stride_q = blocks * k;
    if ((rc = ensure_buf(&ix->partial, &ix->partial_cap, (int64_t)B * stride_q)) != 0) return rc;

    if ((rc = launch_normalize(DT_F32, dev_queries, B, ix->dim, ix->dpad, 1, nullptr, 0, ix->qn, st)) != 0) return rc;

    if (n == 0) {
        // nothing stored: all-empty result (chromadb returns {"ids": [[]], ...})
        HIP_TRY(hipMemsetAsync(ix->partial, 0, (size_t)B * sizeof(u64), st));
        return launch_merge(ix->partial, B, 1, k, out_keys, out_dist, out_rows, st);
    }

    for (int q0 = 0; q0 < B; q0 += 8) {
        const int nq = B - q0 < 8 ? B - q0 : 8;
        const int nb = nq == 1 ? 1 : (nq <= 4 ? 4 : 8);
        const float* qn = ix->qn + (int64_t)q0 * ix->dpad;
        u64* part = ix->partial + (int64_t)q0 * stride_q;
        const dim3 grid((unsigned)blocks);
        const bool timed = ix->profile && 2 * (ix->ev_used + 1) <= (int)ix->ev.size();
        if (timed) HIP_TRY(hipEventRecord(ix->ev[2 * ix->ev_used], st));
        switch (ix->dtype) {
            case DT_F32: rc = launch_scan_nb<DT_F32>(nb, niter, slots, grid, st, ix->rows, n, ix->dpad, qn, nq, k, row_base, part, stride_q); break;
            case DT_BF16: rc = launch_scan_nb<DT_BF16>(nb, niter, slots, grid, st, ix->rows, n, ix->dpad, qn, nq, k, row_base, part, stride_q); break;
            case DT_F16: rc = launch_scan_nb<DT_F16>(nb, niter, slots, grid, st, ix->rows, n, ix->dpad, qn, nq, k, row_base, part, stride_q); break;
            default: rc = fail(CODD_KNN_EINVAL, "unknown dtype%s");
        }
        if (rc != 0) return rc;
        HIP_TRY(hipGetLastError());
        ix->stat_scan_launches++;
        if (timed) {
            HIP_TRY(hipEventRecord(ix->ev[2 * ix->ev_used + 1], st));
            ix->ev_used++;
        }
    }
    return launch_merge(ix->partial, B, stride_q, k, out_keys, out_dist, out_rows, st);
}

}  // namespace

extern "C" {

const char* codd_knn_version(void) { return "codd_knn 0.1.0 gfx950"; }
const char* codd_knn_last_error(void) { return g_err; }

int codd_knn_create(codd_knn_index** out, int device, int dim, int dtype, int metric) {
    if (!out) return fail(CODD_KNN_EINVAL, "null out pointer%s");
    *out = nullptr;
    if (dim < 1 || dim > 4096) return fail(CODD_KNN_EINVAL, "dim out of range [1,4096]%s");
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_F16) return fail(CODD_KNN_EINVAL, "unknown dtype%s");
    if (metric != CODD_KNN_METRIC_COSINE) return fail(CODD_KNN_ENOTSUP, "only the cosine metric exists on this path%s");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CODD_KNN_EINVAL, "no such device%s");
    const int dpad = (dim + 63) / 64 * 64;
    const int niter = (dpad / elems_per_chunk(dtype) + kWave - 1) / kWave;
    if (niter > 4) return fail(CODD_KNN_ENOTSUP, "dim too large for this dtype (f32 <= 1024, bf16/f16 <= 2048)%s");
    codd_knn_index* ix = new (std::nothrow) codd_knn_index();
    if (!ix) return fail(CODD_KNN_ENOMEM, "host allocation failed%s");
    ix->device = device;
    ix->dim = dim;
    ix->dpad = dpad;
    ix->dtype = dtype;
    ix->metric = metric;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ix->num_cus = prop.multiProcessorCount;
    *out = ix;
    return CODD_KNN_OK;
}

int codd_knn_destroy(codd_knn_index* ix) {
    if (!ix) return CODD_KNN_OK;
    DeviceGuard guard(ix->device);
    (void)hipDeviceSynchronize();
    if (ix->rows) (void)hipFree(ix->rows);
    if (ix->qn) (void)hipFree(ix->qn);
    if (ix->partial) (void)hipFree(ix->partial);
    if (ix->keys_tmp) (void)hipFree(ix->keys_tmp);
    for (hipEvent_t e : ix->ev) (void)hipEventDestroy(e);
    delete ix;
    return CODD_KNN_OK;
}

int codd_knn_reserve(codd_knn_index* ix, int64_t rows) {
    if (!ix || rows < 0) return fail(CODD_KNN_EINVAL, "bad reserve arguments%s");
    if (rows >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "row slots must fit 32 bits%s");
    DeviceGuard guard(ix->device);
    if (rows <= ix->capacity) return CODD_KNN_OK;
    HIP_TRY(hipDeviceSynchronize());
    // exact growth: the caller states the final size (288 GB of HBM is the only limit)
    void* fresh = nullptr;
    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    HIP_TRY(hipMalloc(&fresh, (size_t)rows * row_bytes));
    if (ix->rows && ix->count > 0) {
        hipError_t e = hipMemcpy(fresh, ix->rows, (size_t)ix->count * row_bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(CODD_KNN_EDEVICE, "row copy on growth failed: %s", hipGetErrorString(e));
        }
    }
    if (ix->rows) (void)hipFree(ix->rows);
    ix->rows = fresh;
    ix->capacity = rows;
    return CODD_KNN_OK;
}

int codd_knn_upsert_host(codd_knn_index* ix, const int64_t* host_slots, const float* host_vecs, int64_t n, int normalize) {
    if (!ix || (n > 0 && (!host_slots || !host_vecs)) || n < 0) return fail(CODD_KNN_EINVAL, "bad upsert arguments%s");
    if (n == 0) return CODD_KNN_OK;
    int64_t max_slot = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (host_slots[i] < 0 || host_slots[i] >= 0xfffffffell) return fail(CODD_KNN_EINVAL, "row slot out of range%s");
        if (host_slots[i] > max_slot) max_slot = host_slots[i];
    }
    DeviceGuard guard(ix->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = ensure_rows(ix, max_slot + 1);
    if (rc != 0) return rc;
    // stage in bounded pieces (<= 64 MiB of vectors per piece)
    const int64_t piece = (int64_t)(64ll << 20) / ((int64_t)ix->dim * 4) + 1;
    float* dvec = nullptr;
    int64_t* dslot = nullptr;
    const int64_t pn = n < piece ? n : piece;
    HIP_TRY(hipMalloc((void**)&dvec, (size_t)pn * ix->dim * sizeof(float)));
    if (hipMalloc((void**)&dslot, (size_t)pn * sizeof(int64_t)) != hipSuccess) {
        (void)hipFree(dvec);
        return fail(CODD_KNN_ENOMEM, "staging allocation failed%s");
    }
    rc = CODD_KNN_OK;
    for (int64_t i0 = 0; i0 < n && rc == 0; i0 += pn) {
        const int64_t m = n - i0 < pn ? n - i0 : pn;
        hipError_t e = hipMemcpy(dvec, host_vecs + i0 * ix->dim, (size_t)m * ix->dim * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dslot, host_slots + i0, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = fail(CODD_KNN_EDEVICE, "staging copy failed: %s", hipGetErrorString(e));
            break;
        }
        rc = launch_normalize(ix->dtype, dvec, m, ix->dim, ix->dpad, normalize, dslot, 0, ix->rows, nullptr);
        if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = fail(CODD_KNN_EDEVICE, "ingest kernel failed%s");
    }
    (void)hipFree(dvec);
    (void)hipFree(dslot);
    if (rc == 0 && max_slot + 1 > ix->count) ix->count = max_slot + 1;
    return rc;
}

int codd_knn_upsert_device(codd_knn_index* ix, int64_t first_slot, const float* dev_vecs, int64_t n, int normalize, void* stream) {
    if (!ix || n < 0 || first_slot < 0 || (n > 0 && !dev_vecs)) return fail(CODD_KNN_EINVAL, "bad upsert arguments%s");
    if (n == 0) return CODD_KNN_OK;
    if (first_slot + n >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "row slots must fit 32 bits%s");
    DeviceGuard guard(ix->device);
    if (first_slot + n > ix->capacity) {
        HIP_TRY(hipDeviceSynchronize());
        int rc = ensure_rows(ix, first_slot + n);
        if (rc != 0) return rc;
    }
    int rc = launch_normalize(ix->dtype, dev_vecs, n, ix->dim, ix->dpad, normalize, nullptr, first_slot, ix->rows, (hipStream_t)stream);
    if (rc != 0) return rc;
    if (first_slot + n > ix->count) ix->count = first_slot + n;
    return CODD_KNN_OK;
}

int codd_knn_count(const codd_knn_index* ix, int64_t* out) {
    if (!ix || !out) return fail(CODD_KNN_EINVAL, "bad count arguments%s");
    *out = ix->count;
    return CODD_KNN_OK;
}

int codd_knn_dim(const codd_knn_index* ix, int* dim, int* padded_dim, int* dtype) {
    if (!ix) return fail(CODD_KNN_EINVAL, "null index%s");
    if (dim) *dim = ix->dim;
    if (padded_dim) *padded_dim = ix->dpad;
    if (dtype) *dtype = ix->dtype;
    return CODD_KNN_OK;
}

int codd_knn_read_rows(const codd_knn_index* ix, int64_t first, int64_t n, void* host_out) {
    if (!ix || first < 0 || n < 0 || first + n > ix->count || (n > 0 && !host_out)) return fail(CODD_KNN_EINVAL, "bad read_rows range%s");
    if (n == 0) return CODD_KNN_OK;
    DeviceGuard guard(ix->device);
    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, (const char*)ix->rows + (size_t)first * row_bytes, (size_t)n * row_bytes, hipMemcpyDeviceToHost));
    return CODD_KNN_OK;
}

int codd_knn_search(codd_knn_index* ix, const float* dev_queries, int B, int k, float* dev_dist, int64_t* dev_rows, void* stream) {
    if (!dev_dist || !dev_rows) return fail(CODD_KNN_EINVAL, "null output%s");
    return search_impl(ix, dev_queries, B, k, 0u, nullptr, dev_dist, dev_rows, (hipStream_t)stream);
}

int codd_knn_search_keys(codd_knn_index* ix, const float* dev_queries, int B, int k, uint32_t row_base, uint64_t* dev_keys, void* stream) {
    if (!dev_keys) return fail(CODD_KNN_EINVAL, "null output%s");
    if (ix && (int64_t)row_base + ix->count >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "global row ids must fit 32 bits%s");
    return search_impl(ix, dev_queries, B, k, row_base, (u64*)dev_keys, nullptr, nullptr, (hipStream_t)stream);
}

int codd_knn_merge_keys(int device, const uint64_t* dev_keys_in, int B, int m, int k, uint64_t* dev_keys_out, float* dev_dist,
                        int64_t* dev_rows, void* stream) {
    if (!dev_keys_in || B < 1 || m < 1 || k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "bad merge arguments%s");
    DeviceGuard guard(device);
    return launch_merge((const u64*)dev_keys_in, B, m, k, (u64*)dev_keys_out, dev_dist, dev_rows, (hipStream_t)stream);
}

int codd_knn_set_option(codd_knn_index* ix, const char* key, int64_t value) {
    if (!ix || !key) return fail(CODD_KNN_EINVAL, "bad option arguments%s");
    if (strcmp(key, "scan_blocks_per_cu") == 0) {
        if (value < 1 || value > 8) return fail(CODD_KNN_EINVAL, "scan_blocks_per_cu must be in [1,8]%s");
        ix->scan_blocks_per_cu = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "profile") == 0) {
        // value = number of (start, stop) event pairs to keep (0 switches timing off); resets the log
        if (value < 0 || value > 65536) return fail(CODD_KNN_EINVAL, "profile pairs must be in [0,65536]%s");
        DeviceGuard guard(ix->device);
        while ((int64_t)ix->ev.size() < 2 * value) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ix->ev.push_back(e);
        }
        ix->profile = value > 0;
        ix->ev_used = 0;
        return CODD_KNN_OK;
    }
    return fail(CODD_KNN_EINVAL, "unknown option: %s", key);
}

int codd_knn_get_stat(const codd_knn_index* ix, const char* key, int64_t* out) {
    if (!ix || !key || !out) return fail(CODD_KNN_EINVAL, "bad stat arguments%s");
    if (strcmp(key, "searches") == 0) *out = ix->stat_searches;
    else if (strcmp(key, "scan_launches") == 0) *out = ix->stat_scan_launches;
    else if (strcmp(key, "last_scan_blocks") == 0) *out = ix->stat_last_scan_blocks;
    else if (strcmp(key, "scan_events") == 0) *out = ix->ev_used;
    else if (strcmp(key, "scan_time_ns") == 0) {
        // sum of the recorded scan launches' durations; synchronises on the last stop event
        double total_ms = 0.0;
        if (ix->ev_used > 0) {
            DeviceGuard guard(ix->device);
            HIP_TRY(hipEventSynchronize(ix->ev[2 * ix->ev_used - 1]));
            for (int i = 0; i < ix->ev_used; ++i) {
                float ms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms, ix->ev[2 * i], ix->ev[2 * i + 1]));
                total_ms += ms;
            }
        }
        *out = (int64_t)(total_ms * 1e6);
    }
    else if (strcmp(key, "capacity_rows") == 0) *out = ix->capacity;
    else if (strcmp(key, "num_cus") == 0) *out = ix->num_cus;
    else if (strcmp(key, "device_bytes") == 0)
        *out = ix->capacity * (int64_t)ix->dpad * (int64_t)elem_size(ix->dtype) + ix->qn_cap * 4 + ix->partial_cap * 8 + ix->keys_tmp_cap * 8;
    else return fail(CODD_KNN_EINVAL, "unknown stat: %s", key);
    return CODD_KNN_OK;
}

}  // extern "C"
