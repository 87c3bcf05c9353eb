// codd_knn.hip — HIP kernels (gfx950 / CDNA4, wave64) and the C ABI of include/codd_knn.h.
//
// What runs here is the arithmetic half of ChromaDB on Codd's search_relevant_metrics path
// (reference call sites: codd_dal/metrics/metrics_semantic_metadata_store.py:60-69 create,
// :236-238 upsert, :314-316 query; scoring :336).  Kernels:
//
//   normalize_rows_kernel   ingest + query prep: c <- c/|c| (canonical sum of squares, IEEE sqrt/div),
//                           writes the stored row AND its bf16 shadow in MFMA-fragment order
//   prep_queries_kernel     one launch per batch of <= 256 queries: normalise, pack the MFMA fragments,
//                           clear the pass's control block
//   scan_topk_kernel        exact streaming scan (small batches, fallback): one wave owns 4 rows per
//                           step, 16-B/lane coalesced loads, fmaf chains in canonical order,
//                           wave-distributed top-k lists
//   merge_keys_kernel       integer top-k of packed keys (per-block partials, shard partials)
//   filter_gemm.h           large batches: bf16 MFMA filter + exact fp32 re-score (finalize)
//
// HBM-bound byte streaming throughout: the corpus is read once per pass, each row by exactly one
// wave, straight into registers; results leave as 8-byte keys.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "codd_knn.h"

#ifndef CODD_EXPERIMENTS
#define CODD_EXPERIMENTS 0  // 1 (build_variant only): the diagnostic switches that can return wrong results exist
#endif
#include "filter_gemm.h"
#include "filter_i8.h"
#include "row_traits.h"
#include "wave_topk.h"

using namespace codd;

// =============================================================================================
// device code
// =============================================================================================

namespace {

// ---------------------------------------------------------------------------------------------
// normalize_rows_kernel: one wave per input vector.  in: n x d fp32 (row stride d).
// out row = slots ? slots[r] : first_slot + r, width dpad, storage dtype DT.
// Sum of squares in the canonical order with E = 4 (the input is fp32), then IEEE sqrt and
// IEEE division per element; a zero / non-finite norm stores an all-zero row.
// shadow (optional): the bf16 rounding of the STORED value, in fragment order (filter_gemm.h).
// ---------------------------------------------------------------------------------------------
// |x| of one vector by the canonical sum of squares: chunk j of 4 elements belongs to lane j % 64
__device__ __forceinline__ float canonical_norm(const float* __restrict__ x, int d, int nch, int lane) {
    float acc = 0.0f;
    for (int j = lane; j < nch; j += kWave) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = j * 4 + e;
            const float v = i < d ? x[i] : 0.0f;
            acc = __builtin_fmaf(v, v, acc);
        }
    }
    return __builtin_sqrtf(butterfly_sum(acc));
}

// Query preparation of one filter pass in ONE launch (B <= 256): q <- q/|q| exactly as normalize_rows_kernel<f32> does
// (qn, for finalize's exact re-scoring), the same values rounded into the MFMA fragment order (qfrag; queries >= B
// are zero), and the pass's control block cleared.  One wave per query slot, 64 blocks.
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ in, int B, int d, int dpad, float* __restrict__ qn,
                                                           uint2* __restrict__ qfrag, unsigned* __restrict__ ctl, int ctl_words) {
    const int lane = lane_id();
    const int q = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < ctl_words; i += (int)gridDim.x * 256) ctl[i] = 0u;
    const int nch = dpad >> 2;
    if (q >= B) {
        for (int j = lane; j < nch; j += kWave) qfrag[codd::qfrag_piece_index(q, j >> 1) * 2 + (j & 1)] = make_uint2(0u, 0u);
        return;
    }
    const float* x = in + (int64_t)q * d;
    const float nrm = canonical_norm(x, d, nch, lane);
    const bool zero_row = !(nrm > 0.0f) || !(nrm < INFINITY);
    for (int j = lane; j < nch; j += kWave) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = j * 4 + e;
            const float t = i < d ? x[i] : 0.0f;
            v[e] = zero_row ? 0.0f : t / nrm;
        }
        reinterpret_cast<float4*>(qn)[(int64_t)q * nch + j] = make_float4(v[0], v[1], v[2], v[3]);
        qfrag[codd::qfrag_piece_index(q, j >> 1) * 2 + (j & 1)] = make_uint2(codd::pack_bf16x2(v[0], v[1]), codd::pack_bf16x2(v[2], v[3]));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, int64_t n, int d, int dpad,
                                                             int normalize, const int64_t* __restrict__ slots,
                                                             int64_t first_slot, void* __restrict__ out_,
                                                             uint2* __restrict__ shadow) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* x = in + r * (int64_t)d;
    const int nch = dpad >> 2;
    const int nsteps = dpad >> 6;
    float scale_div = 1.0f;
    bool zero_row = false;
    if (normalize) {
        const float nrm = canonical_norm(x, d, nch, lane);
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
            uint16_t hb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hb[e] = DT == DT_BF16 ? f32_to_bf16_rne(v[e]) : f32_to_f16_rne(v[e]);
                // the shadow approximates the STORED value, whatever the shadow element type is
                v[e] = DT == DT_F16 ? f16_bits_to_f32(hb[e]) : __uint_as_float((uint32_t)hb[e] << 16);
            }
            uint2* o = reinterpret_cast<uint2*>(out_) + orow * (int64_t)nch + j;
            *o = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
        }
        if (shadow) {
            // elements [4j, 4j+4) = half (j&1) of 16-byte piece c8 = j>>1
            const int64_t piece = shadow_piece_index(orow, j >> 1, nsteps);
            shadow[piece * 2 + (j & 1)] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// shadow_from_rows_kernel: rebuild the bf16 fragment-order shadow of rows [first, first+n) from the
// stored rows themselves (index load from disk: the rows arrive already normalised and rounded).
template <int DT>
__global__ __launch_bounds__(256) void shadow_from_rows_kernel(const void* __restrict__ rows_, int64_t first, int64_t n, int dpad,
                                                               uint2* __restrict__ shadow) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t row = first + r;
    const int nch = dpad >> 2, nsteps = dpad >> 6;
    for (int j = lane; j < nch; j += kWave) {
        float v[4];
        if (DT == DT_F32) {
            const float4 x = reinterpret_cast<const float4*>(rows_)[row * (int64_t)nch + j];
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
            const uint2 x = reinterpret_cast<const uint2*>(rows_)[row * (int64_t)nch + j];
            const uint16_t hb[4] = {(uint16_t)(x.x & 0xffffu), (uint16_t)(x.x >> 16), (uint16_t)(x.y & 0xffffu), (uint16_t)(x.y >> 16)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = DT == DT_BF16 ? __uint_as_float((uint32_t)hb[e] << 16) : f16_bits_to_f32(hb[e]);
        }
        const int64_t piece = shadow_piece_index(row, j >> 1, nsteps);
        shadow[piece * 2 + (j & 1)] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
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
// scan_topk_body: the scan as a workgroup of NW waves sees it — workgroup `bid` of `nblocks`, lds = NW * NB * SLOTS * 64 u64.
// `total` queries, dense from qn (qlist == nullptr) or listed (qlist[i] = query slot; `listed` also selects the one-launch
// hand-off to the last block, see below).  Shared by scan_topk_kernel and by the scan role of finalize_fb_kernel.
template <int DT, int NB, int NITER, int SLOTS, int NW>
__device__ __forceinline__ void scan_topk_body(const void* __restrict__ rows_, int64_t n, int dpad, const float* __restrict__ qn, int total, int k,
                                               uint32_t row_base, u64* __restrict__ partial, int64_t partial_stride_q,
                                               const unsigned* __restrict__ qlist, bool listed, unsigned* __restrict__ merge_done,
                                               u64* __restrict__ merged_keys, float* __restrict__ merged_dist, int64_t* __restrict__ merged_rows,
                                               unsigned long long* __restrict__ count_total, int bid, int nblocks, u64* __restrict__ lds) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int nchunks = dpad / E;

    // Two ways in: (a) the host names a dense group of nq_arg <= NB queries starting at qn (one pass);
    // (b) LISTED: qcount_ptr / qlist live on the device (the filter's fallback queue) and the kernel walks
    // the queue NB queries at a time — zero queued queries is the common case and costs one empty launch.
    for (int g0 = 0; g0 < total; g0 += NB) {
    const int nq = total - g0 < NB ? total - g0 : NB;

    float qf[NB][NITER][E];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int64_t qi = b < nq ? (qlist ? (int64_t)qlist[g0 + b] : (int64_t)(g0 + b)) : 0;
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int j = lane + kWave * it;
#pragma unroll
            for (int e = 0; e < E; ++e)
                qf[b][it][e] = (b < nq && j < nchunks) ? qn[qi * dpad + (int64_t)j * E + e] : 0.0f;
        }
    }

    WaveTopK<SLOTS> L[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) L[b].init();

    const uint4* base = reinterpret_cast<const uint4*>(rows_);
    const int64_t ngroups = (n + 3) >> 2;
    const int64_t W = (int64_t)nblocks * NW;
    for (int64_t g = (int64_t)bid * NW + wave; g < ngroups; g += W) {
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
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) lds[((wave * NB + b) * SLOTS + s) * kWave + lane] = L[b].v[s];
    __syncthreads();
    for (int b = wave; b < nq; b += NW) {
        WaveTopK<SLOTS> M;
        M.init();
        for (int wv = 0; wv < NW; ++wv)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                u64 cand = lds[((wv * NB + b) * SLOTS + s) * kWave + lane];
                if (s * kWave + lane >= k) cand = 0ull;
                M.offer_lanes(cand, k, lane);
            }
        u64* dst = partial + (int64_t)(g0 + b) * partial_stride_q + (int64_t)bid * k;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int rank = s * kWave + lane;
            if (rank < k) dst[rank] = M.v[s];
        }
    }
    __syncthreads();  // the LDS lists are rewritten by the next group
    }  // group loop

    // LISTED mode with `merge_done`: the block that finishes last merges every queued query's per-block partials and writes
    // the answers into the queries' own slots — the fallback is ONE launch (an empty queue, the normal case, costs one
    // empty launch and touches no counter).  Hand-off: every block's stores, a device-scope fence, the arrival ticket;
    // the last arriver fences again before it reads the other blocks' partials.
    if (listed && merge_done && total > 0) {
        __shared__ unsigned s_last;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(merge_done, 1u) == (unsigned)nblocks - 1u ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        if (threadIdx.x == 0) *merge_done = 0u;  // (ready for the next list-driven launch without a clearing pass)
        if (threadIdx.x == 0 && count_total) atomicAdd(count_total, (unsigned long long)total);
        const int64_t m = (int64_t)nblocks * k;
        for (int qi = wave; qi < total; qi += NW) {
            const u64* src = partial + (int64_t)qi * partial_stride_q;
            WaveTopK<SLOTS> M;
            M.init();
            for (int64_t i0 = 0; i0 < m; i0 += kWave) {
                const int64_t i = i0 + lane;
                M.offer_lanes(i < m ? src[i] : 0ull, k, lane);
            }
            const int64_t o = (int64_t)qlist[qi] * k;
#pragma unroll
            for (int s2 = 0; s2 < SLOTS; ++s2) {
                const int rank = s2 * kWave + lane;
                if (rank < k) {
                    const u64 key = M.v[s2];
                    if (merged_keys) merged_keys[o + rank] = key;
                    if (merged_dist) merged_dist[o + rank] = key ? 1.0f - key_score(key) : INFINITY;
                    if (merged_rows) merged_rows[o + rank] = key ? (int64_t)key_row(key) : (int64_t)-1;
                }
            }
        }
    }
}

template <int DT, int NB, int NITER, int SLOTS>
__global__ __launch_bounds__(256) void scan_topk_kernel(const void* __restrict__ rows_, int64_t n, int dpad,
                                                        const float* __restrict__ qn, int nq_arg, int k,
                                                        uint32_t row_base, u64* __restrict__ partial,
                                                        int64_t partial_stride_q, const unsigned* __restrict__ qlist,
                                                        const unsigned* __restrict__ qcount_ptr, unsigned* __restrict__ merge_done = nullptr,
                                                        u64* __restrict__ merged_keys = nullptr, float* __restrict__ merged_dist = nullptr,
                                                        int64_t* __restrict__ merged_rows = nullptr,
                                                        unsigned long long* __restrict__ count_total = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int total = qcount_ptr ? (int)*qcount_ptr : nq_arg;
    scan_topk_body<DT, NB, NITER, SLOTS, 4>(rows_, n, dpad, qn, total, k, row_base, partial, partial_stride_q, qlist, qcount_ptr != nullptr, merge_done, merged_keys,
                                            merged_dist, merged_rows, count_total, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<u64*>(smem_raw));
}

// ---------------------------------------------------------------------------------------------
// finalize_fb_kernel: finalize and the exact-scan fallback of a filter pass in ONE launch (round 3: the fallback used to be a
// launch of its own behind finalize — normally empty, still 4 us + a dependent-launch gap of ~10 us on every step).
//   workgroups x < nq  : finalize_kernel's work for query x (share blockIdx.y of its candidates);
//   workgroups x >= nq : the exact scan over the queries whose candidate lists were truncated (hit_cnt > cap_q).  They need
//     nothing from the finalize workgroups: each derives the queue from the counters itself (same order in every workgroup)
//     and leaves at once when it is empty — the normal case.  Otherwise: scan_topk_body over the queue, the last scan
//     workgroup to finish merges the per-workgroup partials and writes the answers into the queries' slots.
// k <= 64 only (one list slot per lane); wider k keeps the two launches.
// ---------------------------------------------------------------------------------------------
template <int DT, int NITER>
__global__ __launch_bounds__(kFinThreads) void finalize_fb_kernel(const void* __restrict__ rows_, int dpad, const float* __restrict__ qn, const u64* __restrict__ hits,
                                                                   const unsigned* __restrict__ hit_cnt, int cap_q, unsigned* __restrict__ flags, int k, float two_eps,
                                                                   uint32_t row_base, u64* __restrict__ out_keys, unsigned long long* __restrict__ stats,
                                                                   const float* __restrict__ two_eps_q, u64* __restrict__ part_keys, float* __restrict__ out_dist,
                                                                   int64_t* __restrict__ out_rows, const float2* __restrict__ bmeta, int nq, int64_t n,
                                                                   u64* __restrict__ fb_partial, int64_t fb_stride_q, unsigned* __restrict__ fb_done) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int kWavesHere = kFinThreads / kWave;
    if ((int)blockIdx.x < nq) {
        const int q = blockIdx.x;
        const unsigned total = hit_cnt[q * kHitCntStride];
        const unsigned part = blockIdx.y, nparts = gridDim.y;
        if (total > (unsigned)cap_q) {  // the candidate list was truncated: the scan workgroups answer this query
            if (threadIdx.x == 0 && part == 0) atomicOr(&flags[FLAG_NEED_FALLBACK], 1u);
            return;
        }
        const float eps1 = bmeta ? two_eps_q[256 + q] : 0.5f * (two_eps_q ? two_eps_q[q] : two_eps);
        const float bq = bmeta ? two_eps_q[512 + q] : 0.0f;
        finalize_body<DT, NITER, 1>(rows_, dpad, qn + (int64_t)q * dpad, hits + (int64_t)q * cap_q, total, part, nparts, k, eps1, bq, bmeta, row_base,
                                    out_keys ? out_keys + (int64_t)q * k : nullptr, out_dist ? out_dist + (int64_t)q * k : nullptr,
                                    out_rows ? out_rows + (int64_t)q * k : nullptr, part_keys ? part_keys + ((int64_t)q * nparts + part) * k : nullptr, stats);
        return;
    }
    if (blockIdx.y != 0) return;
    // the queue: queries with a truncated list, in query order (every scan workgroup computes the same list)
    __shared__ unsigned s_fb[kTileQ];
    __shared__ unsigned s_cnt[kWavesHere + 1];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool over = tid < nq && hit_cnt[tid * kHitCntStride] > (unsigned)cap_q;   // (nq <= 256 < kFinThreads)
    const u64 mask = __ballot(over);
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < kWavesHere; ++w) {
        if (w < wave) base += s_cnt[w];
        total += s_cnt[w];
    }
    if (total == 0) return;   // (uniform: nothing was truncated — the normal case)
    if (over) s_fb[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)tid;
    __syncthreads();
    // (4 queries per pass over the rows, not 8: the role is rare, and its query registers must not push the finalize role into scratch)
    scan_topk_body<DT, 4, NITER, 1, kWavesHere>(rows_, n, dpad, qn, (int)total, k, row_base, fb_partial, fb_stride_q, s_fb, true, fb_done, out_keys, out_dist, out_rows,
                                                 stats ? stats + 2 : nullptr, (int)blockIdx.x - nq, (int)gridDim.x - nq, reinterpret_cast<u64*>(smem_raw));
}

// ---------------------------------------------------------------------------------------------
// merge_keys_kernel: block b reduces in[b][0..m) to its k largest keys (descending) and writes
// keys and/or (distance, row).  Used for the per-block partials of a scan, for the all-gathered
// shard partials, and (m == k) to unpack final keys.
// ---------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ __launch_bounds__(256) void merge_keys_kernel(const u64* __restrict__ in, int64_t m, int64_t in_stride, int64_t seg_len,
                                                         int64_t seg_stride, int k, u64* __restrict__ out_keys, float* __restrict__ out_dist,
                                                         int64_t* __restrict__ out_rows, const unsigned* __restrict__ out_list,
                                                         const unsigned* __restrict__ count_ptr,
                                                         unsigned long long* __restrict__ count_total) {
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    // LISTED (fallback queue): only the first *count_ptr blocks work, block i answers query out_list[i]
    if (count_ptr) {
        const unsigned cnt = *count_ptr;
        if (blockIdx.x == 0 && threadIdx.x == 0 && count_total && cnt) atomicAdd(count_total, (unsigned long long)cnt);
        if (blockIdx.x >= cnt) return;
    }
    const int64_t oblock = out_list ? (int64_t)out_list[blockIdx.x] : (int64_t)blockIdx.x;
    const u64* src = in + (int64_t)blockIdx.x * in_stride;
    WaveTopK<SLOTS> L;
    L.init();
    for (int64_t i0 = (int64_t)wave * kWave; i0 < m; i0 += 256) {
        const int64_t i = i0 + lane;
        // a query's m keys are m / seg_len runs of seg_len keys, seg_stride apart (one run when seg_len == m;
        // one run per shard when the input is an all_gather of [B][k] partials)
        const u64 cand = i < m ? src[(i / seg_len) * seg_stride + (i % seg_len)] : 0ull;
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
            const int64_t o = oblock * k + rank;
            if (out_keys) out_keys[o] = key;
            if (out_dist) out_dist[o] = key ? 1.0f - key_score(key) : INFINITY;
            if (out_rows) out_rows[o] = key ? (int64_t)key_row(key) : (int64_t)-1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// IVF (coarse lists + exact scores), SURVEY.md §8(f)4 / BASELINE config 5.
// gather_rows_kernel: rows_ivf[i] = rows[perm[i]] (rows regrouped by coarse list), one wave per row.
// widen_rows_kernel : stored rows -> fp32 [n][dim] (index build reads the corpus back through it).
// ivf_scan_kernel   : block (x, b) scans slice x%split of the list that query b probes at rank x/split,
//                     canonical exact scores, per-wave top-k, keys carry the ORIGINAL row slot.
// ---------------------------------------------------------------------------------------------
// perm_check_kernel: sets *bad when any perm[i] lies outside [0, n) (codd_knn_ivf_install trusts no caller-built table:
// gather_rows_kernel and ivf_scan_kernel index rows with these values)
__global__ __launch_bounds__(256) void perm_check_kernel(const int64_t* __restrict__ perm, int64_t n, unsigned* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && (perm[i] < 0 || perm[i] >= n)) atomicOr(bad, 1u);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int64_t* __restrict__ perm, int64_t n,
                                                          int chunks_per_row, uint4* __restrict__ dst, uint32_t* __restrict__ ids) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t from = perm[r];
    for (int j = lane; j < chunks_per_row; j += kWave) dst[r * chunks_per_row + j] = src[from * chunks_per_row + j];
    if (lane == 0) ids[r] = (uint32_t)from;
}

// ---------------------------------------------------------------------------------------------------------------
// int8 shadow (half the bytes of the bf16 shadow).  Row r is stored as round(c_i / scale_b), scale_b = the largest |c_i| of
// its 32-row block / 127 (rscale[] holds it once per row), in the same fragment order as the bf16 shadow with 16-element pieces and 128-element K-steps.  The
// filter's error bound needs |c - c~| for the worst row: every wave folds its row's error norm into *eps_r (ordered bits).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float butterfly_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}
__device__ __forceinline__ uint32_t quantize4(const float (&v)[4], float inv_scale, float scale, float& err2) {
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = __builtin_rintf(v[e] * inv_scale);
        t = fminf(fmaxf(t, -127.0f), 127.0f);
        const float d = v[e] - t * scale;
        err2 = __builtin_fmaf(d, d, err2);
        packed |= ((uint32_t)(int)t & 0xffu) << (8 * e);
    }
    return packed;
}
// One workgroup quantises one 32-row corpus block (a wave's rows in the filter GEMMs) with ONE scale for the whole block:
// max |c_i| over its 32 rows / 127.  (Round 1 scaled every row by its own maximum.  A scale that is uniform inside a block
// lets the filter's epilogue test accumulators against an integer threshold before any conversion; the price, a coarser
// grid for rows whose own maximum is smaller, is in the measured error norm *eps_r like every other quantisation error.)
// Pass 1 finds the block maximum, pass 2 quantises the block in two halves of 16 rows (4 rows per wave), collects each half in
// LDS and writes its pieces out in fragment order: 16 consecutive lanes = the 16 rows of one piece column = 256 contiguous
// bytes (a wave per row writing its own 16-byte pieces, 256 bytes apart, ran at a tenth of the HBM rate).
template <int DT>
__global__ __launch_bounds__(256) void shadow8_from_rows_kernel(const void* __restrict__ rows_, int64_t first32, int64_t n, int dpad, int dpad8,
                                                                uint4* __restrict__ shadow8, float* __restrict__ rscale,
                                                                float2* __restrict__ bmeta) {
    constexpr int kRowDwords = 516;  // 512 + 4: sixteen rows read column-wise hit 64 different banks
    __shared__ __attribute__((aligned(16))) uint32_t tile[16 * kRowDwords];
    __shared__ float s_max[4];
    __shared__ float s_err[4];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t row32 = first32 + (int64_t)blockIdx.x * 32;
    const int nch = dpad >> 2, nch8 = dpad8 >> 2, nsteps8 = dpad8 >> 7;
    const int nit = (nch8 + kWave - 1) / kWave;  // chunks of 4 elements per lane and row
    // chunk j = lane + 64 * it of a row (zeros past the row's end and for rows past n)
    auto load_chunk = [&](int64_t row, int it, float (&v)[4]) __attribute__((always_inline)) {
        const int j = lane + kWave * it;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.0f;
        if (j < nch && row < n) {
            if (DT == DT_F32) {
                const float4 x = reinterpret_cast<const float4*>(rows_)[row * (int64_t)nch + j];
                v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            } else {
                const uint2 x = reinterpret_cast<const uint2*>(rows_)[row * (int64_t)nch + j];
                const uint16_t hb[4] = {(uint16_t)(x.x & 0xffffu), (uint16_t)(x.x >> 16), (uint16_t)(x.y & 0xffffu), (uint16_t)(x.y >> 16)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = DT == DT_BF16 ? __uint_as_float((uint32_t)hb[e] << 16) : f16_bits_to_f32(hb[e]);
            }
        }
    };
    // pass 1: the block's largest magnitude.  Chunk-major, the wave's 8 rows inside: eight independent loads in flight per
    // lane (row-major with one row at a time, the kernel was a chain of HBM round trips: 26 ms for 10M x 768 rows)
    float vmax = 0.0f;
    for (int it = 0; it < nit; ++it) {
        float v[8][4];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) load_chunk(row32 + wave * 8 + rr, it, v[rr]);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int e = 0; e < 4; ++e) vmax = fmaxf(vmax, fabsf(v[rr][e]));
    }
    vmax = butterfly_max(vmax);
    if (lane == 0) s_max[wave] = vmax;
    __syncthreads();
    vmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const float scale = vmax > 0.0f ? vmax / 127.0f : 1.0f, inv_scale = 1.0f / scale;
    // pass 2: quantise (the rows come from L2 this time), 16 rows at a time, 4 rows per wave
    float wave_err = 0.0f;  // largest error norm among this wave's rows
    for (int half = 0; half < 2; ++half) {
        const int64_t row16 = row32 + 16 * half;
        float err2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < nit; ++it) {
            float v[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) load_chunk(row16 + wave * 4 + rr, it, v[rr]);
            const int j = lane + kWave * it;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (j < nch8) tile[(wave * 4 + rr) * kRowDwords + j] = quantize4(v[rr], inv_scale, scale, err2[rr]);  // (rows past n, chunks past the row: zeros)
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = row16 + wave * 4 + rr;
            const float e2 = butterfly_sum(err2[rr]);
            if (row < n) {
                if (lane == 0) rscale[row] = scale;
                wave_err = fmaxf(wave_err, __builtin_sqrtf(e2) * 1.0001f + 1e-7f);  // inflated a little: the norm itself was accumulated in fp32
            }
        }
        __syncthreads();
        // piece (c16, r) of the half block: 16 bytes of row r at byte 16*c16
        for (int p = (int)threadIdx.x; p < 16 * (dpad8 >> 4); p += 256) {
            const int r = p & 15, c16 = p >> 4;
            shadow8[codd::shadow_piece_index(row16 + r, c16, nsteps8)] = *reinterpret_cast<const uint4*>(&tile[r * kRowDwords + c16 * 4]);
        }
        __syncthreads();
    }
    // The block's meta data: its scale and the largest quantisation error norm |c - c~| among its rows.  The filter's bound is
    // evaluated PER BLOCK with this norm (round 2 folded every row's norm into one device-wide maximum that could only grow: one
    // badly quantising row widened every query's slack for the life of the index); the device-wide maximum that the
    // first-generation kernels and the host's "is int8 usable at all" test still use is re-derived from these after every
    // build (eps_max_kernel), so it follows the rows that are stored NOW.
    if (lane == 0) s_err[wave] = wave_err;
    __syncthreads();
    if (threadIdx.x == 0 && row32 < n) bmeta[row32 >> 5] = make_float2(scale, fmaxf(fmaxf(s_err[0], s_err[1]), fmaxf(s_err[2], s_err[3])));
}

// eps_max_kernel: *eps_r_bits = the largest block error norm over blocks [0, nblocks) (one workgroup; after every shadow build)
// eps_r_bits[1] = how many blocks lie above `wide` (the host's "int8 bound useless" level): the per-block kernels only pay for those blocks
__global__ __launch_bounds__(1024) void eps_max_kernel(const float2* __restrict__ bmeta, int64_t nblocks, unsigned* __restrict__ eps_r_bits, float wide) {
    __shared__ float s_part[16];
    __shared__ unsigned s_wide;
    if (threadIdx.x == 0) s_wide = 0u;
    __syncthreads();
    float m = 0.0f;
    unsigned nw = 0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 1024) {
        const float e = bmeta[i].y;
        m = fmaxf(m, e);
        nw += e > wide ? 1u : 0u;
    }
    m = butterfly_max(m);
    if (nw) atomicAdd(&s_wide, nw);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, s_part[w]);
        eps_r_bits[0] = __float_as_uint(m);
        eps_r_bits[1] = s_wide;
    }
}

// prep_queries_kernel for the int8 filter: qn as always; the query quantised like a row (qfrag8, scale in qmeta[q]);
// qmeta[256 + q] = 2 * eps(q) with eps(q) = |q - q~| (1.01 + eps_r) + 1.001 eps_r + 2e-6 >= |<q~, c~> - <q, c>| for every
// stored row c (|c| <= 1.004 whatever the storage type, |c - c~| <= eps_r, |q| <= 1 + 1e-6; the integer accumulation
// is exact and the two scale multiplications cost < 4e-7)
__global__ __launch_bounds__(256) void prep_queries8_kernel(const float* __restrict__ in, int B, int d, int dpad, int dpad8,
                                                            float* __restrict__ qn, uint32_t* __restrict__ qfrag8, float* __restrict__ qmeta,
                                                            const unsigned* __restrict__ eps_r_bits, unsigned* __restrict__ ctl, int ctl_words,
                                                            float slack_scale) {
    const int lane = lane_id();
    const int q = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < ctl_words; i += (int)gridDim.x * 256) ctl[i] = 0u;
    const int nch = dpad >> 2, nch8 = dpad8 >> 2;
    if (q >= B) {
        for (int j = lane; j < nch8; j += kWave) qfrag8[codd::qfrag_piece_index(q, j >> 2) * 4 + (j & 3)] = 0u;
        if (lane == 0) { qmeta[q] = 0.0f; qmeta[256 + q] = 0.0f; qmeta[512 + q] = 0.0f; qmeta[768 + q] = 0.0f; }
        return;
    }
    const float* x = in + (int64_t)q * d;
    const float nrm = canonical_norm(x, d, nch, lane);
    const bool zero_row = !(nrm > 0.0f) || !(nrm < INFINITY);
    float vmax = 0.0f;
    for (int j = lane; j < nch; j += kWave) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = j * 4 + e;
            const float t = i < d ? x[i] : 0.0f;
            v[e] = zero_row ? 0.0f : t / nrm;
            vmax = fmaxf(vmax, fabsf(v[e]));
        }
        reinterpret_cast<float4*>(qn)[(int64_t)q * nch + j] = make_float4(v[0], v[1], v[2], v[3]);
    }
    vmax = butterfly_max(vmax);
    const float scale = vmax > 0.0f ? vmax / 127.0f : 0.0f, inv_scale = vmax > 0.0f ? 127.0f / vmax : 0.0f;
    float err2 = 0.0f;
    for (int j = lane; j < nch8; j += kWave) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (j < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = j * 4 + e;
                const float t = i < d ? x[i] : 0.0f;
                v[e] = zero_row ? 0.0f : t / nrm;
            }
        }
        qfrag8[codd::qfrag_piece_index(q, j >> 2) * 4 + (j & 3)] = quantize4(v, inv_scale, scale, err2);
    }
    err2 = butterfly_sum(err2);
    if (lane == 0) {
        const float eq = __builtin_sqrtf(err2) * 1.0001f + 1e-7f, er = __uint_as_float(*eps_r_bits);
        qmeta[q] = scale;
        qmeta[256 + q] = slack_scale * 2.0f * (eq * (1.01f + er) + 1.001f * er + 2e-6f);
        // the same bound split by what it depends on: eps(q, block) = A(q) + B(q) * e_block, e_block = the block's own error norm
        // (bmeta[].y) in place of the device-wide maximum er — what i8_tile_kernel and finalize evaluate per 32-row block
        qmeta[512 + q] = slack_scale * (eq * 1.01f + 2e-6f);
        qmeta[768 + q] = slack_scale * (eq + 1.001f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// small_batch_kernel<DT, NITER, NS>: ONE query answered in ONE launch (BASELINE configs[1]: 1M x 768, B = 1).  The filter chain
// of a batch is six dependent launches (query preparation, sample, thresholds, filter, finalize, merge); at B = 1 over 1M rows
// they cost as much again as streaming the int8 shadow once.  Here:
//   every workgroup : wave 0 quantises the query (same arithmetic as prep_queries8_kernel) into LDS while the other waves' first
//     loads are already in flight; then every wave streams its share of the int8 shadow — 16 rows x dpad8 bytes at a time
//     (2 NS contiguous 1-KiB chunks of the fragment order, the next unit's loads issued before this unit's arithmetic);
//     lane (kq, r) holds 16 elements of row r: v_dot4 against the query's 16 bytes from LDS, two cross-lane adds — keeps
//     its best 64 approximate scores (WaveTopK) and publishes the best kSbKeep of them plus `dropmax`, the best key it did NOT
//     publish (0: none);
//   the workgroup that finishes LAST (device-scope fence + arrival ticket) : finalize_body over the published candidates — the
//     k best approximate ones re-scored exactly give L' <= s_k, every row whose approximate score is >= L' - eps(q) is
//     re-scored exactly (canonical fp32 expression), top-k written.  MARGIN TEST: if some wave's dropmax is >= L' - eps(q) a
//     candidate may have been dropped: the query goes to the exact-scan fallback queue (the list-driven scan launch behind
//     this kernel: normally empty).  No thresholds, no sample pass: the bound is the same eps(q) = |q - q~| (1.01 + eps_r) +
//     1.001 eps_r + 2e-6 as the int8 filter's (device-wide eps_r).
// Results are bit-identical to every other path: the survivors' scores are the canonical ones.
// ---------------------------------------------------------------------------------------------------------------
#if !CODD_EXPERIMENTS && (defined(CODD_SB_EXP_NOOFFER) || defined(CODD_SB_EXP_NOSTREAM) || defined(CODD_SB_EXP_NOFINAL))
#error "the CODD_SB_EXP_* switches return wrong results: they exist only in -DCODD_EXPERIMENTS=1 builds (build_variant)"
#endif
constexpr int kSbKeep = 8;        // keys a wave publishes
constexpr int kSbThreads = kFinThreads;
static_assert(kSbThreads == 512, "finalize_body's workgroup");
template <int DT, int NITER, int NS>
__global__ __launch_bounds__(kSbThreads) void small_batch_kernel(const uint4* __restrict__ shadow8, const float2* __restrict__ bmeta, const void* __restrict__ rows_,
                                                                  int64_t n, int d, int dpad, const float* __restrict__ in, int k, uint32_t row_base,
                                                                  const unsigned* __restrict__ eps_r_bits, float* __restrict__ qn, u64* __restrict__ cand,
                                                                  u64* __restrict__ dropmax, unsigned* __restrict__ ticket, unsigned* __restrict__ fb_count,
                                                                  unsigned* __restrict__ fb_list, u64* __restrict__ out_keys, float* __restrict__ out_dist,
                                                                  int64_t* __restrict__ out_rows, unsigned long long* __restrict__ stats) {
    constexpr int kWaves = kSbThreads / kWave;
    constexpr int kDpad8 = NS * 128;
    typedef unsigned sb_u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char s_q8[kDpad8];   // the query's int8 bytes, natural order
    __shared__ float s_qscale, s_eps, s_lo;
    __shared__ unsigned s_last;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = dpad >> 2;
    const int64_t G = gridDim.x;
    const int64_t nunits = (n + 15) >> 4;
    const int64_t W = G * kWaves;
    const int64_t wave_global = (int64_t)blockIdx.x * kWaves + wave;

    // the chunks of unit u (16 rows): read-once stream, non-temporal.  A unit index past the end is clamped (the loads are
    // unconditional: a conditional load makes hipcc drain the queue at every join); its results are never offered.
    auto load_unit = [&](sb_u32x4(&c)[2 * NS], float& scale, int64_t u) __attribute__((always_inline)) {
        const int64_t uu = u < nunits ? u : nunits - 1;
        scale = bmeta[uu >> 1].x;   // (in front of the chunks: the queue retires in order, a load issued behind them would drain the prefetch)
        const uint4* src = shadow8 + ((uu >> 1) * NS * 4 + (uu & 1) * 2) * 64 + lane;   // chunk (s, ks) at + (s * 4 + ks) * 64 pieces
#pragma unroll
        for (int j = 0; j < 2 * NS; ++j) c[j] = __builtin_nontemporal_load(reinterpret_cast<const sb_u32x4*>(src + ((j >> 1) * 4 + (j & 1)) * 64));
    };
    sb_u32x4 ca[2 * NS], cb[2 * NS];
    float sa = 0.0f, sb = 0.0f;
    int64_t u = wave_global;
    load_unit(ca, sa, u);

    // ---- the query: normalise (block 0 also stores qn: the last workgroup and the fallback scan read it) and quantise ----
    if (wave == 0) {
        const float nrm = canonical_norm(in, d, nch, lane);
        const bool zero_row = !(nrm > 0.0f) || !(nrm < INFINITY);
        float vmax = 0.0f;
        for (int j = lane; j < nch; j += kWave) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = j * 4 + e;
                const float t = i < d ? in[i] : 0.0f;
                v[e] = zero_row ? 0.0f : t / nrm;
                vmax = fmaxf(vmax, fabsf(v[e]));
            }
            if (blockIdx.x == 0) reinterpret_cast<float4*>(qn)[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
        vmax = butterfly_max(vmax);
        const float scale = vmax > 0.0f ? vmax / 127.0f : 0.0f, inv_scale = vmax > 0.0f ? 127.0f / vmax : 0.0f;
        float err2 = 0.0f;
        for (int j = lane; j < kDpad8 / 4; j += kWave) {
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (j < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = j * 4 + e;
                    const float t = i < d ? in[i] : 0.0f;
                    v[e] = zero_row ? 0.0f : t / nrm;
                }
            }
            reinterpret_cast<uint32_t*>(s_q8)[j] = quantize4(v, inv_scale, scale, err2);
        }
        err2 = butterfly_sum(err2);
        if (lane == 0) {
            const float eq = __builtin_sqrtf(err2) * 1.0001f + 1e-7f, er = __uint_as_float(*eps_r_bits);
            s_qscale = scale;
            s_eps = eq * (1.01f + er) + 1.001f * er + 2e-6f;
        }
    }
    __syncthreads();
    const float qscale = s_qscale;

    // ---- the stream ----
    WaveTopK<1> L;
    L.init();
    const int kq = lane >> 4, r = lane & 15;
    auto score_unit = [&](const sb_u32x4(&c)[2 * NS], float scale, int64_t uu) __attribute__((always_inline)) {
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 2 * NS; ++j) {
            const uint4 b = *reinterpret_cast<const uint4*>(s_q8 + j * 64 + kq * 16);
            acc = __builtin_amdgcn_sdot4((int)c[j][0], (int)b.x, acc, false);
            acc = __builtin_amdgcn_sdot4((int)c[j][1], (int)b.y, acc, false);
            acc = __builtin_amdgcn_sdot4((int)c[j][2], (int)b.z, acc, false);
            acc = __builtin_amdgcn_sdot4((int)c[j][3], (int)b.w, acc, false);
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        const int64_t block = uu >> 1;
        const int64_t row = (block << 5) + 16 * (uu & 1) + r;
        const float score = (float)acc * scale * qscale;
        const u64 key = (lane < 16 && row < n) ? make_key(score, (uint32_t)row) : 0ull;
#ifdef CODD_SB_EXP_NOOFFER
        asm volatile("" ::"v"(key));  // diagnostic: scores computed, no top-k insert
#else
        L.offer_lanes(key, kWave, lane);
#endif
    };
#ifdef CODD_SB_EXP_NOSTREAM
    u = nunits;  // diagnostic: no stream at all
#endif
    while (u < nunits) {
        load_unit(cb, sb, u + W);
        score_unit(ca, sa, u);
        u += W;
        if (u >= nunits) break;
        load_unit(ca, sa, u + W);
        score_unit(cb, sb, u);
        u += W;
    }

    // ---- publish: the wave's best kSbKeep keys and the best key it keeps to itself ----
    if (lane < kSbKeep) cand[wave_global * kSbKeep + lane] = L.v[0];
    const u64 dropped = readlane_u64(L.v[0], kSbKeep);
    if (lane == 0) dropmax[wave_global] = dropped;

    // ---- the last workgroup to arrive answers the query ----
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(ticket, 1u) == (unsigned)(G - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
#ifdef CODD_SB_EXP_NOFINAL
    if (tid == 0) { *ticket = 0u; *fb_count = 0u; }
    return;  // diagnostic: nothing behind the arrival ticket (results are garbage)
#endif
    if (tid == 0) {
        *ticket = 0u;    // (the next search finds the ticket at zero: no clearing launch)
        *fb_count = 0u;  // (the fallback queue of THIS search starts empty; the scan behind this kernel reads it)
    }
    __syncthreads();
    finalize_body<DT, NITER, 1>(rows_, dpad, qn, cand, (unsigned)(W * kSbKeep), 0u, 1u, k, s_eps, 0.0f, nullptr, row_base, out_keys, out_dist, out_rows, nullptr, stats,
                                &s_lo);
    // margin test: could a wave have kept a row to itself that belongs among the survivors?
    const float lo = s_lo;
    unsigned bad = 0;
    for (int64_t i = tid; i < W; i += kSbThreads) {
        const u64 dm = dropmax[i];
        if (dm && key_score(dm) >= lo) bad = 1u;
    }
    if (__syncthreads_or((int)bad) && tid == 0) fb_list[atomicAdd(fb_count, 1u)] = 0u;
}

// row_norm_check_kernel: largest | |c| - 1 | over stored rows [first, first + n) that are not all-zero, folded into *dev_bits
// (non-negative floats order as their bits).  codd_knn_load_rows trusts nothing it is handed: both filters' bounds assume unit
// rows, and rows written with normalize = 0 (or a damaged rows.bin) must switch them off, not return wrong neighbours.
template <int DT>
__global__ __launch_bounds__(256) void row_norm_check_kernel(const void* __restrict__ rows_, int64_t first, int64_t n, int dpad,
                                                             unsigned* __restrict__ dev_bits) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int nchunks = dpad / E;
    const uint4* p = reinterpret_cast<const uint4*>(rows_) + (first + r) * (int64_t)nchunks;
    float acc = 0.0f;
    for (int j = lane; j < nchunks; j += kWave) {
        float w[E];
        RT::widen(p[j], w);
#pragma unroll
        for (int e = 0; e < E; ++e) acc = __builtin_fmaf(w[e], w[e], acc);
    }
    const float nrm = __builtin_sqrtf(butterfly_sum(acc));
    if (lane == 0 && nrm != 0.0f) {
        const float dev = nrm == nrm ? fabsf(nrm - 1.0f) : INFINITY;  // NaN rows count as infinitely far from unit
        atomicMax(dev_bits, __float_as_uint(dev));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void widen_rows_kernel(const void* __restrict__ rows_, int64_t first, int64_t n, int dim, int dpad,
                                                         float* __restrict__ out) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int nchunks = dpad / E;
    const uint4* p = reinterpret_cast<const uint4*>(rows_) + (first + r) * (int64_t)nchunks;
    for (int j = lane; j < nchunks; j += kWave) {
        float w[E];
        RT::widen(p[j], w);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (j * E + e < dim) out[r * (int64_t)dim + j * E + e] = w[e];
    }
}

template <int DT, int NITER, int SLOTS>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const void* __restrict__ rows_, const uint32_t* __restrict__ ids,
                                                       const int64_t* __restrict__ offsets, const u64* __restrict__ probe_keys,
                                                       int nprobe, int split, int dpad, const float* __restrict__ qn, int k,
                                                       uint32_t row_base, u64* __restrict__ partial) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int b = blockIdx.y, x = blockIdx.x;
    const int nchunks = dpad / E;
    u64* dst = partial + ((int64_t)b * nprobe * split + x) * k;

    WaveTopK<SLOTS> L;
    L.init();
    const u64 pk = probe_keys[(int64_t)b * nprobe + x / split];
    if (pk != 0ull) {  // block-uniform: fewer lists than nprobe leave empty probe slots
        const uint32_t list = key_row(pk);
        const int64_t lo = offsets[list], hi = offsets[list + 1];
        const int64_t len = hi - lo, part = (len + split - 1) / split;
        const int64_t begin = lo + (x % split) * part;
        const int64_t end = begin + part < hi ? begin + part : hi;

        float qf[NITER][E];
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int j = lane + kWave * it;
#pragma unroll
            for (int e = 0; e < E; ++e) qf[it][e] = j < nchunks ? qn[(int64_t)b * dpad + (int64_t)j * E + e] : 0.0f;
        }
        const uint4* base = reinterpret_cast<const uint4*>(rows_);
        for (int64_t g = begin + wave * 4; g < end; g += 16) {
            float w[4][NITER][E];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = g + r < end ? g + r : end - 1;
                const uint4* p = base + row * (int64_t)nchunks + lane;
#pragma unroll
                for (int it = 0; it < NITER; ++it) {
                    uint4 c = make_uint4(0u, 0u, 0u, 0u);
                    if (lane + kWave * it < nchunks) c = p[kWave * it];
                    RT::widen(c, w[r][it]);
                }
            }
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int it = 0; it < NITER; ++it)
#pragma unroll
                    for (int e = 0; e < E; ++e) acc = __builtin_fmaf(qf[it][e], w[r][it][e], acc);
                a[r] = acc;
            }
            const float y = butterfly_sum4(a[0], a[1], a[2], a[3], lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), 16 * r));
                if (g + r < end) L.offer(make_key(s, row_base + ids[g + r]), k, lane);
            }
        }
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
        if (rank < k) dst[rank] = L.v[s];
    }
}

// ---------------------------------------------------------------------------------------------
// IVF at batch (round 3): a probed list is scanned ONCE for all the queries of the batch that probe it.  ivf_scan_kernel reads a
// list once per (query, list) pair — 256 queries x 32 probes over 2,048 lists read every list four times.  Here the pairs are
// grouped by list on the device (count, prefix sums, scatter: three tiny launches), every work item is (list, up to kIvfNB
// pairs): the list's rows enter registers once and are scored against the item's queries with the canonical exact expression.
// The item writes each pair's top-k to the pair's own slot of the partial buffer, so the merge behind it is the same launch
// as for the unshared scan and the results are the same bits.
// ---------------------------------------------------------------------------------------------
constexpr int kIvfNB = 4;  // queries per work item
__global__ __launch_bounds__(256) void ivf_pair_count_kernel(const u64* __restrict__ probe_keys, int npairs, unsigned* __restrict__ cnt) {
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (p < npairs && probe_keys[p] != 0ull) atomicAdd(&cnt[key_row(probe_keys[p])], 1u);
}
// one workgroup: exclusive prefix sums over the lists — pair_start[l] (where list l's pairs go) and item_start[l] (its work
// items: ceil(cnt / kIvfNB)); cnt[] is cleared on the way (the scatter reuses it as its fill counters); nitems at item_start[nlist]
__global__ __launch_bounds__(1024) void ivf_pair_offsets_kernel(unsigned* __restrict__ cnt, int nlist, unsigned* __restrict__ pair_start,
                                                                unsigned* __restrict__ item_start) {
    __shared__ unsigned s_p[1024], s_i[1024];
    const int tid = (int)threadIdx.x;
    const int per = (nlist + 1023) / 1024;
    unsigned sp = 0, si = 0;
    for (int j = 0; j < per; ++j) {
        const int l = tid * per + j;
        const unsigned c = l < nlist ? cnt[l] : 0u;
        sp += c;
        si += (c + kIvfNB - 1) / kIvfNB;
    }
    s_p[tid] = sp;
    s_i[tid] = si;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        const unsigned a = tid >= off ? s_p[tid - off] : 0u, b = tid >= off ? s_i[tid - off] : 0u;
        __syncthreads();
        s_p[tid] += a;
        s_i[tid] += b;
        __syncthreads();
    }
    unsigned bp = s_p[tid] - sp, bi = s_i[tid] - si;   // exclusive bases of this thread's lists
    for (int j = 0; j < per; ++j) {
        const int l = tid * per + j;
        if (l < nlist) {
            const unsigned c = cnt[l];
            pair_start[l] = bp;
            item_start[l] = bi;
            bp += c;
            bi += (c + kIvfNB - 1) / kIvfNB;
            cnt[l] = 0u;
        }
    }
    if (tid == 1023) {
        pair_start[nlist] = s_p[1023];
        item_start[nlist] = s_i[1023];
    }
}
__global__ __launch_bounds__(256) void ivf_pair_scatter_kernel(const u64* __restrict__ probe_keys, int npairs, const unsigned* __restrict__ pair_start,
                                                               unsigned* __restrict__ fill, unsigned* __restrict__ sorted_pairs) {
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (p >= npairs || probe_keys[p] == 0ull) return;
    const uint32_t l = key_row(probe_keys[p]);
    sorted_pairs[pair_start[l] + atomicAdd(&fill[l], 1u)] = (unsigned)p;
}
// one work item per workgroup: item -> (list, its kIvfNB-pair group) by binary search in item_start
template <int DT, int NITER, int SLOTS>
__global__ __launch_bounds__(256) void ivf_scan_shared_kernel(const void* __restrict__ rows_, const uint32_t* __restrict__ ids, const int64_t* __restrict__ offsets,
                                                              const unsigned* __restrict__ pair_start, const unsigned* __restrict__ item_start,
                                                              const unsigned* __restrict__ sorted_pairs, int nlist, int nprobe, int dpad,
                                                              const float* __restrict__ qn, int k, uint32_t row_base, u64* __restrict__ partial) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    const unsigned item = blockIdx.x;
    if (item >= item_start[nlist]) return;   // (the grid is sized for the worst case: one item per pair)
    int lo_l = 0, hi_l = nlist;              // the last list whose item_start <= item
    while (hi_l - lo_l > 1) {
        const int mid = (lo_l + hi_l) >> 1;
        if (item_start[mid] <= item) lo_l = mid; else hi_l = mid;
    }
    const int list = lo_l;
    const unsigned g = item - item_start[list];
    const unsigned p0 = pair_start[list] + g * kIvfNB, p1e = pair_start[list + 1];
    const int nq = (int)((p1e - p0) < (unsigned)kIvfNB ? (p1e - p0) : (unsigned)kIvfNB);
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int nchunks = dpad / E;
    unsigned pair[kIvfNB];
    float qf[kIvfNB][NITER][E];
#pragma unroll
    for (int b = 0; b < kIvfNB; ++b) {
        pair[b] = b < nq ? sorted_pairs[p0 + b] : 0u;
        const int64_t qi = (int64_t)(pair[b] / (unsigned)nprobe);
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int j = lane + kWave * it;
#pragma unroll
            for (int e = 0; e < E; ++e) qf[b][it][e] = (b < nq && j < nchunks) ? qn[qi * dpad + (int64_t)j * E + e] : 0.0f;
        }
    }
    WaveTopK<SLOTS> L[kIvfNB];
#pragma unroll
    for (int b = 0; b < kIvfNB; ++b) L[b].init();
    const int64_t begin = offsets[list], end = offsets[list + 1];
    const uint4* base = reinterpret_cast<const uint4*>(rows_);
    // the rows of step g4 + 16 are on their way while step g4 is scored (raw 16-byte chunks: half the registers of widened rows)
    uint4 cur[4][NITER], nxt[4][NITER];
    auto fetch = [&](int64_t g4, uint4 (&dst)[4][NITER]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int64_t row = g4 + r < end ? g4 + r : end - 1;
            row = row < begin ? begin : row;
            const uint4* p = base + row * (int64_t)nchunks + lane;
#pragma unroll
            for (int it = 0; it < NITER; ++it) dst[r][it] = (lane + kWave * it < nchunks) ? p[kWave * it] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (end > begin) fetch(begin + wave * 4, cur);
    for (int64_t g4 = begin + wave * 4; g4 < end; g4 += 16) {
        if (g4 + 16 < end) fetch(g4 + 16, nxt);
        float a[kIvfNB][4];
#pragma unroll
        for (int b = 0; b < kIvfNB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[b][r] = 0.0f;
#pragma unroll
        for (int it = 0; it < NITER; ++it)   // (it-major as in the per-pair kernel: the same fma order per (row, query), the same bits)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float w[E];
                RT::widen(cur[r][it], w);
#pragma unroll
                for (int b = 0; b < kIvfNB; ++b)
                    if (b < nq) {
#pragma unroll
                        for (int e = 0; e < E; ++e) a[b][r] = __builtin_fmaf(qf[b][it][e], w[e], a[b][r]);
                    }
            }
#pragma unroll
        for (int b = 0; b < kIvfNB; ++b)
            if (b < nq) {
                const float y = butterfly_sum4(a[b][0], a[b][1], a[b][2], a[b][3], lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), 16 * r));
                    if (g4 + r < end) L[b].offer(make_key(sc, row_base + ids[g4 + r]), k, lane);
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int it = 0; it < NITER; ++it) cur[r][it] = nxt[r][it];
    }
    __shared__ u64 lds[4 * kIvfNB * SLOTS * kWave];
#pragma unroll
    for (int b = 0; b < kIvfNB; ++b)
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) lds[((wave * kIvfNB + b) * SLOTS + sl) * kWave + lane] = L[b].v[sl];
    __syncthreads();
    for (int b = wave; b < nq; b += 4) {
        WaveTopK<SLOTS> M;
        M.init();
        for (int wv = 0; wv < 4; ++wv)
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                u64 cand = lds[((wv * kIvfNB + b) * SLOTS + sl) * kWave + lane];
                if (sl * kWave + lane >= k) cand = 0ull;
                M.offer_lanes(cand, k, lane);
            }
        u64* dst = partial + (int64_t)pair[b] * k;   // the pair's own slot: [query][probe rank][k]
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int rank = sl * kWave + lane;
            if (rank < k) dst[rank] = M.v[sl];
        }
    }
}

}  // namespace

// =============================================================================================
// host side: the index object and the C ABI
// =============================================================================================

enum { EV_SCAN = 0, EV_FILTER = 1, EV_SAMPLE = 2, EV_FINALIZE = 3, EV_KINDS = 4 };
static const char* const kEvNames[EV_KINDS] = {"scan", "filter", "sample", "finalize"};

// device control block of one filter pass (zeroed by ONE memset per pass)
struct FilterCtl {
    unsigned hit_cnt[kTileQ * kHitCntStride];
    unsigned flags[FLAG_WORDS];
    unsigned fb_count;         // queries queued for the exact-scan fallback ...
    unsigned fb_done;          // ... blocks of the fallback scan that have finished (the last one merges)
    unsigned sb_ticket;        // small_batch_kernel's arrival counter (its last workgroup puts it back to zero)
    unsigned pad_[1];
    unsigned fb_list[kTileQ];  // ... and which ones
};

// Search workspace.  An index keeps up to kMaxWork of them, one per HIP stream that searches it, so that searches
// issued on different streams (the next batch while this batch's small kernels and its collective are still in
// flight) never share scratch memory.  The index inherits one set of these fields: WorkScope loads the calling
// stream's set into them for the duration of one call and stores it back.
struct WorkBufs {
    // (grown on demand, never inside a captured region after warm-up)
    float* qn = nullptr;       int64_t qn_cap = 0;        // [B][dpad] normalised queries
    u64* partial = nullptr;    int64_t partial_cap = 0;   // [B][blocks][k]
    u64* keys_tmp = nullptr;   int64_t keys_tmp_cap = 0;  // [B][k]
    uint4* qfrag = nullptr;    int64_t qfrag_cap = 0;     // pieces
    float* thr = nullptr;                                  // [512]: thr[256], thr0[256]
    u64* bucket_max = nullptr; int64_t bucket_cap = 0;     // [256][sample tiles] best (score, row) key per sampled tile
    u64* hits = nullptr;       int64_t hits_cap = 0;      // [256][hit_cap_q]
    FilterCtl* ctl = nullptr;                              // device
    u64* fb_partial = nullptr; int64_t fb_partial_cap = 0; // [256][blocks][k] partials of the fallback scan
    uint4* qfrag8 = nullptr;   int64_t qfrag8_cap = 0;    // int8 query fragments (pieces)
    float* qmeta = nullptr;                                // [1024]: per query: scale, 2*eps (device-wide bound), A, B (eps per block = A + B e_block)
    u64* sb_cand = nullptr;    int64_t sb_cand_cap = 0;    // small_batch_kernel: [waves][kSbKeep] published keys, then [waves] dropmax
    u64* probe_keys = nullptr; int64_t probe_cap = 0;      // IVF: [B][nprobe] coarse keys
    u64* ivf_partial = nullptr; int64_t ivf_partial_cap = 0;
    unsigned* ivf_group = nullptr; int64_t ivf_group_cap = 0;  // IVF at batch: [nlist] counters, [nlist+1] pair starts, [nlist+1] item starts, [B*nprobe] pairs by list
};
constexpr int kMaxWork = 4;
struct WorkSlot {
    WorkBufs bufs;
    hipStream_t stream = nullptr;
    hipEvent_t handover = nullptr;  // recorded on the old stream when the slot changes hands
    bool used = false;
    uint64_t tick = 0;              // last use (LRU)
};

struct codd_knn_index : WorkBufs {
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
    // bf16 fragment-order copy, whole 256-row tiles.  Derived data like the int8 shadow: built lazily from the stored rows by the
    // first search that needs it (batches above 256 queries, an index with the int8 filter off or cooling down, the IVF
    // build), brought up to date incrementally over the rows written since
    uint4* shadow = nullptr;
    int64_t shadow_rows = 0;       // rows the shadow allocation covers (multiple of 256)
    int64_t shadow_epoch = -1;
    int64_t dirty16_lo = 0, dirty16_hi = 0;
    hipStream_t shadow_stream = nullptr;
    hipEvent_t shadow_ready = nullptr;
    int64_t stat_shadow_builds = 0;
    int64_t shadow_nomem_epoch = -1;   // row epoch at which allocating the bf16 shadow failed: not retried until rows change
    int64_t stat_shadow_nomem = 0;
    int debug_fail_shadow_alloc = 0;   // test hook ("debug_fail_shadow_alloc"): the next bf16-shadow allocation reports out of memory
    // stored rows written on a caller's stream (upsert_device): searches on other streams wait for this on the device
    hipEvent_t rows_ready = nullptr;
    hipStream_t rows_stream = nullptr;
    bool rows_event_set = false;
    // ... and rows read on a caller's stream outside a search (copy_rows_f32): the next writer on another stream waits for it
    hipEvent_t reader_done = nullptr;
    hipStream_t reader_stream = nullptr;
    bool reader_event_set = false;
    // staging of codd_knn_upsert_host (kept between calls: the indexer job upserts in small batches)
    float* stage_vec = nullptr;   int64_t stage_vec_cap = 0;
    int64_t* stage_slot = nullptr; int64_t stage_slot_cap = 0;
    bool all_normalized = true;    // false once a caller stored rows with normalize = 0

    // filter path knobs
    int filter_enabled = 1;
    int64_t filter_min_rows = 1;             // batches >= filter_min_batch: only the tile-count condition applies
    int64_t filter_min_rows_small = 100000;  // batches below filter_min_batch: filter when rows * B reaches this
    int filter_min_batch = 9;
    int sample_tiles = 4096;  // upper bound on sampled tiles (reached from 42M rows on)
    int sample_div = 40;      // sample about 1/40 of the tiles (2.5 % extra GEMM work), see sample_tile_count()
    int hit_cap_q = 131072;  // candidates kept per query before it falls back to the exact scan (256 MiB per searching stream at 256 queries:
                             // a cluster of 60,000 near-identical rows — closer to each other than either filter's slack — stays on the filter path)

    unsigned long long* dstats = nullptr;                  // device counters: hits, survivors, fallback queries

    // int8 shadow for small batches (optional; rebuilt lazily from the stored rows when they have changed)
    int shadow8_enabled = 1;
    int shadow8_max_batch = 256;  // batches up to this size (one query pass) take the int8 filter
    int resident_q = 1;           // rows of <= 512 int8 elements: the query block stays in LDS ("resident_q" option)
    int i8v2 = 2;                 // batches of 65..256 queries on rows of >= 384 elements (3 K-steps) take i8_tile_kernel (filter_i8.h); 1: only rows of
                                  // more than 512 elements (below, the first-generation kernel keeps the query block resident in LDS: 6-12 % slower); 0: never
    int i8v2_half = 1;            // ... and so do batches of 65..128 queries (its 8-query-block instantiation)
    int ivf_share = 1;            // IVF search: from 1,024 (query, list) pairs on, a probed list is scanned once for all its queries ("ivf_share": 0 = per pair)
    int fuse_fallback = 1;        // batches above 64 queries, k <= 64: finalize and the exact-scan fallback in one launch ("fuse_fallback": 0 = two launches)
    int small_batch_max = 0;      // 1: a single query is answered in one launch by small_batch_kernel where it applies.  OFF by default: measured SLOWER than the
                                  // six-launch chain (1M x 768, B = 1: kernel 0.27 ms, p50 0.33 ms against 0.25 ms; profiles/r3/small_batch_latency.txt)
    int64_t stat_small_batch = 0;
    int per_block = 7;            // the int8 bound per 32-row block: bit 0 in i8_tile_kernel, bit 1 in finalize ("per_block" option; 0 = the device-wide bound everywhere;
                                  // bit 2 chose between block metadata and per-row scales in the tile kernel's filter pass until the per-row path was removed: ignored)
    int i8_pair = 2;              // rows of 6, 12, ... K-steps: the staged tile program with one workgroup barrier per two K-steps ("i8_pair" option: 0 = one per K-step;
                                  // 2 = ... and rows of exactly 6 K-steps its static form, i8_tile_kernel<., 3, ., false>)
    int sample_div8 = 28;         // its thresholds come from a larger sample (the int8 slack is ~5x the bf16 one)
    int sample_rounds8 = 2;       // ... of at least this many rounds of workgroups (one tile each) when the batch has more than 32 queries (3 until the round-3 epilogue:
                                  // at 1.25M rows two rounds trade 13 us of sample for 7 us of filter, gpurun_out/r3n/ab_sample_1p25m.txt)
    uint4* shadow8 = nullptr;
    int64_t shadow8_rows = 0;     // rows the allocation covers (multiple of 256)
    float* rscale = nullptr;      // [shadow8_rows]
    float2* bmeta = nullptr;      // [shadow8_rows / 32]: per 32-row block {scale, largest error norm |c - c~| of its rows}; NaN scale: no row of the block exists
    unsigned* eps_r_bits = nullptr;  // device scalar: max row error norm (float bits)
    int64_t shadow8_epoch = -1;
    int64_t dirty_lo = 0, dirty_hi = 0;  // rows written since the int8 shadow was last brought up to date: [lo, hi)
    hipStream_t shadow8_stream = nullptr;  // the stream the last rebuild ran on, and its completion
    hipEvent_t shadow8_ready = nullptr;
    int64_t stat_shadow8_builds = 0, stat_shadow8_passes = 0, stat_i8v2_passes = 0, stat_f16_tile_passes = 0;
    int f16_tile = 1;             // the 2-byte filter of 129..256 queries over rows of 6, 12, ... 64-element K-steps (768 elements: 12) runs the tile program of
                                  // filter_i8.h on fp16 operands ("f16_tile" option; 0 = gemm_filter_kernel)
    // the worst row's quantisation error, copied back asynchronously after every build: a corpus with badly
    // quantisable rows (one large element, many small ones) would make the int8 bound useless and send every small batch
    // to the exact-scan fallback, so such an index keeps the bf16 filter.  Performance only: never needed for exactness.
    float* eps_r_host = nullptr;        // pinned
    hipEvent_t eps_r_copied = nullptr;
    float eps_r_known = 0.0f;
    int64_t wide_blocks_known = 0;      // blocks whose error norm exceeds shadow8_max_eps, as last read back
    float shadow8_max_eps = 0.04f;
    // Which filter suits the DATA is watched too: on corpora with dense clusters the wider int8 slack lets thousands of
    // rows per query through to the exact re-scoring, where the bf16 filter (a fifth of the slack) is the faster one
    // (profiles/r1/clustered_data_check.txt).  The device counters of the filter passes are copied back
    // asynchronously; when the int8 passes since the last look left more than shadow8_max_surv survivors per query, or
    // sent queries to the fallback, the next shadow8_cooldown searches take the bf16 filter, then the int8 one is
    // tried again.  Performance only: either filter returns the same bits.
    unsigned long long* watch_host = nullptr;  // pinned copy of dstats[0..3]
    hipEvent_t watch_copied = nullptr;
    bool watch_pending = false;
    unsigned long long watch_surv = 0, watch_fb = 0;  // counter values at the last look
    int64_t watch_q8 = 0, watch_q16 = 0;              // queries filtered since then, by path
    int64_t watch_q8_sent = 0, watch_q16_sent = 0;    // ... as of the copy in flight
    int shadow8_max_surv = 4000;
    int shadow8_cooldown = 256;
    int cooldown_left = 0;
    int64_t stat_cooldowns = 0;
    double cooldown_surv8 = 0.0;          // survivors per query of the int8 passes that started the current cooldown
    int64_t cooldown_useless_epoch = -1;  // row epoch at which the bf16 filter was seen to leave as many survivors as the int8 one: no more cooldowns until rows change
    float exp_slack_scale = 1.0f;  // always 1 in the shipped library; "exp_slack_pct" exists only in -DCODD_EXPERIMENTS=1 builds

    // IVF (optional): rows regrouped by coarse list, original slots, list offsets, the coarse index
    codd_knn_index* coarse = nullptr;  // nlist centroids, f32
    void* rows_ivf = nullptr;
    uint32_t* ivf_ids = nullptr;
    int64_t* ivf_offsets = nullptr;    // [nlist + 1]
    int64_t ivf_count = 0;             // rows covered by the IVF layout (must equal count to be fresh)
    int ivf_nlist = 0;
    int64_t ivf_epoch = -1, epoch = 0;  // epoch bumps on every row write; search requires ivf_epoch == epoch

    int64_t stat_searches = 0, stat_scan_launches = 0, stat_last_scan_blocks = 0;
    int64_t stat_filter_passes = 0;

    // optional HIP-event timing of the heavy kernels (bench.py's roofline figure): one (start, stop)
    // pair per launch, recorded on the launch stream, read after a sync
    bool profile = false;
    std::vector<hipEvent_t> ev;  // 2 * pairs
    std::vector<int> ev_kind;
    int ev_used = 0;

    WorkSlot slots[kMaxWork];
    uint64_t tick = 0;
    std::mutex mu;  // one host thread at a time enqueues a search (the launches themselves are asynchronous)
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

// every row write: bumps the epoch (IVF / int8 shadow staleness) and widens the dirty row range
void rows_written(codd_knn_index* ix, int64_t lo, int64_t hi) {
    ix->epoch++;
    if (ix->dirty_hi <= ix->dirty_lo) { ix->dirty_lo = lo; ix->dirty_hi = hi; }
    else {
        if (lo < ix->dirty_lo) ix->dirty_lo = lo;
        if (hi > ix->dirty_hi) ix->dirty_hi = hi;
    }
    if (ix->dirty16_hi <= ix->dirty16_lo) { ix->dirty16_lo = lo; ix->dirty16_hi = hi; }
    else {
        if (lo < ix->dirty16_lo) ix->dirty16_lo = lo;
        if (hi > ix->dirty16_hi) ix->dirty16_hi = hi;
    }
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
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

struct EvScope {  // records a (start, stop) pair around one launch when profiling is on
    codd_knn_index* ix;
    hipStream_t st;
    int slot = -1;
    EvScope(codd_knn_index* ix_, int kind, hipStream_t st_) : ix(ix_), st(st_) {
        if (ix->profile && 2 * (ix->ev_used + 1) <= (int)ix->ev.size()) {
            slot = ix->ev_used++;
            ix->ev_kind[slot] = kind;
            (void)hipEventRecord(ix->ev[2 * slot], st);
        }
    }
    ~EvScope() {
        if (slot >= 0) (void)hipEventRecord(ix->ev[2 * slot + 1], st);
    }
};

// Loads the calling stream's workspace into the index for one call (see WorkBufs).  A stream keeps its slot; a new
// stream takes a free slot, or the least recently used one after making itself wait for everything the previous
// owner has enqueued so far (no host synchronisation; if that stream no longer exists the device is drained instead).
struct WorkScope {
    codd_knn_index* ix;
    int slot = 0;
    WorkScope(codd_knn_index* ix_, hipStream_t st) : ix(ix_) {
        ix->mu.lock();
        int free_slot = -1, lru = 0;
        slot = -1;
        for (int i = 0; i < kMaxWork; ++i) {
            WorkSlot& w = ix->slots[i];
            if (w.used && w.stream == st) { slot = i; break; }
            if (!w.used && free_slot < 0) free_slot = i;
            if (w.used && ix->slots[lru].used && w.tick < ix->slots[lru].tick) lru = i;
        }
        if (slot < 0 && free_slot >= 0) slot = free_slot;
        if (slot < 0) {
            slot = lru;
            WorkSlot& w = ix->slots[slot];
            bool ordered = false;
            if (!w.handover) (void)hipEventCreateWithFlags(&w.handover, hipEventDisableTiming);
            if (w.handover && hipEventRecord(w.handover, w.stream) == hipSuccess) ordered = hipStreamWaitEvent(st, w.handover, 0) == hipSuccess;
            if (!ordered) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
        }
        WorkSlot& w = ix->slots[slot];
        w.used = true;
        w.stream = st;
        w.tick = ++ix->tick;
        static_cast<WorkBufs&>(*ix) = w.bufs;
    }
    ~WorkScope() {
        ix->slots[slot].bufs = static_cast<WorkBufs&>(*ix);
        static_cast<WorkBufs&>(*ix) = WorkBufs();
        ix->mu.unlock();
    }
    WorkScope(const WorkScope&) = delete;
    WorkScope& operator=(const WorkScope&) = delete;
};

size_t elem_size(int dtype) { return dtype == DT_F32 ? 4 : 2; }
int elems_per_chunk(int dtype) { return dtype == DT_F32 ? 4 : 8; }

// (re)allocate row storage for at least `need` slots, preserving contents (the shadows are derived data with their own
// allocations: ensure_shadow / ensure_shadow8)
int grow_rows(codd_knn_index* ix, int64_t need, bool exact) {
    if (need <= ix->capacity) return CODD_KNN_OK;
    int64_t cap = need;
    if (!exact) {
        cap = ix->capacity > 0 ? ix->capacity : 1024;
        while (cap < need) cap += cap / 2 + 1024;
    }
    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    void* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, (size_t)cap * row_bytes));
    hipError_t e = hipMemset(fresh, 0, (size_t)cap * row_bytes);
    if (e == hipSuccess && ix->rows && ix->count > 0) e = hipMemcpy(fresh, ix->rows, (size_t)ix->count * row_bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
        (void)hipFree(fresh);
        return fail(CODD_KNN_EDEVICE, "row copy on growth failed: %s", hipGetErrorString(e));
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
                     int64_t first_slot, void* out, uint2* shadow, hipStream_t st) {
    if (n <= 0) return CODD_KNN_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    switch (dtype) {
        case DT_F32:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_F32>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out, shadow);
            break;
        case DT_BF16:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_BF16>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out, shadow);
            break;
        case DT_F16:
            hipLaunchKernelGGL(normalize_rows_kernel<DT_F16>, grid, block, 0, st, in, n, d, dpad, normalize, slots, first_slot, out, shadow);
            break;
        default:
            return fail(CODD_KNN_EINVAL, "unknown dtype%s");
    }
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

// ---- exact scan dispatch ---------------------------------------------------------------------

struct ScanArgs {
    const void* rows;
    int64_t n;
    int dpad;
    const float* qn;
    int nq, k;
    uint32_t row_base;
    u64* partial;
    int64_t stride_q;
    const unsigned* qlist = nullptr;   // LISTED mode (device-side fallback queue)
    const unsigned* qcount = nullptr;
    unsigned* merge_done = nullptr;    // LISTED mode: the last block merges and writes the answers (one launch)
    u64* merged_keys = nullptr;
    float* merged_dist = nullptr;
    int64_t* merged_rows = nullptr;
    unsigned long long* count_total = nullptr;
};

template <int DT, int NB, int NITER>
void launch_scan_slots(int slots, dim3 grid, size_t lds, hipStream_t st, const ScanArgs& a) {
    if (slots == 1)
        hipLaunchKernelGGL((scan_topk_kernel<DT, NB, NITER, 1>), grid, dim3(256), lds, st, a.rows, a.n, a.dpad, a.qn, a.nq, a.k, a.row_base, a.partial, a.stride_q, a.qlist, a.qcount, a.merge_done, a.merged_keys, a.merged_dist, a.merged_rows, a.count_total);
    else
        hipLaunchKernelGGL((scan_topk_kernel<DT, NB, NITER, 2>), grid, dim3(256), lds, st, a.rows, a.n, a.dpad, a.qn, a.nq, a.k, a.row_base, a.partial, a.stride_q, a.qlist, a.qcount, a.merge_done, a.merged_keys, a.merged_dist, a.merged_rows, a.count_total);
}

template <int DT, int NB>
int launch_scan_niter(int niter, int slots, dim3 grid, size_t lds, hipStream_t st, const ScanArgs& a) {
    switch (niter) {
        case 1: launch_scan_slots<DT, NB, 1>(slots, grid, lds, st, a); break;
        case 2: launch_scan_slots<DT, NB, 2>(slots, grid, lds, st, a); break;
        case 3: launch_scan_slots<DT, NB, 3>(slots, grid, lds, st, a); break;
        case 4: launch_scan_slots<DT, NB, 4>(slots, grid, lds, st, a); break;
        default: return fail(CODD_KNN_ENOTSUP, "row too wide for the scan kernel%s");
    }
    return CODD_KNN_OK;
}

template <int DT>
int launch_scan_nb(int nb, int niter, int slots, dim3 grid, hipStream_t st, const ScanArgs& a) {
    const size_t lds = (size_t)4 * nb * slots * kWave * sizeof(u64);
    switch (nb) {
        case 1: return launch_scan_niter<DT, 1>(niter, slots, grid, lds, st, a);
        case 4: return launch_scan_niter<DT, 4>(niter, slots, grid, lds, st, a);
        case 8: return launch_scan_niter<DT, 8>(niter, slots, grid, lds, st, a);
        default: return fail(CODD_KNN_EINVAL, "bad query group%s");
    }
}

int launch_merge(const u64* in, int B, int64_t m, int64_t in_stride, int k, u64* out_keys, float* out_dist, int64_t* out_rows,
                 hipStream_t st, const unsigned* out_list = nullptr, const unsigned* count_ptr = nullptr,
                 unsigned long long* count_total = nullptr, int64_t seg_len = 0, int64_t seg_stride = 0) {
    if (seg_len <= 0) seg_len = m > 0 ? m : 1;
    if (k <= 64)
        hipLaunchKernelGGL(merge_keys_kernel<1>, dim3(B), dim3(256), 0, st, in, m, in_stride, seg_len, seg_stride, k, out_keys, out_dist, out_rows,
                           out_list, count_ptr, count_total);
    else
        hipLaunchKernelGGL(merge_keys_kernel<2>, dim3(B), dim3(256), 0, st, in, m, in_stride, seg_len, seg_stride, k, out_keys, out_dist, out_rows,
                           out_list, count_ptr, count_total);
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

int scan_geometry(const codd_knn_index* ix, int64_t n, int* niter, int64_t* blocks) {
    const int nchunks = ix->dpad / elems_per_chunk(ix->dtype);
    *niter = (nchunks + kWave - 1) / kWave;
    if (*niter > 4) return fail(CODD_KNN_ENOTSUP, "dim too large for this dtype (f32 <= 1024, bf16/f16 <= 2048)%s");
    const int64_t ngroups = (n + 3) / 4;
    int64_t b = (ngroups + 3) / 4;
    const int64_t cap_blocks = (int64_t)ix->num_cus * ix->scan_blocks_per_cu;
    if (b > cap_blocks) b = cap_blocks;
    if (b < 1) b = 1;
    *blocks = b;
    return CODD_KNN_OK;
}

// exact scan of `nqueries` dense normalised queries -> keys_out[nqueries][k]
int exact_scan(codd_knn_index* ix, const float* qn, int nqueries, int k, uint32_t row_base, u64* keys_out, float* dist_out,
               int64_t* rows_out, hipStream_t st) {
    const int64_t n = ix->count;
    int niter;
    int64_t blocks;
    int rc = scan_geometry(ix, n, &niter, &blocks);
    if (rc != 0) return rc;
    ix->stat_last_scan_blocks = blocks;
    const int slots = k <= 64 ? 1 : 2;
    const int64_t stride_q = blocks * k;
    if ((rc = ensure_buf(&ix->partial, &ix->partial_cap, (int64_t)nqueries * stride_q)) != 0) return rc;
    {
        // ONE launch: up to 8 queries ride along per pass over the rows; a larger batch loops over groups
        // of 8 inside the kernel (small corpora stay L2-resident across the groups, and a batch costs one
        // launch instead of B/8)
        const int nb = nqueries == 1 ? 1 : (nqueries <= 4 ? 4 : 8);
        ScanArgs a{ix->rows, n, ix->dpad, qn, nqueries, k, row_base, ix->partial, stride_q};
        const dim3 grid((unsigned)blocks);
        {
            EvScope ev(ix, EV_SCAN, st);
            switch (ix->dtype) {
                case DT_F32: rc = launch_scan_nb<DT_F32>(nb, niter, slots, grid, st, a); break;
                case DT_BF16: rc = launch_scan_nb<DT_BF16>(nb, niter, slots, grid, st, a); break;
                case DT_F16: rc = launch_scan_nb<DT_F16>(nb, niter, slots, grid, st, a); break;
                default: rc = fail(CODD_KNN_EINVAL, "unknown dtype%s");
            }
        }
        if (rc != 0) return rc;
        HIP_TRY(hipGetLastError());
        ix->stat_scan_launches++;
    }
    return launch_merge(ix->partial, nqueries, stride_q, stride_q, k, keys_out, dist_out, rows_out, st);
}

// ---- filter path -----------------------------------------------------------------------------

float filter_eps(const codd_knn_index* ix) {
    // |approx - exact| for unit-norm q and c.  u = unit roundoff of the shadow element type under round-to-nearest:
    // half the spacing of the significand grid relative to the value (bf16: 8 significant bits, spacing 2^-7 -> u = 2^-8;
    // fp16: 11 significant bits -> u = 2^-11).  Round 1 shipped 2^-9 for bf16, half the true bound: a unit vector of 239
    // equal entries scores 0.99286 against itself in bf16 (error 0.0071 > the old eps 0.0040):
    //   rounding  : (2u + u^2) when q and the stored row are both rounded (Cauchy-Schwarz over the
    //               element-wise relative errors); u when the stored rows already are exactly
    //               representable (bf16 rows under a bf16 shadow, f16 rows under an fp16 shadow);
    //   subnormal : fp16 only — below 2^-14 the grid is absolute (2^-25 per element, <= sqrt(dpad) * 2^-25
    //               per operand after Cauchy-Schwarz against a unit vector);
    //   summation : two fp32 accumulations of <= dpad terms whose magnitudes sum to <= 1: dpad * 2^-24 each.
#if CODD_SHADOW_F16
    const float u = 4.8828125e-4f;
    const bool exact_rows = ix->dtype == DT_F16;
    const float subnormal = 2.0f * sqrtf((float)ix->dpad) * 2.9802322e-8f;
#else
    const float u = 0.00390625f;
    const bool exact_rows = ix->dtype == DT_BF16;
    const float subnormal = 0.0f;
#endif
    const float rounding = exact_rows ? u : 2.0f * u + u * u;
    const float summation = (float)ix->dpad * 1.1920929e-7f;  // dpad * 2^-23
    return (rounding + subnormal + summation) * 1.001f;
}

template <int DT, int NITER>
void launch_finalize_slots(int slots, int B, hipStream_t st, const codd_knn_index* ix, const float* qn, int k, float two_eps,
                           uint32_t row_base, u64* out_keys, const float* two_eps_q, int nparts, u64* part_keys, float* out_dist, int64_t* out_rows) {
    FilterCtl* c = ix->ctl;
    const float2* bm = two_eps_q && (ix->per_block & 2) ? ix->bmeta : nullptr;  // (the int8 passes: slack per 32-row block)
    if (slots == 1)
        hipLaunchKernelGGL((finalize_kernel<DT, NITER, 1>), dim3(B, nparts), dim3(kFinThreads), 0, st, ix->rows, ix->dpad, qn, ix->hits, c->hit_cnt,
                           ix->hit_cap_q, c->flags, k, two_eps, row_base, out_keys, &c->fb_count, c->fb_list, ix->dstats, two_eps_q, part_keys, out_dist, out_rows, bm);
    else
        hipLaunchKernelGGL((finalize_kernel<DT, NITER, 2>), dim3(B, nparts), dim3(kFinThreads), 0, st, ix->rows, ix->dpad, qn, ix->hits, c->hit_cnt,
                           ix->hit_cap_q, c->flags, k, two_eps, row_base, out_keys, &c->fb_count, c->fb_list, ix->dstats, two_eps_q, part_keys, out_dist, out_rows, bm);
}

template <int DT>
int launch_finalize(int niter, int slots, int B, hipStream_t st, const codd_knn_index* ix, const float* qn, int k, float two_eps,
                    uint32_t row_base, u64* out_keys, const float* two_eps_q = nullptr, int nparts = 1, u64* part_keys = nullptr,
                    float* out_dist = nullptr, int64_t* out_rows = nullptr) {
    switch (niter) {
        case 1: launch_finalize_slots<DT, 1>(slots, B, st, ix, qn, k, two_eps, row_base, out_keys, two_eps_q, nparts, part_keys, out_dist, out_rows); break;
        case 2: launch_finalize_slots<DT, 2>(slots, B, st, ix, qn, k, two_eps, row_base, out_keys, two_eps_q, nparts, part_keys, out_dist, out_rows); break;
        case 3: launch_finalize_slots<DT, 3>(slots, B, st, ix, qn, k, two_eps, row_base, out_keys, two_eps_q, nparts, part_keys, out_dist, out_rows); break;
        case 4: launch_finalize_slots<DT, 4>(slots, B, st, ix, qn, k, two_eps, row_base, out_keys, two_eps_q, nparts, part_keys, out_dist, out_rows); break;
        default: return fail(CODD_KNN_ENOTSUP, "row too wide for the finalize kernel%s");
    }
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

size_t filter_lds_bytes(int mode) {
    return (size_t)kLdsQBytes + kLdsWords * 4 + (mode == MODE_FILTER ? (size_t)kHitCap * 12 : 0);
}

int ensure_filter_workspace(codd_knn_index* ix) {
    int rc;
    if ((rc = ensure_buf(&ix->qfrag, &ix->qfrag_cap, (int64_t)kTileQ * (ix->dpad / 8))) != 0) return rc;
    if ((rc = ensure_buf(&ix->bucket_max, &ix->bucket_cap, (int64_t)ix->sample_tiles * kTileQ)) != 0) return rc;
    if ((rc = ensure_buf(&ix->hits, &ix->hits_cap, (int64_t)kTileQ * ix->hit_cap_q)) != 0) return rc;
    if (!ix->thr) HIP_TRY(hipMalloc((void**)&ix->thr, 2 * kTileQ * sizeof(float)));  // thr[q], then thr0[q] (per-block form, int8 tile kernel)
    if (!ix->ctl) {
        HIP_TRY(hipMalloc((void**)&ix->ctl, sizeof(FilterCtl)));
        HIP_TRY(hipMemset(ix->ctl, 0, sizeof(FilterCtl)));  // (the filter passes clear it per pass; small_batch_kernel relies on zeros left behind)
    }
    if (!ix->dstats) {
        HIP_TRY(hipMalloc((void**)&ix->dstats, 4 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(ix->dstats, 0, 4 * sizeof(unsigned long long)));
    }
    static std::atomic<bool> attr_set{false};  // dynamic LDS above 64 KiB needs the opt-in (idempotent)
    if (!attr_set.load(std::memory_order_acquire)) {
        const void* fns[] = {
            (const void*)&gemm_filter_kernel<MODE_FILTER, 1>, (const void*)&gemm_filter_kernel<MODE_FILTER, 2>,
            (const void*)&gemm_filter_kernel<MODE_FILTER, 4>, (const void*)&gemm_filter_kernel<MODE_FILTER, 8>,
            (const void*)&gemm_filter_kernel<MODE_SAMPLE, 1>, (const void*)&gemm_filter_kernel<MODE_SAMPLE, 2>,
            (const void*)&gemm_filter_kernel<MODE_SAMPLE, 4>, (const void*)&gemm_filter_kernel<MODE_SAMPLE, 8>,
            (const void*)&gemm_filter_kernel<MODE_DUMP, 8>};
        for (const void* fn : fns)
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)filter_lds_bytes(MODE_FILTER)));
        const void* tile_fns[] = {(const void*)&i8_tile_kernel<MODE_FILTER, 0>, (const void*)&i8_tile_kernel<MODE_FILTER, 1>, (const void*)&i8_tile_kernel<MODE_FILTER, 2>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 0, 8>, (const void*)&i8_tile_kernel<MODE_FILTER, 1, 8>, (const void*)&i8_tile_kernel<MODE_FILTER, 2, 8>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 3>, (const void*)&i8_tile_kernel<MODE_FILTER, 3, 8>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 2, 16, false, true>, (const void*)&i8_tile_kernel<MODE_FILTER, 3, 16, false, true>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 4, 16, false, true>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 1, 16, true>, (const void*)&i8_tile_kernel<MODE_FILTER, 0, 16, true>,
                                  (const void*)&i8_tile_kernel<MODE_FILTER, 1, 8, true>, (const void*)&i8_tile_kernel<MODE_FILTER, 0, 8, true>};
        for (const void* fn : tile_fns) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)i8_lds_bytes(MODE_FILTER)));
        const void* tile_sample_fns[] = {(const void*)&i8_tile_kernel<MODE_SAMPLE, 0>, (const void*)&i8_tile_kernel<MODE_SAMPLE, 0, 8>,
                                         (const void*)&i8_tile_kernel<MODE_SAMPLE, 0, 16, true>, (const void*)&i8_tile_kernel<MODE_SAMPLE, 0, 8, true>};
        for (const void* fn : tile_sample_fns) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)i8_lds_bytes(MODE_SAMPLE)));
        attr_set.store(true, std::memory_order_release);
    }
    return CODD_KNN_OK;
}

// how many evenly spaced tiles set the thresholds: ~ntiles/sample_div (so the sample costs a fixed
// fraction of the main pass at every shard size and the hit volume per query stays ~k*sample_div),
// at least max(64, 4k) where the corpus has that many tiles, at most sample_tiles
int64_t sample_tile_count(const codd_knn_index* ix, int64_t ntiles, int k, bool use8 = false, int nbq = 8) {
    int64_t ts = ntiles / (use8 ? (nbq == 1 ? 2 * ix->sample_div8 : ix->sample_div8) : ix->sample_div);
    if (use8) {
        // the int8 filter pays more per hit (wider slack, more of them) and less per sampled tile: "sample_rounds8" rounds of
        // workgroups (2 since round 3's cheaper epilogue; 3 before) where that is still under a quarter of the corpus
        // (1/10 of a 1.25M-row shard, 1/28 of 10M rows; twice as sparse for <= 32 queries, whose sample is a pure byte stream)
        const int64_t rounds = (nbq == 1 ? 1 : ix->sample_rounds8) * (int64_t)ix->num_cus;
        const int64_t floor8 = rounds < ntiles / 4 ? rounds : ntiles / 4;
        if (ts < floor8) ts = floor8;
    }
    const int64_t lo = 4 * (int64_t)k > 64 ? 4 * (int64_t)k : 64;
    if (ts < lo) ts = lo;
    // a sample of fewer tiles than there are CUs takes as long as one full round of workgroups (one tile each), and a
    // bigger sample means a tighter threshold: fewer hits for the filter's epilogue (scripts/rows_sweep.py)
    if (ts < ix->num_cus && ntiles >= 2 * (int64_t)ix->num_cus) ts = ix->num_cus;
    // whole rounds of workgroups: the last round costs a round whether it is full or not
    if (ts > ix->num_cus) ts = (ts + ix->num_cus - 1) / ix->num_cus * ix->num_cus;
    if (ts > ix->sample_tiles) ts = ix->sample_tiles;
    if (ts > ntiles) ts = ntiles;
    return ts < 1 ? 1 : ts;
}

int dpad8_of(const codd_knn_index* ix) { return (ix->dpad + 127) / 128 * 128; }

// Searches on a stream other than the one rows were last written on (codd_knn_upsert_device is asynchronous on the caller's
// stream) wait for that write on the device before they read rows or rebuild a shadow from them.
int wait_rows(codd_knn_index* ix, hipStream_t st) {
    if (ix->rows_event_set && st != ix->rows_stream) HIP_TRY(hipStreamWaitEvent(st, ix->rows_ready, 0));
    return CODD_KNN_OK;
}

// Before a derived buffer is freed (a shadow outgrown by the corpus): wait on the host for the streams that search THIS index
// — what they have enqueued so far may still read the old allocation — not for the whole device.  (Growth is the one
// place where a search call blocks; steady-state searches never do.)
int wait_searching_streams(codd_knn_index* ix) {
    for (WorkSlot& w : ix->slots) {
        if (!w.used) continue;
        if (!w.handover) HIP_TRY(hipEventCreateWithFlags(&w.handover, hipEventDisableTiming));
        if (hipEventRecord(w.handover, w.stream) == hipSuccess) HIP_TRY(hipEventSynchronize(w.handover));
        else (void)hipGetLastError();  // (a stream that no longer exists has nothing in flight)
    }
    return CODD_KNN_OK;
}

// The bf16 shadow, like the int8 one, is derived data: (re)built from the stored rows on the searching stream whenever rows
// have been written since the last build, over the dirty row range only.  An index that only ever sees batches of <= 256
// queries (the int8 filter) never allocates it: 38 GB instead of 54 GB for the 10M x 768 fp32 corpus.
int ensure_shadow(codd_knn_index* ix, hipStream_t st) {
    const int64_t n = ix->count;
    if (ix->shadow_epoch == ix->epoch && ix->shadow) {
        if (st != ix->shadow_stream && ix->shadow_ready) HIP_TRY(hipStreamWaitEvent(st, ix->shadow_ready, 0));
        return CODD_KNN_OK;
    }
    const int64_t need = (n + kTileRows - 1) / kTileRows * kTileRows;
    int64_t first = ix->dirty16_lo, m = ix->dirty16_hi - ix->dirty16_lo;
    if (need > ix->shadow_rows) {
        // A failed allocation is remembered until rows change (no multi-GB hipMalloc retried by every search); the caller
        // answers the search another way (search_impl: the int8 filter or the exact scan).
        if (ix->shadow_nomem_epoch == ix->epoch) return CODD_KNN_ENOMEM;
        int rc;
        if ((rc = wait_searching_streams(ix)) != 0) return rc;
        if (ix->shadow) (void)hipFree(ix->shadow);
        ix->shadow = nullptr; ix->shadow_rows = 0;
        const int64_t rows = need + need / 8;
        const int64_t rows_al = (rows + kTileRows - 1) / kTileRows * kTileRows;
        hipError_t me = ix->debug_fail_shadow_alloc ? hipErrorOutOfMemory : hipMalloc((void**)&ix->shadow, (size_t)rows_al * ix->dpad * 2);
        if (me != hipSuccess) {
            (void)hipGetLastError();
            ix->shadow = nullptr;
            ix->shadow_nomem_epoch = ix->epoch;
            ix->stat_shadow_nomem++;
            return fail(CODD_KNN_ENOMEM, "bf16 shadow: %s", hipGetErrorString(me));
        }
        ix->shadow_rows = rows_al;
        HIP_TRY(hipMemsetAsync(ix->shadow, 0, (size_t)rows_al * ix->dpad * 2, st));  // rows beyond the count are masked, their bytes only have to be defined
        first = 0; m = n;
    }
    if (!ix->shadow_ready) HIP_TRY(hipEventCreateWithFlags(&ix->shadow_ready, hipEventDisableTiming));
    if (m > 0) {
        const dim3 grid((unsigned)((m + 3) / 4)), block(256);
        uint2* sh = reinterpret_cast<uint2*>(ix->shadow);
        switch (ix->dtype) {
            case DT_F32: hipLaunchKernelGGL(shadow_from_rows_kernel<DT_F32>, grid, block, 0, st, ix->rows, first, m, ix->dpad, sh); break;
            case DT_BF16: hipLaunchKernelGGL(shadow_from_rows_kernel<DT_BF16>, grid, block, 0, st, ix->rows, first, m, ix->dpad, sh); break;
            default: hipLaunchKernelGGL(shadow_from_rows_kernel<DT_F16>, grid, block, 0, st, ix->rows, first, m, ix->dpad, sh); break;
        }
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ix->shadow_ready, st));
    ix->shadow_stream = st;
    ix->shadow_epoch = ix->epoch;
    ix->dirty16_lo = ix->dirty16_hi = 0;
    ix->stat_shadow_builds++;
    return CODD_KNN_OK;
}

// The int8 shadow is derived data: (re)built from the stored rows on the searching stream whenever rows have been
// written since the last build.  Other streams that search before the build has finished wait for it on the device.
int ensure_shadow8(codd_knn_index* ix, hipStream_t st) {
    const int64_t n = ix->count;
    const int dpad8 = dpad8_of(ix);
    if (ix->shadow8_epoch == ix->epoch && ix->shadow8) {
        if (st != ix->shadow8_stream && ix->shadow8_ready) HIP_TRY(hipStreamWaitEvent(st, ix->shadow8_ready, 0));
        return CODD_KNN_OK;
    }
    const int64_t need = (n + kTileRows - 1) / kTileRows * kTileRows;
    int64_t first = ix->dirty_lo, m = ix->dirty_hi - ix->dirty_lo;  // rows to (re)quantise
    if (!ix->eps_r_bits) {
        HIP_TRY(hipMalloc((void**)&ix->eps_r_bits, 2 * sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(ix->eps_r_bits, 0, 2 * sizeof(unsigned), st));
    }
    if (need > ix->shadow8_rows) {
        // (every stream that may still read the old allocation has to be done with it)
        int rcw;
        if ((rcw = wait_searching_streams(ix)) != 0) return rcw;
        if (ix->shadow8) (void)hipFree(ix->shadow8);
        if (ix->rscale) (void)hipFree(ix->rscale);
        if (ix->bmeta) (void)hipFree(ix->bmeta);
        ix->shadow8 = nullptr; ix->rscale = nullptr; ix->bmeta = nullptr; ix->shadow8_rows = 0;
        const int64_t rows = need + need / 8;  // head room: appends do not reallocate every time
        const int64_t rows_al = (rows + kTileRows - 1) / kTileRows * kTileRows;
        HIP_TRY(hipMalloc((void**)&ix->shadow8, (size_t)rows_al * dpad8));
        HIP_TRY(hipMalloc((void**)&ix->rscale, (size_t)rows_al * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&ix->bmeta, (size_t)(rows_al / 32 + 64) * sizeof(float2)));  // (+64: a wave's DMA reads 32 blocks from a tile's first one)
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ix->bmeta, (int)0x7fc00000, (size_t)(rows_al / 32 + 64) * 2, st));  // NaN: no such block yet
        ix->shadow8_rows = rows_al;
        // rows beyond the count are masked (row < n), their bytes only have to be defined
        HIP_TRY(hipMemsetAsync(ix->shadow8, 0, (size_t)rows_al * dpad8, st));
        HIP_TRY(hipMemsetAsync(ix->eps_r_bits, 0, 2 * sizeof(unsigned), st));
        first = 0; m = n;  // a fresh allocation holds nothing yet
    }
    if (!ix->shadow8_ready) HIP_TRY(hipEventCreateWithFlags(&ix->shadow8_ready, hipEventDisableTiming));
    if (m > 0) {
        // Whole 32-row blocks (one scale per block): the rows that share a block with the dirty range are quantised again,
        // with the block's new scale, and the block's error norm is measured again.
        const int64_t last = first + m;
        first = first / 32 * 32;
        m = (last + 31) / 32 * 32 - first;
        const dim3 grid((unsigned)(m / 32)), block(256);
        uint4* s8 = ix->shadow8;
        switch (ix->dtype) {
            case DT_F32: hipLaunchKernelGGL(shadow8_from_rows_kernel<DT_F32>, grid, block, 0, st, ix->rows, first, n, ix->dpad, dpad8, s8, ix->rscale, ix->bmeta); break;
            case DT_BF16: hipLaunchKernelGGL(shadow8_from_rows_kernel<DT_BF16>, grid, block, 0, st, ix->rows, first, n, ix->dpad, dpad8, s8, ix->rscale, ix->bmeta); break;
            default: hipLaunchKernelGGL(shadow8_from_rows_kernel<DT_F16>, grid, block, 0, st, ix->rows, first, n, ix->dpad, dpad8, s8, ix->rscale, ix->bmeta); break;
        }
        HIP_TRY(hipGetLastError());
        // the device-wide maximum of the block error norms, over the rows stored now (not a running maximum)
        hipLaunchKernelGGL(eps_max_kernel, dim3(1), dim3(1024), 0, st, ix->bmeta, (n + 31) / 32, ix->eps_r_bits, ix->shadow8_max_eps);
        HIP_TRY(hipGetLastError());
    }
    {
        // i8_tile_kernel reads whole tiles of scales: rows past the count carry NaN (`acc * NaN >= thr` is false for every
        // threshold, so its epilogue needs no row < count test).  Rows appended later get real scales from the build above.
        const int64_t pad = need - n;
        if (pad > 0) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ix->rscale + n), (int)0x7fc00000, (size_t)pad, st));
    }
    HIP_TRY(hipEventRecord(ix->shadow8_ready, st));
    if (!ix->eps_r_host) {
        HIP_TRY(hipHostMalloc((void**)&ix->eps_r_host, 2 * sizeof(float), hipHostMallocDefault));
        ix->eps_r_host[0] = 0.0f;
        ix->eps_r_host[1] = 0.0f;
        HIP_TRY(hipEventCreateWithFlags(&ix->eps_r_copied, hipEventDisableTiming));
    }
    HIP_TRY(hipMemcpyAsync(ix->eps_r_host, ix->eps_r_bits, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(ix->eps_r_copied, st));
    ix->shadow8_stream = st;
    ix->shadow8_epoch = ix->epoch;
    ix->dirty_lo = ix->dirty_hi = 0;
    ix->stat_shadow8_builds++;
    return CODD_KNN_OK;
}

// one pass of <= 256 queries through sample -> threshold -> filter -> finalize (+ exact fallback)
// keys_out and / or (dist_out, rows_out): what the caller wants written per query (any may be null)
int filter_pass(codd_knn_index* ix, const float* qn, int nq, int k, uint32_t row_base, u64* keys_out, float* dist_out, int64_t* rows_out,
                hipStream_t st, bool prepared = false, bool use8 = false) {
    const int64_t n = ix->count;
    const int nsteps = use8 ? dpad8_of(ix) / 128 : ix->dpad / 64;
    const uint4* shadow = use8 ? ix->shadow8 : ix->shadow;
    const uint4* qfrag = use8 ? ix->qfrag8 : ix->qfrag;
    const float* slack_q = use8 ? ix->qmeta + 256 : nullptr;  // int8: 2*eps per query, written by prep_queries8_kernel
    if (use8) ix->stat_shadow8_passes++;
    const bool resident = use8 && nsteps <= 4 && ix->resident_q;
    // 65..128 queries only: with <= 64 the kernel is a byte stream (nothing to gain); with 8 query blocks the six-step body
    // needs more than the 256 registers of a wave (hipcc spills, no gain measured)
    const bool partial6 = use8 && nsteps == 6 && nq > 64 && nq <= 128 && ix->resident_q;
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    const int slots = k <= 64 ? 1 : 2;
    const float eps = filter_eps(ix);
    int rc;
    ix->stat_filter_passes++;

    if (!prepared) {  // (a batch of <= 256 queries arrives with its fragments and a cleared control block: prep_queries_kernel)
        hipLaunchKernelGGL(qfrag_kernel, dim3((kTileQ * (ix->dpad / 8) + 255) / 256), dim3(256), 0, st, qn, nq, ix->dpad, ix->qfrag,
                           reinterpret_cast<unsigned*>(ix->ctl), (int)(sizeof(FilterCtl) / 4));
        HIP_TRY(hipGetLastError());
    }

    // sample: every `stride`-th tile
    const int nbq = nq <= 32 ? 1 : (nq <= 64 ? 2 : (nq <= 128 ? 4 : 8));  // 32-query blocks the GEMM multiplies
    // full query blocks: the second-generation int8 kernel (filter_i8.h); rows whose query block fits the LDS keep the resident one
    const bool tile_v2 = use8 && (nbq == 8 || (nbq == 4 && ix->i8v2_half)) && ((ix->i8v2 == 1 && !resident && nsteps >= 3) || (ix->i8v2 == 2 && nsteps >= 3));
    if (tile_v2) ix->stat_i8v2_passes++;
    // the 2-byte filter (dense clusters the int8 slack cannot separate, the int8 filter switched off): the same tile program on fp16 operands
    const bool tile_f16 = CODD_SHADOW_F16 && CODD_MFMA16 && !use8 && nbq == 8 && nsteps % 6 == 0 && ix->f16_tile;
    if (tile_f16) ix->stat_f16_tile_passes++;
    const int64_t ts = sample_tile_count(ix, ntiles, k, use8, nbq);
    const int64_t stride = ntiles / ts;
    {
        EvScope ev(ix, EV_SAMPLE, st);
        const dim3 g((unsigned)(ts < ix->num_cus ? ts : ix->num_cus)), b(kFilterThreads);
        const size_t lds = filter_lds_bytes(MODE_SAMPLE);
#define CODD_LAUNCH_SAMPLE(NBQ)                                                                                              \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_SAMPLE, NBQ>), g, b, lds, st, shadow, qfrag, n, nsteps, ts, stride, \
                       nullptr, ix->bucket_max, nullptr, nullptr, 0, nullptr, nullptr)
#define CODD_LAUNCH_SAMPLE8(NBQ)                                                                                             \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_SAMPLE, NBQ, 1>), g, b, lds, st, shadow, qfrag, n, nsteps, ts, stride, \
                       nullptr, ix->bucket_max, nullptr, nullptr, 0, nullptr, nullptr, ix->rscale, ix->qmeta)
#define CODD_LAUNCH_SAMPLE8R(NBQ)                                                                                            \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_SAMPLE, NBQ, 1, 1>), g, b, lds, st, shadow, qfrag, n, nsteps, ts, stride, \
                       nullptr, ix->bucket_max, nullptr, nullptr, 0, nullptr, nullptr, ix->rscale, ix->qmeta)
#define CODD_LAUNCH_SAMPLE8P(NBQ)                                                                                            \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_SAMPLE, NBQ, 1, 2>), g, b, lds, st, shadow, qfrag, n, nsteps, ts, stride, \
                       nullptr, ix->bucket_max, nullptr, nullptr, 0, nullptr, nullptr, ix->rscale, ix->qmeta)
        if (tile_v2) {
            // (the sample pass keeps the generic program: its tile-structured instantiations — the static six-step one too, tried in round 3 —
            //  spill inside the loop: the fold's registers on top of two corpus ring slots in flight)
#define CODD_LAUNCH_TILE8_SAMPLE(NQB, RES)                                                                                                    \
    hipLaunchKernelGGL((i8_tile_kernel<MODE_SAMPLE, 0, NQB, RES>), g, b, i8_lds_bytes(MODE_SAMPLE), st, shadow, qfrag, n, nsteps, ts, stride, \
                       nullptr, ix->bucket_max, nullptr, nullptr, 0, nullptr, ix->rscale, ix->qmeta)
            const bool res = i8_tile_resident(nsteps, nbq) && ix->resident_q;  // the query block fits the four LDS slices: loaded once per workgroup
            if (nbq == 8) {
                if (res) CODD_LAUNCH_TILE8_SAMPLE(16, true);
                else CODD_LAUNCH_TILE8_SAMPLE(16, false);
            } else {  // 65..128 queries: half the query blocks
                if (res) CODD_LAUNCH_TILE8_SAMPLE(8, true);
                else CODD_LAUNCH_TILE8_SAMPLE(8, false);
            }
#undef CODD_LAUNCH_TILE8_SAMPLE
        } else if (use8 && partial6) {  // 768 int8 elements: two of the six query slices stay in LDS
            CODD_LAUNCH_SAMPLE8P(4);
        } else if (use8 && resident) {  // the whole int8 query block fits the LDS slices: loaded once per workgroup
            switch (nbq) {
                case 1: CODD_LAUNCH_SAMPLE8R(1); break;
                case 2: CODD_LAUNCH_SAMPLE8R(2); break;
                case 4: CODD_LAUNCH_SAMPLE8R(4); break;
                default: CODD_LAUNCH_SAMPLE8R(8); break;
            }
        } else if (use8) {
            switch (nbq) {
                case 1: CODD_LAUNCH_SAMPLE8(1); break;
                case 2: CODD_LAUNCH_SAMPLE8(2); break;
                case 4: CODD_LAUNCH_SAMPLE8(4); break;
                default: CODD_LAUNCH_SAMPLE8(8); break;
            }
        } else
        switch (nbq) {
            case 1: CODD_LAUNCH_SAMPLE(1); break;
            case 2: CODD_LAUNCH_SAMPLE(2); break;
            case 4: CODD_LAUNCH_SAMPLE(4); break;
            default: CODD_LAUNCH_SAMPLE(8); break;
        }
#undef CODD_LAUNCH_SAMPLE
#undef CODD_LAUNCH_SAMPLE8
#undef CODD_LAUNCH_SAMPLE8R
#undef CODD_LAUNCH_SAMPLE8P
    }
    HIP_TRY(hipGetLastError());
    {
        // thresholds anchored on the exact scores of the k best sampled rows (anchor_thr_kernel)
        const int nch_t = ix->dpad / elems_per_chunk(ix->dtype);
        const int niter_t = (nch_t + kWave - 1) / kWave;
#define CODD_ANCHOR(DT, NI, SL)                                                                                                       \
    hipLaunchKernelGGL((anchor_thr_kernel<DT, NI, SL>), dim3(kTileQ), dim3(kAnchorWaves * kWave), 0, st, ix->bucket_max, ts, nq, k, ix->rows, ix->dpad, qn, \
                       eps, slack_q, ix->thr, use8 ? ix->thr + kTileQ : nullptr)
#define CODD_ANCHOR_NI(DT, SL)                                \
    switch (niter_t) {                                         \
        case 1: CODD_ANCHOR(DT, 1, SL); break;                 \
        case 2: CODD_ANCHOR(DT, 2, SL); break;                 \
        case 3: CODD_ANCHOR(DT, 3, SL); break;                 \
        default: CODD_ANCHOR(DT, 4, SL); break;                \
    }
        if (ix->dtype == DT_F32) { if (slots == 1) { CODD_ANCHOR_NI(DT_F32, 1) } else { CODD_ANCHOR_NI(DT_F32, 2) } }
        else if (ix->dtype == DT_BF16) { if (slots == 1) { CODD_ANCHOR_NI(DT_BF16, 1) } else { CODD_ANCHOR_NI(DT_BF16, 2) } }
        else { if (slots == 1) { CODD_ANCHOR_NI(DT_F16, 1) } else { CODD_ANCHOR_NI(DT_F16, 2) } }
#undef CODD_ANCHOR_NI
#undef CODD_ANCHOR
    }
    HIP_TRY(hipGetLastError());
    {
        EvScope ev(ix, EV_FILTER, st);
        const dim3 g((unsigned)(ntiles < ix->num_cus ? ntiles : ix->num_cus)), b(kFilterThreads);
        const size_t lds = filter_lds_bytes(MODE_FILTER);
#define CODD_LAUNCH_FILTER(NBQ)                                                                                             \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_FILTER, NBQ>), g, b, lds, st, shadow, qfrag, n, nsteps, ntiles,     \
                       (int64_t)1, ix->thr, nullptr, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, nullptr)
#define CODD_LAUNCH_FILTER8(NBQ)                                                                                            \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_FILTER, NBQ, 1>), g, b, lds, st, shadow, qfrag, n, nsteps, ntiles,  \
                       (int64_t)1, ix->thr, nullptr, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, nullptr, ix->rscale, ix->qmeta)
#define CODD_LAUNCH_FILTER8R(NBQ)                                                                                           \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_FILTER, NBQ, 1, 1>), g, b, lds, st, shadow, qfrag, n, nsteps, ntiles, \
                       (int64_t)1, ix->thr, nullptr, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, nullptr, ix->rscale, ix->qmeta)
#define CODD_LAUNCH_FILTER8P(NBQ)                                                                                           \
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_FILTER, NBQ, 1, 2>), g, b, lds, st, shadow, qfrag, n, nsteps, ntiles, \
                       (int64_t)1, ix->thr, nullptr, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, nullptr, ix->rscale, ix->qmeta)
        if (tile_f16) {
            // (thr doubles as the 256 readable bytes the kernel's per-tile metadata request needs; fp16 operands carry no scales)
            // rows of 768 elements are 12 K-steps of 64, rows of 384 are 6: the static forms of the tile program ("i8_pair" = 2); 18, 24, ...: run-time cursors
#define CODD_LAUNCH_TILE16(S3)                                                                                                                            \
    hipLaunchKernelGGL((i8_tile_kernel<MODE_FILTER, S3, 16, false, true>), g, b, i8_lds_bytes(MODE_FILTER), st, shadow, qfrag, n, nsteps, ntiles, (int64_t)1, \
                       ix->thr, nullptr, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, nullptr, nullptr, reinterpret_cast<const float2*>(ix->thr), 0.0f)
            if (nsteps == 12 && ix->i8_pair == 2) CODD_LAUNCH_TILE16(4);
            else if (nsteps == 6 && ix->i8_pair == 2) CODD_LAUNCH_TILE16(3);
            else CODD_LAUNCH_TILE16(2);
#undef CODD_LAUNCH_TILE16
        } else if (tile_v2) {
#ifdef CODD_I8_EXP_STAMPS  // (diagnostic build: the filter pass writes its per-wave phase stamps over the sample's bucket keys, which anchor_thr has consumed)
#define CODD_STAMP_BUF ix->bucket_max
#else
#define CODD_STAMP_BUF nullptr
#endif
#define CODD_LAUNCH_TILE8(S3, NQB, RES)                                                                                                              \
    hipLaunchKernelGGL((i8_tile_kernel<MODE_FILTER, S3, NQB, RES>), g, b, i8_lds_bytes(MODE_FILTER), st, shadow, qfrag, n, nsteps, ntiles, (int64_t)1, \
                       (ix->per_block & 1) ? ix->thr + kTileQ : ix->thr, CODD_STAMP_BUF, ix->hits, ix->ctl->hit_cnt, ix->hit_cap_q, ix->ctl->flags, ix->rscale, ix->qmeta, \
                       ix->bmeta, (ix->per_block & 1) ? 1.0f : 0.0f)
            const bool res = i8_tile_resident(nsteps, nbq) && ix->resident_q;
            // tile structure: rows of 6, 12, ... K-steps (768 elements: the headline shape) run the staged program with one barrier
            // per TWO K-steps; other multiples of 3 one per K-step; the rest the generic interval loop
            // ("i8_pair" = 2, the default: rows of exactly 6 K-steps take that program with every cursor a compile-time constant)
            const bool pair = nsteps % 6 == 0 && ix->i8_pair;
            const bool static6 = nsteps == 6 && ix->i8_pair == 2;
            if (nbq == 8) {
                if (res) { if (nsteps % 3 == 0) CODD_LAUNCH_TILE8(1, 16, true); else CODD_LAUNCH_TILE8(0, 16, true); }
                else if (static6) CODD_LAUNCH_TILE8(3, 16, false);
                else if (pair) CODD_LAUNCH_TILE8(2, 16, false);
                else { if (nsteps % 3 == 0) CODD_LAUNCH_TILE8(1, 16, false); else CODD_LAUNCH_TILE8(0, 16, false); }
            } else {
                if (res) { if (nsteps % 3 == 0) CODD_LAUNCH_TILE8(1, 8, true); else CODD_LAUNCH_TILE8(0, 8, true); }
                else if (static6) CODD_LAUNCH_TILE8(3, 8, false);
                else if (pair) CODD_LAUNCH_TILE8(2, 8, false);
                else { if (nsteps % 3 == 0) CODD_LAUNCH_TILE8(1, 8, false); else CODD_LAUNCH_TILE8(0, 8, false); }
            }
#undef CODD_LAUNCH_TILE8
        } else if (use8 && partial6) {
            CODD_LAUNCH_FILTER8P(4);
        } else if (use8 && resident) {
            switch (nbq) {
                case 1: CODD_LAUNCH_FILTER8R(1); break;
                case 2: CODD_LAUNCH_FILTER8R(2); break;
                case 4: CODD_LAUNCH_FILTER8R(4); break;
                default: CODD_LAUNCH_FILTER8R(8); break;
            }
        } else if (use8) {
            switch (nbq) {
                case 1: CODD_LAUNCH_FILTER8(1); break;
                case 2: CODD_LAUNCH_FILTER8(2); break;
                case 4: CODD_LAUNCH_FILTER8(4); break;
                default: CODD_LAUNCH_FILTER8(8); break;
            }
        } else
        switch (nbq) {
            case 1: CODD_LAUNCH_FILTER(1); break;
            case 2: CODD_LAUNCH_FILTER(2); break;
            case 4: CODD_LAUNCH_FILTER(4); break;
            default: CODD_LAUNCH_FILTER(8); break;
        }
#undef CODD_LAUNCH_FILTER
#undef CODD_LAUNCH_FILTER8
#undef CODD_LAUNCH_FILTER8R
#undef CODD_LAUNCH_FILTER8P
    }
    HIP_TRY(hipGetLastError());
    const int nchunks = ix->dpad / elems_per_chunk(ix->dtype);
    const int niter = (nchunks + kWave - 1) / kWave;
    // the int8 filter leaves thousands of survivors per query and is only used for a handful of queries: share each
    // query's re-scoring out between several workgroups, then merge their lists
    const int nparts = use8 ? (nq <= 8 ? 16 : (nq <= 32 ? 8 : (nq <= 64 ? 4 : 1))) : 1;
    if (nparts > 1 && (rc = ensure_buf(&ix->partial, &ix->partial_cap, (int64_t)nq * nparts * k)) != 0) return rc;
    // one list slot per lane and one workgroup per query (the large batches): finalize and the exact-scan fallback share ONE
    // launch — the scan workgroups derive the queue from the hit counters and leave at once when it is empty
    if (slots == 1 && nparts == 1 && ix->fuse_fallback && !(ix->dtype != DT_F32 && niter == 4)) {  // (2-byte rows above 1536 elements: the fused kernel spills)
        int nit;
        int64_t blocks;
        if ((rc = scan_geometry(ix, n, &nit, &blocks)) != 0) return rc;
        if (blocks > ix->num_cus) blocks = ix->num_cus;  // bounds the queue's partial buffer
        const int64_t stride_q = blocks * k;
        if ((rc = ensure_buf(&ix->fb_partial, &ix->fb_partial_cap, (int64_t)kTileQ * stride_q)) != 0) return rc;
        EvScope ev(ix, EV_FINALIZE, st);
        const float2* bm = slack_q && (ix->per_block & 2) ? ix->bmeta : nullptr;
        FilterCtl* c = ix->ctl;
        const dim3 grid((unsigned)(nq + blocks), 1);
        const size_t lds = (size_t)(kFinThreads / kWave) * 4 * kWave * sizeof(u64);   // the scan role's lists: 8 waves x 4 queries x 64 keys
#define CODD_FIN_FB(DT, NI)                                                                                                                                    \
    hipLaunchKernelGGL((finalize_fb_kernel<DT, NI>), grid, dim3(kFinThreads), lds, st, ix->rows, ix->dpad, qn, ix->hits, c->hit_cnt, ix->hit_cap_q, c->flags, k, \
                       2.0f * eps, row_base, keys_out, ix->dstats, slack_q, (u64*)nullptr, dist_out, rows_out, bm, nq, n, ix->fb_partial, stride_q, &c->fb_done)
#define CODD_FIN_FB_NI(DT)                          \
    switch (niter) {                                \
        case 1: CODD_FIN_FB(DT, 1); break;          \
        case 2: CODD_FIN_FB(DT, 2); break;          \
        case 3: CODD_FIN_FB(DT, 3); break;          \
        case 4: CODD_FIN_FB(DT, 4); break;          \
        default: return fail(CODD_KNN_ENOTSUP, "row too wide for the finalize kernel%s"); \
    }
        if (ix->dtype == DT_F32) { CODD_FIN_FB_NI(DT_F32) }
        else if (ix->dtype == DT_BF16) { CODD_FIN_FB_NI(DT_BF16) }
        else { CODD_FIN_FB_NI(DT_F16) }
#undef CODD_FIN_FB_NI
#undef CODD_FIN_FB
        HIP_TRY(hipGetLastError());
        return CODD_KNN_OK;
    }
    {
        EvScope ev(ix, EV_FINALIZE, st);
        switch (ix->dtype) {
            case DT_F32: rc = launch_finalize<DT_F32>(niter, slots, nq, st, ix, qn, k, 2.0f * eps, row_base, keys_out, slack_q, nparts, ix->partial, dist_out, rows_out); break;
            case DT_BF16: rc = launch_finalize<DT_BF16>(niter, slots, nq, st, ix, qn, k, 2.0f * eps, row_base, keys_out, slack_q, nparts, ix->partial, dist_out, rows_out); break;
            default: rc = launch_finalize<DT_F16>(niter, slots, nq, st, ix, qn, k, 2.0f * eps, row_base, keys_out, slack_q, nparts, ix->partial, dist_out, rows_out); break;
        }
    }
    if (rc != 0) return rc;
    if (nparts > 1 && (rc = launch_merge(ix->partial, nq, (int64_t)nparts * k, (int64_t)nparts * k, k, keys_out, dist_out, rows_out, st)) != 0) return rc;

    // exact-scan fallback for the queries finalize queued (normally none), entirely on the device and in ONE launch: the
    // scan walks the queue (an empty queue costs one empty launch), its last block merges the per-block partials and
    // writes each answer into its query's slot.  No host round trip: the whole search stays asynchronous on `st`.
    {
        int nit;
        int64_t blocks;
        if ((rc = scan_geometry(ix, n, &nit, &blocks)) != 0) return rc;
        if (blocks > ix->num_cus) blocks = ix->num_cus;  // bounds the queue's partial buffer
        const int64_t stride_q = blocks * k;
        if ((rc = ensure_buf(&ix->fb_partial, &ix->fb_partial_cap, (int64_t)kTileQ * stride_q)) != 0) return rc;
        ScanArgs a{ix->rows, n, ix->dpad, qn, 0, k, row_base, ix->fb_partial, stride_q, ix->ctl->fb_list, &ix->ctl->fb_count};
        a.merge_done = &ix->ctl->fb_done;
        a.merged_keys = keys_out;
        a.merged_dist = dist_out;
        a.merged_rows = rows_out;
        a.count_total = &ix->dstats[2];
        {
            EvScope ev(ix, EV_SCAN, st);
            switch (ix->dtype) {
                case DT_F32: rc = launch_scan_nb<DT_F32>(8, nit, slots, dim3((unsigned)blocks), st, a); break;
                case DT_BF16: rc = launch_scan_nb<DT_BF16>(8, nit, slots, dim3((unsigned)blocks), st, a); break;
                default: rc = launch_scan_nb<DT_F16>(8, nit, slots, dim3((unsigned)blocks), st, a); break;
            }
        }
        if (rc != 0) return rc;
        HIP_TRY(hipGetLastError());
    }
    return CODD_KNN_OK;
}

// ---- one launch for a single query (small_batch_kernel) + the (normally empty) list-driven fallback scan ----
template <int DT, int NS>
void launch_small_batch(int64_t nunits, hipStream_t st, const codd_knn_index* ix, const float* dev_queries, int k, uint32_t row_base, u64* cand,
                        u64* out_keys, float* out_dist, int64_t* out_rows) {
    constexpr int E = DT == DT_F32 ? 4 : 8;
    constexpr int NITER = (NS * 128 / E + kWave - 1) / kWave;  // chunks of the PADDED row per lane (dpad <= NS * 128)
    constexpr int kWavesPerWg = kSbThreads / kWave;
    // one round of workgroups: as many as are resident at once (the last one to arrive answers the query)
    static std::atomic<int> per_cu{0};
    int nb = per_cu.load(std::memory_order_relaxed);
    if (nb == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)&small_batch_kernel<DT, NITER, NS>, kSbThreads, 0) != hipSuccess || nb < 1) nb = 1;
        if (nb > 2) nb = 2;
        (void)hipGetLastError();
        per_cu.store(nb, std::memory_order_relaxed);
    }
    int64_t G = (int64_t)nb * ix->num_cus;
    if (G * kWavesPerWg > nunits) G = (nunits + kWavesPerWg - 1) / kWavesPerWg;
    const dim3 grid((unsigned)G);
    u64* dropmax = cand + G * kWavesPerWg * kSbKeep;
    FilterCtl* c = ix->ctl;
    hipLaunchKernelGGL((small_batch_kernel<DT, NITER, NS>), grid, dim3(kSbThreads), 0, st, ix->shadow8, ix->bmeta, ix->rows, ix->count, ix->dim, ix->dpad,
                       dev_queries, k, row_base, ix->eps_r_bits, ix->qn, cand, dropmax, &c->sb_ticket, &c->fb_count, c->fb_list, out_keys, out_dist, out_rows,
                       ix->dstats);
}
template <int DT>
bool launch_small_batch_ns(int ns, int64_t nunits, hipStream_t st, const codd_knn_index* ix, const float* dev_queries, int k, uint32_t row_base, u64* cand,
                           u64* out_keys, float* out_dist, int64_t* out_rows) {
    switch (ns) {
        case 3: launch_small_batch<DT, 3>(nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); return true;
        case 4: launch_small_batch<DT, 4>(nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); return true;
        case 6: launch_small_batch<DT, 6>(nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); return true;
        case 8: launch_small_batch<DT, 8>(nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); return true;
        default: return false;
    }
}
bool small_batch_applies(const codd_knn_index* ix, int B, int k) {
    const int ns = dpad8_of(ix) / 128;
    return ix->small_batch_max > 0 && B == 1 && k <= kWave && (ns == 3 || ns == 4 || ns == 6 || ns == 8) && ix->count >= 4096;
}
int small_batch_search(codd_knn_index* ix, const float* dev_queries, int B, int k, uint32_t row_base, u64* out_keys, float* out_dist, int64_t* out_rows,
                       hipStream_t st) {
    (void)B;
    const int ns = dpad8_of(ix) / 128;
    const int64_t n = ix->count;
    constexpr int kWavesPerWg = kSbThreads / kWave;
    const int64_t nunits = (n + 15) / 16;
    int rc;
    if ((rc = ensure_buf(&ix->sb_cand, &ix->sb_cand_cap, (int64_t)(2 * ix->num_cus) * kWavesPerWg * (kSbKeep + 1))) != 0) return rc;
    u64* cand = ix->sb_cand;
    ix->stat_small_batch++;
    {
        EvScope ev(ix, EV_FILTER, st);
        bool ok;
        switch (ix->dtype) {
            case DT_F32: ok = launch_small_batch_ns<DT_F32>(ns, nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); break;
            case DT_BF16: ok = launch_small_batch_ns<DT_BF16>(ns, nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); break;
            default: ok = launch_small_batch_ns<DT_F16>(ns, nunits, st, ix, dev_queries, k, row_base, cand, out_keys, out_dist, out_rows); break;
        }
        if (!ok) return fail(CODD_KNN_EINVAL, "small batch: unsupported row width%s");
    }
    HIP_TRY(hipGetLastError());
    // a query the margin test could not clear (normally none): the list-driven exact scan, its last block merges
    {
        int nit;
        int64_t blocks;
        if ((rc = scan_geometry(ix, n, &nit, &blocks)) != 0) return rc;
        if (blocks > ix->num_cus) blocks = ix->num_cus;
        const int64_t stride_q = blocks * k;
        if ((rc = ensure_buf(&ix->fb_partial, &ix->fb_partial_cap, (int64_t)kTileQ * stride_q)) != 0) return rc;
        ScanArgs a{ix->rows, n, ix->dpad, ix->qn, 0, k, row_base, ix->fb_partial, stride_q, ix->ctl->fb_list, &ix->ctl->fb_count};
        a.merge_done = &ix->ctl->fb_done;
        a.merged_keys = out_keys;
        a.merged_dist = out_dist;
        a.merged_rows = out_rows;
        a.count_total = &ix->dstats[2];
        EvScope ev(ix, EV_SCAN, st);
        switch (ix->dtype) {
            case DT_F32: rc = launch_scan_nb<DT_F32>(8, nit, 1, dim3((unsigned)blocks), st, a); break;
            case DT_BF16: rc = launch_scan_nb<DT_BF16>(8, nit, 1, dim3((unsigned)blocks), st, a); break;
            default: rc = launch_scan_nb<DT_F16>(8, nit, 1, dim3((unsigned)blocks), st, a); break;
        }
        if (rc != 0) return rc;
        HIP_TRY(hipGetLastError());
    }
    return CODD_KNN_OK;
}

bool filter_applies(const codd_knn_index* ix, int B, int k) {
    // the thresholds come from the k-th largest of the sampled tile maxima: need comfortably more tiles than k
    const int64_t ntiles = (ix->count + kTileRows - 1) / kTileRows;
    const int64_t ts = sample_tile_count(ix, ntiles, k);
    if (!(ix->filter_enabled && ix->all_normalized) || ts < 2 * (int64_t)k) return false;
    // measured on MI355X (scripts/crossover.py, d = 768): with more than 8 queries the filter wins at every size
    // it is sound for; up to 8 queries the exact scan's single launch wins until rows * B reaches ~100k
    if (B >= ix->filter_min_batch) return ix->count >= ix->filter_min_rows;
    return ix->count * (int64_t)B >= ix->filter_min_rows_small;
}

// the index looks at its own device counters now and then (one look in flight; after its first 32 searches only every 8th:
// the copy is a launch of its own on the searching stream)
int after_filter_search(codd_knn_index* ix, int B, bool use8, hipStream_t st) {
    (use8 ? ix->watch_q8 : ix->watch_q16) += B;
    if (ix->shadow8_enabled && !ix->watch_pending && ix->dstats && (ix->stat_searches <= 32 || (ix->stat_searches & 7) == 0)) {
        if (!ix->watch_host) {
            HIP_TRY(hipHostMalloc((void**)&ix->watch_host, 4 * sizeof(unsigned long long), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ix->watch_copied, hipEventDisableTiming));
        }
        HIP_TRY(hipMemcpyAsync(ix->watch_host, ix->dstats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(ix->watch_copied, st));
        ix->watch_pending = true;
        ix->watch_q8_sent = ix->watch_q8;
        ix->watch_q16_sent = ix->watch_q16;
    }
    return CODD_KNN_OK;
}

// the whole shard-local search: normalise queries, then filter passes or exact scans, keys out.
int search_impl(codd_knn_index* ix, const float* dev_queries, int B, int k, uint32_t row_base, u64* out_keys,
                float* out_dist, int64_t* out_rows, hipStream_t st) {
    if (!ix) return fail(CODD_KNN_EINVAL, "null index%s");
    if (!dev_queries) return fail(CODD_KNN_EINVAL, "null queries%s");
    if (B < 1 || B > CODD_KNN_MAX_BATCH) return fail(CODD_KNN_EINVAL, "B out of range [1,1024]%s");
    if (k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "k out of range [1,128]%s");
    DeviceGuard guard(ix->device);
    ix->stat_searches++;
    const int64_t n = ix->count;
    int rc;
    if ((rc = ensure_buf(&ix->qn, &ix->qn_cap, (int64_t)B * ix->dpad)) != 0) return rc;
    if ((rc = ensure_buf(&ix->keys_tmp, &ix->keys_tmp_cap, (int64_t)B * k)) != 0) return rc;
    if ((rc = wait_rows(ix, st)) != 0) return rc;
    bool use_filter = n > 0 && filter_applies(ix, B, k);
    if (ix->eps_r_copied) {
        if (hipEventQuery(ix->eps_r_copied) == hipSuccess) {
            ix->eps_r_known = ix->eps_r_host[0];
            ix->wide_blocks_known = (int64_t)reinterpret_cast<const unsigned*>(ix->eps_r_host)[1];
        }
        else (void)hipGetLastError();  // "not ready" must not surface in a later error check
    }
    if (ix->watch_pending && ix->watch_copied) {
        if (hipEventQuery(ix->watch_copied) == hipSuccess) {
            ix->watch_pending = false;
            const unsigned long long surv = ix->watch_host[1], fb = ix->watch_host[2];
            if (ix->watch_q8_sent > 0 && ix->watch_q16_sent == 0) {  // only int8 passes in the window: the deltas are theirs
                const double per_q = (double)(surv - ix->watch_surv) / (double)ix->watch_q8_sent;
                if ((per_q > (double)ix->shadow8_max_surv || (fb - ix->watch_fb) * 20 > (unsigned long long)ix->watch_q8_sent) &&
                    ix->cooldown_useless_epoch != ix->epoch) {
                    ix->cooldown_left = ix->shadow8_cooldown;
                    ix->cooldown_surv8 = per_q;  // what the 2-byte filter has to beat
                    ix->stat_cooldowns++;
                }
            } else if (ix->watch_q16_sent > 0 && ix->watch_q8_sent == 0) {
                // only bf16 passes (a cooldown): when those leave the re-scoring just as crowded — rows closer to each other than
                // EITHER slack — the slower bf16 kernel buys nothing: stay on int8 until rows change
                // (relative, not absolute: the fp16 filter's passes cost about twice an int8 pass, so they pay as soon as they
                // leave well under half of the exact re-scoring — 7,000 survivors per query are a bargain against 62,000)
                const double per_q = (double)(surv - ix->watch_surv) / (double)ix->watch_q16_sent;
                if (per_q > (double)ix->shadow8_max_surv && per_q > 0.5 * ix->cooldown_surv8) {
                    ix->cooldown_useless_epoch = ix->epoch;
                    ix->cooldown_left = 0;
                }
            }
            ix->watch_surv = surv;
            ix->watch_fb = fb;
            ix->watch_q8 -= ix->watch_q8_sent;
            ix->watch_q16 -= ix->watch_q16_sent;
        } else {
            (void)hipGetLastError();
        }
    }
    const bool cooling = ix->cooldown_left > 0;
    if (cooling && use_filter) ix->cooldown_left--;
    // the int8 filter: every pass of <= 256 queries prepares its own block (a batch above 256 queries is several passes)
    const bool can8 = use_filter && ix->shadow8_enabled && (B <= ix->shadow8_max_batch || (B > kTileQ && ix->shadow8_max_batch >= kTileQ)) && CODD_MFMA16;
    // A corpus whose WORST block quantises badly keeps the int8 filter for the batches i8_tile_kernel takes as long as such blocks are
    // rare (under 1 %): that kernel and finalize evaluate the bound per 32-row block, so only the bad blocks' rows come through
    // as extra candidates.  The first-generation kernels (other batch sizes, rows under 384 elements) use the device-wide bound.
    const int nsteps8 = dpad8_of(ix) / 128;
    const bool per_block = ix->i8v2 != 0 && nsteps8 >= 3 && B > 64 && (B <= 128 ? ix->i8v2_half != 0 : B <= kTileQ) && (ix->i8v2 == 2 || nsteps8 > 4 || !ix->resident_q);
    const bool eps_ok = ix->eps_r_known <= ix->shadow8_max_eps || (per_block && ix->wide_blocks_known * 100 <= (ix->count + 31) / 32);
    bool use8 = can8 && !cooling && eps_ok;
    if (use_filter && !use8) {
        // the bf16 filter: its shadow is allocated by the first search that needs it.  When HBM cannot hold it the search is
        // still answered exactly — through the int8 filter where the index may use it (a cooldown or a wide eps_r only make
        // that one slower), else by the exact scan — and the failed allocation is not retried until rows change.
        if ((rc = ensure_filter_workspace(ix)) != 0) return rc;
        rc = ensure_shadow(ix, st);
        if (rc == CODD_KNN_ENOMEM) {
            if (can8) use8 = true;
            else use_filter = false;
        } else if (rc != 0) {
            return rc;
        }
    }
    const bool fused_prep = use_filter && B <= kTileQ;
    const int dpad8 = dpad8_of(ix);
    auto prep8 = [&](int q0, int nq) -> int {
        hipLaunchKernelGGL(prep_queries8_kernel, dim3(kTileQ / 4), dim3(256), 0, st, dev_queries + (int64_t)q0 * ix->dim, nq, ix->dim, ix->dpad, dpad8,
                           ix->qn + (int64_t)q0 * ix->dpad, reinterpret_cast<uint32_t*>(ix->qfrag8), ix->qmeta, ix->eps_r_bits,
                           reinterpret_cast<unsigned*>(ix->ctl), (int)(sizeof(FilterCtl) / 4), ix->exp_slack_scale);
        HIP_TRY(hipGetLastError());
        return CODD_KNN_OK;
    };
    if (use8) {
        if ((rc = ensure_filter_workspace(ix)) != 0) return rc;
        if ((rc = ensure_shadow8(ix, st)) != 0) return rc;
        // a handful of queries: ONE launch streams the int8 shadow, keeps the best approximate scores per wave and re-scores the
        // survivors exactly in its last workgroup (small_batch_kernel) instead of the six-launch filter chain
        if (small_batch_applies(ix, B, k)) {
            if ((rc = ensure_buf(&ix->qn, &ix->qn_cap, (int64_t)B * ix->dpad)) != 0) return rc;
            if ((rc = small_batch_search(ix, dev_queries, B, k, row_base, out_keys, out_dist, out_rows, st)) != 0) return rc;
            return after_filter_search(ix, B, true, st);
        }
        if ((rc = ensure_buf(&ix->qfrag8, &ix->qfrag8_cap, (int64_t)kTileQ * (dpad8 / 16))) != 0) return rc;
        if (!ix->qmeta) HIP_TRY(hipMalloc((void**)&ix->qmeta, 1024 * sizeof(float)));
    } else if (fused_prep) {
        hipLaunchKernelGGL(prep_queries_kernel, dim3(kTileQ / 4), dim3(256), 0, st, dev_queries, B, ix->dim, ix->dpad, ix->qn,
                           reinterpret_cast<uint2*>(ix->qfrag), reinterpret_cast<unsigned*>(ix->ctl), (int)(sizeof(FilterCtl) / 4));
        HIP_TRY(hipGetLastError());
    } else if ((rc = launch_normalize(DT_F32, dev_queries, B, ix->dim, ix->dpad, 1, nullptr, 0, ix->qn, nullptr, st)) != 0) {
        return rc;
    }

    // the filter passes write what the caller asked for — packed keys (the shard-local half of a sharded search) and / or
    // (distance, row) — straight from finalize: no unpack launch behind them
    if (n == 0) {
        // nothing stored: all-empty result (chromadb returns {"ids": [[]], ...})
        u64* keys_dst = out_keys ? out_keys : ix->keys_tmp;
        HIP_TRY(hipMemsetAsync(keys_dst, 0, (size_t)B * k * sizeof(u64), st));
        if (!out_dist && !out_rows) return CODD_KNN_OK;
        return launch_merge(keys_dst, B, k, k, k, nullptr, out_dist, out_rows, st);
    }
    if (!use_filter)  // small batches: the per-block partials merge straight into the caller's buffers
        return exact_scan(ix, ix->qn, B, k, row_base, out_keys, out_dist, out_rows, st);
    if ((rc = ensure_filter_workspace(ix)) != 0) return rc;
    for (int q0 = 0; q0 < B; q0 += kTileQ) {
        const int nq = B - q0 < kTileQ ? B - q0 : kTileQ;
        if (use8 && (rc = prep8(q0, nq)) != 0) return rc;
        if ((rc = filter_pass(ix, ix->qn + (int64_t)q0 * ix->dpad, nq, k, row_base, out_keys ? out_keys + (int64_t)q0 * k : nullptr,
                              out_dist ? out_dist + (int64_t)q0 * k : nullptr, out_rows ? out_rows + (int64_t)q0 * k : nullptr, st,
                              fused_prep || use8, use8)) != 0)
            return rc;
    }
    return after_filter_search(ix, B, use8, st);
}

}  // namespace

extern "C" {

const char* codd_knn_version(void) {
    return "codd_knn 0.6.0 gfx950"
#if CODD_SHADOW_F16
           " shadow=f16"
#else
           " shadow=bf16"
#endif
#if CODD_MFMA16
           " mfma=16x16x32 int8=16x16x64";
#else
           " mfma=32x32x16";
#endif
}
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
    void* bufs[] = {ix->rows, ix->shadow, ix->dstats, ix->rows_ivf, ix->ivf_ids, ix->ivf_offsets, ix->shadow8, ix->rscale, ix->bmeta, ix->eps_r_bits};
    if (ix->shadow8_ready) (void)hipEventDestroy(ix->shadow8_ready);
    if (ix->shadow_ready) (void)hipEventDestroy(ix->shadow_ready);
    if (ix->rows_ready) (void)hipEventDestroy(ix->rows_ready);
    if (ix->reader_done) (void)hipEventDestroy(ix->reader_done);
    if (ix->stage_vec) (void)hipFree(ix->stage_vec);
    if (ix->stage_slot) (void)hipFree(ix->stage_slot);
    if (ix->watch_copied) (void)hipEventDestroy(ix->watch_copied);
    if (ix->watch_host) (void)hipHostFree(ix->watch_host);
    if (ix->eps_r_copied) (void)hipEventDestroy(ix->eps_r_copied);
    if (ix->eps_r_host) (void)hipHostFree(ix->eps_r_host);
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (WorkSlot& w : ix->slots) {
        void* wb[] = {w.bufs.qn, w.bufs.partial, w.bufs.keys_tmp, w.bufs.qfrag, w.bufs.thr, w.bufs.bucket_max, w.bufs.hits, w.bufs.ctl,
                      w.bufs.fb_partial, w.bufs.probe_keys, w.bufs.ivf_partial, w.bufs.ivf_group, w.bufs.qfrag8, w.bufs.qmeta, w.bufs.sb_cand};
        for (void* b : wb)
            if (b) (void)hipFree(b);
        if (w.handover) (void)hipEventDestroy(w.handover);
    }
    if (ix->coarse) (void)codd_knn_destroy(ix->coarse);
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
    return grow_rows(ix, rows, /*exact=*/true);  // the caller states the final size (288 GB of HBM is the only limit)
}

int codd_knn_upsert_host(codd_knn_index* ix, const int64_t* host_slots, const float* host_vecs, int64_t n, int normalize) {
    if (!ix || (n > 0 && (!host_slots || !host_vecs)) || n < 0) return fail(CODD_KNN_EINVAL, "bad upsert arguments%s");
    if (n == 0) return CODD_KNN_OK;
    int64_t max_slot = -1, min_slot = INT64_MAX;
    for (int64_t i = 0; i < n; ++i) {
        if (host_slots[i] < 0 || host_slots[i] >= 0xfffffffell) return fail(CODD_KNN_EINVAL, "row slot out of range%s");
        if (host_slots[i] > max_slot) max_slot = host_slots[i];
        if (host_slots[i] < min_slot) min_slot = host_slots[i];
    }
    DeviceGuard guard(ix->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = grow_rows(ix, max_slot + 1, /*exact=*/false);
    if (rc != 0) return rc;
    // stage in bounded pieces (<= 64 MiB of vectors per piece) through buffers the index keeps between calls (the indexer job
    // upserts one small batch at a time: a hipMalloc / hipFree pair per call used to dominate them)
    const int64_t piece = (int64_t)(64ll << 20) / ((int64_t)ix->dim * 4) + 1;
    const int64_t pn = n < piece ? n : piece;
    if (pn * ix->dim > ix->stage_vec_cap) {
        if (ix->stage_vec) (void)hipFree(ix->stage_vec);
        ix->stage_vec = nullptr; ix->stage_vec_cap = 0;
        const int64_t want = pn * ix->dim < 65536 ? 65536 : pn * ix->dim;
        HIP_TRY(hipMalloc((void**)&ix->stage_vec, (size_t)want * sizeof(float)));
        ix->stage_vec_cap = want;
    }
    if (pn > ix->stage_slot_cap) {
        if (ix->stage_slot) (void)hipFree(ix->stage_slot);
        ix->stage_slot = nullptr; ix->stage_slot_cap = 0;
        const int64_t want = pn < 4096 ? 4096 : pn;
        HIP_TRY(hipMalloc((void**)&ix->stage_slot, (size_t)want * sizeof(int64_t)));
        ix->stage_slot_cap = want;
    }
    float* dvec = ix->stage_vec;
    int64_t* dslot = ix->stage_slot;
    rc = CODD_KNN_OK;
    for (int64_t i0 = 0; i0 < n && rc == 0; i0 += pn) {
        const int64_t m = n - i0 < pn ? n - i0 : pn;
        hipError_t e = hipMemcpy(dvec, host_vecs + i0 * ix->dim, (size_t)m * ix->dim * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dslot, host_slots + i0, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = fail(CODD_KNN_EDEVICE, "staging copy failed: %s", hipGetErrorString(e));
            break;
        }
        rc = launch_normalize(ix->dtype, dvec, m, ix->dim, ix->dpad, normalize, dslot, 0, ix->rows, nullptr, nullptr);
        // (one synchronisation per piece: the staging buffers are reused by the next piece, and the call is synchronous by contract)
        if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = fail(CODD_KNN_EDEVICE, "ingest kernel failed%s");
    }
    ix->rows_event_set = false;  // (everything is complete on the device)
    ix->reader_event_set = false;
    if (rc == 0) {
        if (max_slot + 1 > ix->count) ix->count = max_slot + 1;
        rows_written(ix, min_slot, max_slot + 1);
        if (!normalize) ix->all_normalized = false;
    }
    return rc;
}

int codd_knn_upsert_device(codd_knn_index* ix, int64_t first_slot, const float* dev_vecs, int64_t n, int normalize, void* stream) {
    if (!ix || n < 0 || first_slot < 0 || (n > 0 && !dev_vecs)) return fail(CODD_KNN_EINVAL, "bad upsert arguments%s");
    if (n == 0) return CODD_KNN_OK;
    if (first_slot + n >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "row slots must fit 32 bits%s");
    DeviceGuard guard(ix->device);
    if (first_slot + n > ix->capacity) {
        HIP_TRY(hipDeviceSynchronize());
        int rc = grow_rows(ix, first_slot + n, /*exact=*/false);
        if (rc != 0) return rc;
    }
    hipStream_t wst = (hipStream_t)stream;
    // device-side ordering against searches still in flight on OTHER streams (the call is exclusive on the host, but their
    // kernels may still be reading the rows this launch overwrites): the writing stream waits for every searching stream
    for (WorkSlot& w : ix->slots) {
        if (!w.used || w.stream == wst) continue;
        if (!w.handover) HIP_TRY(hipEventCreateWithFlags(&w.handover, hipEventDisableTiming));
        if (hipEventRecord(w.handover, w.stream) == hipSuccess) HIP_TRY(hipStreamWaitEvent(wst, w.handover, 0));
        else (void)hipGetLastError();  // (a stream that no longer exists has nothing in flight)
    }
    // ... and for the previous writer, when that was another stream: rows_ready is ONE event, re-recorded by every write, so
    // chaining the writers makes the last record cover every earlier write (a reader only ever waits for the last one)
    if (ix->rows_event_set && ix->rows_stream != wst) HIP_TRY(hipStreamWaitEvent(wst, ix->rows_ready, 0));
    // ... and for streams that only READ rows outside a search (codd_knn_copy_rows_f32)
    if (ix->reader_event_set && ix->reader_stream != wst) HIP_TRY(hipStreamWaitEvent(wst, ix->reader_done, 0));
    int rc = launch_normalize(ix->dtype, dev_vecs, n, ix->dim, ix->dpad, normalize, nullptr, first_slot, ix->rows, nullptr, wst);
    if (rc != 0) return rc;
    // ... and searches on other streams wait for this write (wait_rows)
    if (!ix->rows_ready) HIP_TRY(hipEventCreateWithFlags(&ix->rows_ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ix->rows_ready, wst));
    ix->rows_stream = wst;
    ix->rows_event_set = true;
    if (first_slot + n > ix->count) ix->count = first_slot + n;
    rows_written(ix, first_slot, first_slot + n);
    if (!normalize) ix->all_normalized = false;
    return CODD_KNN_OK;
}

int codd_knn_load_rows(codd_knn_index* ix, int64_t first_slot, const void* host_rows, int64_t n) {
    if (!ix || first_slot < 0 || n < 0 || (n > 0 && !host_rows)) return fail(CODD_KNN_EINVAL, "bad load_rows arguments%s");
    if (n == 0) return CODD_KNN_OK;
    if (first_slot + n >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "row slots must fit 32 bits%s");
    DeviceGuard guard(ix->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = grow_rows(ix, first_slot + n, /*exact=*/false);
    if (rc != 0) return rc;
    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    HIP_TRY(hipMemcpy((char*)ix->rows + (size_t)first_slot * row_bytes, host_rows, (size_t)n * row_bytes, hipMemcpyHostToDevice));
    // the loaded rows are taken as they are, so their norms are checked: the filters stay on only for unit rows
    // (tolerance = the storage type's rounding of a unit vector)
    unsigned* dev_bits = nullptr;
    HIP_TRY(hipMalloc((void**)&dev_bits, sizeof(unsigned)));
    hipError_t e = hipMemset(dev_bits, 0, sizeof(unsigned));
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    if (e == hipSuccess) {
        switch (ix->dtype) {
            case DT_F32: hipLaunchKernelGGL(row_norm_check_kernel<DT_F32>, grid, block, 0, nullptr, ix->rows, first_slot, n, ix->dpad, dev_bits); break;
            case DT_BF16: hipLaunchKernelGGL(row_norm_check_kernel<DT_BF16>, grid, block, 0, nullptr, ix->rows, first_slot, n, ix->dpad, dev_bits); break;
            default: hipLaunchKernelGGL(row_norm_check_kernel<DT_F16>, grid, block, 0, nullptr, ix->rows, first_slot, n, ix->dpad, dev_bits); break;
        }
        e = hipGetLastError();
    }
    float worst = 0.0f;
    if (e == hipSuccess) e = hipMemcpy(&worst, dev_bits, sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dev_bits);
    if (e != hipSuccess) return fail(CODD_KNN_EDEVICE, "norm check of the loaded rows failed: %s", hipGetErrorString(e));
    const float tol = ix->dtype == DT_F32 ? 1e-4f : (ix->dtype == DT_BF16 ? 4e-3f : 6e-4f);
    if (!(worst <= tol)) ix->all_normalized = false;
    ix->rows_event_set = false;
    if (first_slot + n > ix->count) ix->count = first_slot + n;
    rows_written(ix, first_slot, first_slot + n);
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
    if (!ix) return fail(CODD_KNN_EINVAL, "null index%s");
    WorkScope work(ix, (hipStream_t)stream);
    return search_impl(ix, dev_queries, B, k, 0u, nullptr, dev_dist, dev_rows, (hipStream_t)stream);
}

int codd_knn_search_keys(codd_knn_index* ix, const float* dev_queries, int B, int k, uint32_t row_base, uint64_t* dev_keys, void* stream) {
    if (!dev_keys) return fail(CODD_KNN_EINVAL, "null output%s");
    if (!ix) return fail(CODD_KNN_EINVAL, "null index%s");
    if ((int64_t)row_base + ix->count >= 0xffffffffll) return fail(CODD_KNN_EINVAL, "global row ids must fit 32 bits%s");
    WorkScope work(ix, (hipStream_t)stream);
    return search_impl(ix, dev_queries, B, k, row_base, (u64*)dev_keys, nullptr, nullptr, (hipStream_t)stream);
}

int codd_knn_merge_keys(int device, const uint64_t* dev_keys_in, int B, int m, int k, uint64_t* dev_keys_out, float* dev_dist,
                        int64_t* dev_rows, void* stream) {
    if (!dev_keys_in || B < 1 || m < 1 || k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "bad merge arguments%s");
    DeviceGuard guard(device);
    return launch_merge((const u64*)dev_keys_in, B, m, m, k, (u64*)dev_keys_out, dev_dist, dev_rows, (hipStream_t)stream);
}

int codd_knn_merge_shards(int device, const uint64_t* dev_keys_in, int G, int B, int k_in, int k, uint64_t* dev_keys_out, float* dev_dist,
                          int64_t* dev_rows, void* stream) {
    if (!dev_keys_in || G < 1 || B < 1 || k_in < 1 || k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "bad merge arguments%s");
    DeviceGuard guard(device);
    return launch_merge((const u64*)dev_keys_in, B, (int64_t)G * k_in, k_in, k, (u64*)dev_keys_out, dev_dist, dev_rows, (hipStream_t)stream, nullptr,
                        nullptr, nullptr, k_in, (int64_t)B * k_in);
}

int codd_knn_approx_scores(codd_knn_index* ix, const float* dev_queries, int B, float* dev_scores, void* stream) {
    if (!ix || !dev_queries || !dev_scores || B < 1 || B > kTileQ) return fail(CODD_KNN_EINVAL, "bad debug arguments%s");
    if (ix->count < 1) return fail(CODD_KNN_EINVAL, "empty index%s");
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    WorkScope work(ix, st);
    int rc;
    if ((rc = wait_rows(ix, st)) != 0) return rc;
    if ((rc = ensure_buf(&ix->qn, &ix->qn_cap, (int64_t)B * ix->dpad)) != 0) return rc;
    if ((rc = ensure_filter_workspace(ix)) != 0) return rc;
    if (ix->shadow8_enabled && B <= 32 && CODD_MFMA16) {
        // the int8 filter's scores (what a batch of <= 32 queries is filtered with when "shadow8" is on)
        if ((rc = ensure_shadow8(ix, st)) != 0) return rc;
        const int dpad8 = dpad8_of(ix);
        if ((rc = ensure_buf(&ix->qfrag8, &ix->qfrag8_cap, (int64_t)kTileQ * (dpad8 / 16))) != 0) return rc;
        if (!ix->qmeta) HIP_TRY(hipMalloc((void**)&ix->qmeta, 1024 * sizeof(float)));
        hipLaunchKernelGGL(prep_queries8_kernel, dim3(kTileQ / 4), dim3(256), 0, st, dev_queries, B, ix->dim, ix->dpad, dpad8, ix->qn,
                           reinterpret_cast<uint32_t*>(ix->qfrag8), ix->qmeta, ix->eps_r_bits, reinterpret_cast<unsigned*>(ix->ctl),
                           (int)(sizeof(FilterCtl) / 4), 1.0f);
        const int64_t ntiles8 = (ix->count + kTileRows - 1) / kTileRows;
        const int64_t g8 = ntiles8 < ix->num_cus ? ntiles8 : ix->num_cus;
        hipLaunchKernelGGL((gemm_filter_kernel<MODE_DUMP, 1, 1>), dim3((unsigned)g8), dim3(kFilterThreads), filter_lds_bytes(MODE_DUMP), st,
                           ix->shadow8, ix->qfrag8, ix->count, dpad8 / 128, ntiles8, (int64_t)1, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                           dev_scores, ix->rscale, ix->qmeta);
        HIP_TRY(hipGetLastError());
        return CODD_KNN_OK;
    }
    if ((rc = ensure_shadow(ix, st)) != 0) return rc;
    if ((rc = launch_normalize(DT_F32, dev_queries, B, ix->dim, ix->dpad, 1, nullptr, 0, ix->qn, nullptr, st)) != 0) return rc;
    hipLaunchKernelGGL(qfrag_kernel, dim3((kTileQ * (ix->dpad / 8) + 255) / 256), dim3(256), 0, st, ix->qn, B, ix->dpad, ix->qfrag,
                       (unsigned*)nullptr, 0);
    const int64_t ntiles = (ix->count + kTileRows - 1) / kTileRows;
    const int64_t g = ntiles < ix->num_cus ? ntiles : ix->num_cus;
    hipLaunchKernelGGL((gemm_filter_kernel<MODE_DUMP, 8>), dim3((unsigned)g), dim3(kFilterThreads), filter_lds_bytes(MODE_DUMP), st, ix->shadow,
                       ix->qfrag, ix->count, ix->dpad / 64, ntiles, (int64_t)1, nullptr, nullptr, nullptr, nullptr, 0, nullptr, dev_scores);
    HIP_TRY(hipGetLastError());
    return CODD_KNN_OK;
}

int codd_knn_copy_rows_f32(codd_knn_index* ix, int64_t first, int64_t n, float* dev_out, void* stream) {
    if (!ix || first < 0 || n < 0 || first + n > ix->count || (n > 0 && !dev_out)) return fail(CODD_KNN_EINVAL, "bad copy_rows range%s");
    if (n == 0) return CODD_KNN_OK;
    DeviceGuard guard(ix->device);
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(ix->mu);
    // a read of the stored rows on the caller's stream: behind the last asynchronous write (as a search is), and the next
    // writer on another stream is ordered behind it (reader_done)
    int rc;
    if ((rc = wait_rows(ix, st)) != 0) return rc;
    switch (ix->dtype) {
        case DT_F32: hipLaunchKernelGGL(widen_rows_kernel<DT_F32>, grid, block, 0, st, ix->rows, first, n, ix->dim, ix->dpad, dev_out); break;
        case DT_BF16: hipLaunchKernelGGL(widen_rows_kernel<DT_BF16>, grid, block, 0, st, ix->rows, first, n, ix->dim, ix->dpad, dev_out); break;
        default: hipLaunchKernelGGL(widen_rows_kernel<DT_F16>, grid, block, 0, st, ix->rows, first, n, ix->dim, ix->dpad, dev_out); break;
    }
    HIP_TRY(hipGetLastError());
    if (!ix->reader_done) HIP_TRY(hipEventCreateWithFlags(&ix->reader_done, hipEventDisableTiming));
    if (ix->reader_event_set && ix->reader_stream != st) HIP_TRY(hipStreamWaitEvent(st, ix->reader_done, 0));  // (one event: chain the readers too)
    HIP_TRY(hipEventRecord(ix->reader_done, st));
    ix->reader_stream = st;
    ix->reader_event_set = true;
    return CODD_KNN_OK;
}

int codd_knn_ivf_install(codd_knn_index* ix, const float* dev_centroids, int nlist, const int64_t* dev_perm,
                         const int64_t* dev_offsets, void* stream) {
    if (!ix || !dev_centroids || !dev_perm || !dev_offsets || nlist < 1 || nlist > (1 << 20)) return fail(CODD_KNN_EINVAL, "bad ivf_install arguments%s");
    if (ix->count < 1) return fail(CODD_KNN_EINVAL, "empty index%s");
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipDeviceSynchronize());
    // drop a previous layout
    void* old[] = {ix->rows_ivf, ix->ivf_ids, ix->ivf_offsets};
    for (void* b : old)
        if (b) (void)hipFree(b);
    ix->rows_ivf = nullptr; ix->ivf_ids = nullptr; ix->ivf_offsets = nullptr;
    if (ix->coarse) { (void)codd_knn_destroy(ix->coarse); ix->coarse = nullptr; }
    ix->ivf_epoch = -1;

    const size_t row_bytes = (size_t)ix->dpad * elem_size(ix->dtype);
    const int64_t n = ix->count;
    // validate the caller's tables before anything indexes memory with them: offsets on the host (nlist + 1 values),
    // the permutation on the device
    {
        std::vector<int64_t> off((size_t)nlist + 1);
        HIP_TRY(hipMemcpyAsync(off.data(), dev_offsets, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool ok = off[0] == 0 && off[(size_t)nlist] == n;
        for (int l = 0; ok && l < nlist; ++l) ok = off[(size_t)l] <= off[(size_t)l + 1];
        if (!ok) return fail(CODD_KNN_EINVAL, "ivf_install: offsets must start at 0, never decrease and end at the row count%s");
        unsigned* bad = nullptr;
        HIP_TRY(hipMalloc((void**)&bad, sizeof(unsigned)));
        hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned), st);
        unsigned host_bad = 1;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(perm_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dev_perm, n, bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&host_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(bad);
        if (e != hipSuccess) return fail(CODD_KNN_EDEVICE, "ivf_install: permutation check failed: %s", hipGetErrorString(e));
        if (host_bad) return fail(CODD_KNN_EINVAL, "ivf_install: permutation entries must lie in [0, count)%s");
    }
    auto drop_partial = [&]() {  // a failed install leaves no half-built layout (and no leak) behind
        void* part[] = {ix->rows_ivf, ix->ivf_ids, ix->ivf_offsets};
        for (void* b_ : part)
            if (b_) (void)hipFree(b_);
        ix->rows_ivf = nullptr; ix->ivf_ids = nullptr; ix->ivf_offsets = nullptr;
        if (ix->coarse) { (void)codd_knn_destroy(ix->coarse); ix->coarse = nullptr; }
    };
    int rc = CODD_KNN_OK;
    hipError_t he = hipMalloc(&ix->rows_ivf, (size_t)n * row_bytes);
    if (he == hipSuccess) he = hipMalloc((void**)&ix->ivf_ids, (size_t)n * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMalloc((void**)&ix->ivf_offsets, (size_t)(nlist + 1) * sizeof(int64_t));
    if (he == hipSuccess) he = hipMemcpyAsync(ix->ivf_offsets, dev_offsets, (size_t)(nlist + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const uint4*>(ix->rows), dev_perm, n,
                           (int)(row_bytes / 16), reinterpret_cast<uint4*>(ix->rows_ivf), ix->ivf_ids);
        he = hipGetLastError();
    }
    if (he != hipSuccess) {
        drop_partial();
        return fail(he == hipErrorOutOfMemory ? CODD_KNN_ENOMEM : CODD_KNN_EDEVICE, "ivf_install: building the list layout failed: %s", hipGetErrorString(he));
    }
    rc = codd_knn_create(&ix->coarse, ix->device, ix->dim, DT_F32, CODD_KNN_METRIC_COSINE);
    if (rc == 0) rc = codd_knn_upsert_device(ix->coarse, 0, dev_centroids, nlist, 1, stream);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = fail(CODD_KNN_EDEVICE, "ivf_install: device work failed%s");
    if (rc != 0) {
        drop_partial();
        return rc;
    }
    ix->ivf_nlist = nlist;
    ix->ivf_count = n;
    ix->ivf_epoch = ix->epoch;
    return CODD_KNN_OK;
}

int codd_knn_ivf_search(codd_knn_index* ix, const float* dev_queries, int B, int k, int nprobe, uint32_t row_base,
                        uint64_t* dev_keys, float* dev_dist, int64_t* dev_rows, void* stream) {
    if (!ix || !dev_queries) return fail(CODD_KNN_EINVAL, "null index or queries%s");
    if (B < 1 || B > CODD_KNN_MAX_BATCH) return fail(CODD_KNN_EINVAL, "B out of range [1,1024]%s");
    if (k < 1 || k > CODD_KNN_MAX_K) return fail(CODD_KNN_EINVAL, "k out of range [1,128]%s");
    if (!ix->coarse || ix->ivf_epoch != ix->epoch) return fail(CODD_KNN_EINVAL, "no IVF layout, or rows changed since codd_knn_ivf_install%s");
    if (nprobe < 1) return fail(CODD_KNN_EINVAL, "nprobe must be >= 1%s");
    if (nprobe > ix->ivf_nlist) nprobe = ix->ivf_nlist;
    if (nprobe > CODD_KNN_MAX_K) return fail(CODD_KNN_ENOTSUP, "nprobe above 128 is not supported (probe the whole index with codd_knn_search)%s");
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    WorkScope work(ix, st), work_coarse(ix->coarse, st);
    int rc;
    if ((rc = wait_rows(ix, st)) != 0) return rc;
    if ((rc = ensure_buf(&ix->qn, &ix->qn_cap, (int64_t)B * ix->dpad)) != 0) return rc;
    if ((rc = ensure_buf(&ix->probe_keys, &ix->probe_cap, (int64_t)B * nprobe)) != 0) return rc;
    if ((rc = launch_normalize(DT_F32, dev_queries, B, ix->dim, ix->dpad, 1, nullptr, 0, ix->qn, nullptr, st)) != 0) return rc;
    // 1. coarse: the nprobe best lists per query (exact scan of the centroids, tiny)
    if ((rc = exact_scan(ix->coarse, ix->qn, B, nprobe, 0u, ix->probe_keys, nullptr, nullptr, st)) != 0) return rc;
    const int nchunks = ix->dpad / elems_per_chunk(ix->dtype);
    const int niter = (nchunks + kWave - 1) / kWave;
    const int slots = k <= 64 ? 1 : 2;
    // 2a. a batch with enough (query, list) pairs to fill the chip without splitting lists: group the pairs by list on the
    //     device and scan every probed list once per kIvfNB of its queries (ivf_scan_shared_kernel)
    const int64_t npairs = (int64_t)B * nprobe;
    //     (worth it once a list is probed by two queries or more on average: below that every work item holds one pair and the
    //     grouping launches are pure overhead — 12.5M x 1024 fp16, 2,048 lists, B = 256: nprobe 8 5.98 ms per pair vs 7.22 shared)
    if (ix->ivf_share && npairs >= 1024 && npairs >= 2 * (int64_t)ix->ivf_nlist && !(ix->dtype != DT_F32 && niter == 4)) {
        const int nlist = ix->ivf_nlist;
        if ((rc = ensure_buf(&ix->ivf_group, &ix->ivf_group_cap, 3 * (int64_t)nlist + 2 + npairs)) != 0) return rc;
        if ((rc = ensure_buf(&ix->ivf_partial, &ix->ivf_partial_cap, npairs * k)) != 0) return rc;
        unsigned* cnt = ix->ivf_group;
        unsigned* pair_start = cnt + nlist;
        unsigned* item_start = pair_start + nlist + 1;
        unsigned* sorted_pairs = item_start + nlist + 1;
        {
        EvScope ev(ix, EV_SCAN, st);
        HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)nlist * sizeof(unsigned), st));
        HIP_TRY(hipMemsetAsync(ix->ivf_partial, 0, (size_t)(npairs * k) * sizeof(u64), st));  // (an empty probe slot stays an empty list)
        const unsigned pb = (unsigned)((npairs + 255) / 256);
        hipLaunchKernelGGL(ivf_pair_count_kernel, dim3(pb), dim3(256), 0, st, ix->probe_keys, (int)npairs, cnt);
        hipLaunchKernelGGL(ivf_pair_offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, nlist, pair_start, item_start);
        hipLaunchKernelGGL(ivf_pair_scatter_kernel, dim3(pb), dim3(256), 0, st, ix->probe_keys, (int)npairs, pair_start, cnt, sorted_pairs);
        // work items <= sum over lists of ceil(pairs / kIvfNB) <= min(pairs, lists + pairs / kIvfNB): the grid covers the bound, surplus workgroups leave at once
        const int64_t bound = std::min<int64_t>(npairs, (int64_t)nlist + npairs / kIvfNB);
#define CODD_IVFS_LAUNCH(DT, NI, SL)                                                                                              \
    hipLaunchKernelGGL((ivf_scan_shared_kernel<DT, NI, SL>), dim3((unsigned)bound), dim3(256), 0, st, ix->rows_ivf, ix->ivf_ids, \
                       ix->ivf_offsets, pair_start, item_start, sorted_pairs, nlist, nprobe, ix->dpad, ix->qn, k, row_base, ix->ivf_partial)
#define CODD_IVFS_NITER(DT, SL)                                       \
    switch (niter) {                                                  \
        case 1: CODD_IVFS_LAUNCH(DT, 1, SL); break;                   \
        case 2: CODD_IVFS_LAUNCH(DT, 2, SL); break;                   \
        case 3: CODD_IVFS_LAUNCH(DT, 3, SL); break;                   \
        default: CODD_IVFS_LAUNCH(DT, 4, SL); break;                  \
    }
        if (ix->dtype == DT_F32) { if (slots == 1) { CODD_IVFS_NITER(DT_F32, 1) } else { CODD_IVFS_NITER(DT_F32, 2) } }
        else if (ix->dtype == DT_BF16) { if (slots == 1) { CODD_IVFS_NITER(DT_BF16, 1) } else { CODD_IVFS_NITER(DT_BF16, 2) } }
        else { if (slots == 1) { CODD_IVFS_NITER(DT_F16, 1) } else { CODD_IVFS_NITER(DT_F16, 2) } }
#undef CODD_IVFS_NITER
#undef CODD_IVFS_LAUNCH
        HIP_TRY(hipGetLastError());
        }
        const int64_t ms = (int64_t)nprobe * k;
        return launch_merge(ix->ivf_partial, B, ms, ms, k, (u64*)dev_keys, dev_dist, dev_rows, st);
    }
    // 2. scan the probed lists; split each list over several blocks when the batch alone cannot fill the chip
    int split = (int)((4 * (int64_t)ix->num_cus + (int64_t)B * nprobe - 1) / ((int64_t)B * nprobe));
    split = split < 1 ? 1 : (split > 16 ? 16 : split);
    const int64_t m = (int64_t)nprobe * split * k;
    if ((rc = ensure_buf(&ix->ivf_partial, &ix->ivf_partial_cap, (int64_t)B * m)) != 0) return rc;
    const dim3 grid((unsigned)(nprobe * split), (unsigned)B), block(256);
#define CODD_IVF_LAUNCH(DT, NI, SL)                                                                                          \
    hipLaunchKernelGGL((ivf_scan_kernel<DT, NI, SL>), grid, block, 0, st, ix->rows_ivf, ix->ivf_ids, ix->ivf_offsets,       \
                       ix->probe_keys, nprobe, split, ix->dpad, ix->qn, k, row_base, ix->ivf_partial)
#define CODD_IVF_NITER(DT, SL)                                       \
    switch (niter) {                                                  \
        case 1: CODD_IVF_LAUNCH(DT, 1, SL); break;                    \
        case 2: CODD_IVF_LAUNCH(DT, 2, SL); break;                    \
        case 3: CODD_IVF_LAUNCH(DT, 3, SL); break;                    \
        default: CODD_IVF_LAUNCH(DT, 4, SL); break;                   \
    }
    {
        EvScope ev(ix, EV_SCAN, st);
        if (ix->dtype == DT_F32) { if (slots == 1) { CODD_IVF_NITER(DT_F32, 1) } else { CODD_IVF_NITER(DT_F32, 2) } }
        else if (ix->dtype == DT_BF16) { if (slots == 1) { CODD_IVF_NITER(DT_BF16, 1) } else { CODD_IVF_NITER(DT_BF16, 2) } }
        else { if (slots == 1) { CODD_IVF_NITER(DT_F16, 1) } else { CODD_IVF_NITER(DT_F16, 2) } }
    }
#undef CODD_IVF_NITER
#undef CODD_IVF_LAUNCH
    HIP_TRY(hipGetLastError());
    // 3. top-k of the nprobe*split partial lists
    return launch_merge(ix->ivf_partial, B, m, m, k, (u64*)dev_keys, dev_dist, dev_rows, st);
}

#ifdef CODD_I8_EXP_STAMPS
// diagnostic builds only (not in include/codd_knn.h): the per-wave phase stamps of the last i8_tile_kernel<FILTER> launch on `stream`'s
// workspace — [workgroup][wave][8] u64: issue, corpus wait, MFMA phase, end-of-interval sync, epilogue, intervals, total, tiles
int codd_knn_exp_read_stamps(codd_knn_index* ix, unsigned long long* host_out, int n_u64, void* stream) {
    if (!ix || !host_out) return CODD_KNN_EINVAL;
    WorkScope work(ix, (hipStream_t)stream);
    if (!ix->bucket_max || (int64_t)n_u64 > ix->bucket_cap) return CODD_KNN_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, ix->bucket_max, (size_t)n_u64 * 8, hipMemcpyDeviceToHost));
    return CODD_KNN_OK;
}
#endif

int codd_knn_set_option(codd_knn_index* ix, const char* key, int64_t value) {
    if (!ix || !key) return fail(CODD_KNN_EINVAL, "bad option arguments%s");
    if (strcmp(key, "scan_blocks_per_cu") == 0) {
        if (value < 1 || value > 8) return fail(CODD_KNN_EINVAL, "scan_blocks_per_cu must be in [1,8]%s");
        ix->scan_blocks_per_cu = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "filter") == 0) { ix->filter_enabled = value != 0; return CODD_KNN_OK; }
    if (strcmp(key, "filter_min_rows") == 0) { ix->filter_min_rows = value < 1 ? 1 : value; return CODD_KNN_OK; }
    if (strcmp(key, "filter_min_rows_small") == 0) { ix->filter_min_rows_small = value < 1 ? 1 : value; return CODD_KNN_OK; }
    if (strcmp(key, "filter_min_batch") == 0) { ix->filter_min_batch = value < 1 ? 1 : (int)value; return CODD_KNN_OK; }
    if (strcmp(key, "shadow8") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "shadow8 must be 0 or 1%s");
        ix->shadow8_enabled = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "shadow8_max_batch") == 0) {
        if (value < 1 || value > 256) return fail(CODD_KNN_EINVAL, "shadow8_max_batch must be in [1,256]%s");
        ix->shadow8_max_batch = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "shadow8_max_surv") == 0) {
        if (value < 1 || value > 1000000) return fail(CODD_KNN_EINVAL, "shadow8_max_surv must be in [1,1000000]%s");
        ix->shadow8_max_surv = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "shadow8_cooldown") == 0) {
        if (value < 0 || value > 1000000) return fail(CODD_KNN_EINVAL, "shadow8_cooldown must be in [0,1000000]%s");
        ix->shadow8_cooldown = (int)value;
        ix->cooldown_left = 0;
        return CODD_KNN_OK;
    }
#if CODD_EXPERIMENTS
    if (strcmp(key, "exp_slack_pct") == 0) {  // what-if timing only, experiment builds only: < 100 makes the int8 filter UNSOUND
        if (value < 1 || value > 100) return fail(CODD_KNN_EINVAL, "exp_slack_pct must be in [1,100]%s");
        ix->exp_slack_scale = (float)value / 100.0f;
        return CODD_KNN_OK;
    }
#endif
    if (strcmp(key, "debug_fail_shadow_alloc") == 0) {  // test hook: the bf16 shadow's allocation fails as if HBM were full
        ix->debug_fail_shadow_alloc = value != 0;
        if (!value) ix->shadow_nomem_epoch = -1;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "all_normalized") == 0) {
        // 0: the caller knows of rows that are not unit vectors (a persisted index written with normalize = 0): both filters
        // off, every search takes the exact scan.  The flag cannot be switched back on from outside.
        if (value != 0) return fail(CODD_KNN_EINVAL, "all_normalized can only be cleared (0)%s");
        ix->all_normalized = false;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "i8v2") == 0) {
        if (value < 0 || value > 2) return fail(CODD_KNN_EINVAL, "i8v2 must be 0, 1 or 2%s");
        ix->i8v2 = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "f16_tile") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "f16_tile must be 0 or 1%s");
        ix->f16_tile = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "ivf_share") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "ivf_share must be 0 or 1%s");
        ix->ivf_share = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "fuse_fallback") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "fuse_fallback must be 0 or 1%s");
        ix->fuse_fallback = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "small_batch_max") == 0) {
        if (value < 0 || value > 1) return fail(CODD_KNN_EINVAL, "small_batch_max must be 0 or 1%s");
        ix->small_batch_max = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "per_block") == 0) {
        if (value < 0 || value > 7) return fail(CODD_KNN_EINVAL, "per_block must be in [0,7]%s");
        ix->per_block = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "i8_pair") == 0) {
        if (value < 0 || value > 2) return fail(CODD_KNN_EINVAL, "i8_pair must be 0, 1 or 2%s");
        ix->i8_pair = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "i8v2_half") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "i8v2_half must be 0 or 1%s");
        ix->i8v2_half = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "resident_q") == 0) {
        if (value != 0 && value != 1) return fail(CODD_KNN_EINVAL, "resident_q must be 0 or 1%s");
        ix->resident_q = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "sample_rounds8") == 0) {
        if (value < 1 || value > 16) return fail(CODD_KNN_EINVAL, "sample_rounds8 must be in [1,16]%s");
        ix->sample_rounds8 = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "sample_div8") == 0) {
        if (value < 1 || value > 1000) return fail(CODD_KNN_EINVAL, "sample_div8 must be in [1,1000]%s");
        ix->sample_div8 = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "sample_tiles") == 0) {
        if (value < 1 || value > 65536) return fail(CODD_KNN_EINVAL, "sample_tiles must be in [1,65536]%s");
        ix->sample_tiles = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "sample_div") == 0) {
        if (value < 1 || value > 4096) return fail(CODD_KNN_EINVAL, "sample_div must be in [1,4096]%s");
        ix->sample_div = (int)value;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "hit_cap") == 0) {
        if (value < 16 || value > (1 << 20)) return fail(CODD_KNN_EINVAL, "hit_cap must be in [16,2^20]%s");
        ix->hit_cap_q = (int)value;
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
        ix->ev_kind.assign(ix->ev.size() / 2, 0);
        ix->profile = value > 0;
        ix->ev_used = 0;
        return CODD_KNN_OK;
    }
    return fail(CODD_KNN_EINVAL, "unknown option: %s", key);
}

int codd_knn_get_stat(const codd_knn_index* ix, const char* key, int64_t* out) {
    if (!ix || !key || !out) return fail(CODD_KNN_EINVAL, "bad stat arguments%s");
    // "time_ns:<kernel>" / "events:<kernel>" with kernel in {scan, filter, sample, finalize}
    const bool want_time = strncmp(key, "time_ns:", 8) == 0, want_events = strncmp(key, "events:", 7) == 0;
    if (want_time || want_events) {
        const char* name = key + (want_time ? 8 : 7);
        int kind = -1;
        for (int i = 0; i < EV_KINDS; ++i)
            if (strcmp(name, kEvNames[i]) == 0) kind = i;
        if (kind < 0) return fail(CODD_KNN_EINVAL, "unknown kernel name in stat: %s", key);
        double total_ms = 0.0;
        int64_t events = 0;
        if (ix->ev_used > 0) {
            DeviceGuard guard(ix->device);
            HIP_TRY(hipEventSynchronize(ix->ev[2 * ix->ev_used - 1]));
            for (int i = 0; i < ix->ev_used; ++i) {
                if (ix->ev_kind[i] != kind) continue;
                float ms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms, ix->ev[2 * i], ix->ev[2 * i + 1]));
                total_ms += ms;
                events++;
            }
        }
        *out = want_time ? (int64_t)(total_ms * 1e6) : events;
        return CODD_KNN_OK;
    }
    if (strcmp(key, "searches") == 0) *out = ix->stat_searches;
    else if (strcmp(key, "scan_launches") == 0) *out = ix->stat_scan_launches;
    else if (strcmp(key, "last_scan_blocks") == 0) *out = ix->stat_last_scan_blocks;
    else if (strcmp(key, "filter_passes") == 0) *out = ix->stat_filter_passes;
    else if (strcmp(key, "shadow8_builds") == 0) *out = ix->stat_shadow8_builds;
    else if (strcmp(key, "shadow16_builds") == 0) *out = ix->stat_shadow_builds;
    else if (strcmp(key, "shadow16_alloc_failures") == 0) *out = ix->stat_shadow_nomem;
    else if (strcmp(key, "all_normalized") == 0) *out = ix->all_normalized ? 1 : 0;
    else if (strcmp(key, "shadow8_passes") == 0) *out = ix->stat_shadow8_passes;
    else if (strcmp(key, "i8v2_passes") == 0) *out = ix->stat_i8v2_passes;
    else if (strcmp(key, "f16_tile_passes") == 0) *out = ix->stat_f16_tile_passes;
    else if (strcmp(key, "small_batch_passes") == 0) *out = ix->stat_small_batch;
    else if (strcmp(key, "shadow8_cooldowns") == 0) *out = ix->stat_cooldowns;
    else if (strcmp(key, "shadow8_wide_blocks") == 0) {  // blocks whose error norm exceeds shadow8_max_eps, as last read back
        if (ix->eps_r_copied && hipEventQuery(ix->eps_r_copied) == hipSuccess) *out = (int64_t)reinterpret_cast<const unsigned*>(ix->eps_r_host)[1];
        else {
            (void)hipGetLastError();
            *out = ix->wide_blocks_known;
        }
    }
    else if (strcmp(key, "shadow8_eps_r_micro") == 0) {  // worst row's quantisation error norm x 1e6, as last read back
        if (ix->eps_r_copied && hipEventQuery(ix->eps_r_copied) == hipSuccess) *out = (int64_t)(ix->eps_r_host[0] * 1e6f);
        else {
            (void)hipGetLastError();
            *out = (int64_t)(ix->eps_r_known * 1e6f);
        }
    }
    else if (strcmp(key, "fallback_queries") == 0 || strcmp(key, "filter_hits") == 0 || strcmp(key, "filter_survivors") == 0) {
        // device-side counters (the search itself never reads them back): synchronises
        unsigned long long h[4] = {0, 0, 0, 0};
        if (ix->dstats) {
            DeviceGuard guard(ix->device);
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(h, ix->dstats, sizeof(h), hipMemcpyDeviceToHost));
        }
        *out = (int64_t)(key[0] == 'f' && key[1] == 'a' ? h[2] : (strcmp(key, "filter_hits") == 0 ? h[0] : h[1]));
    }
    else if (strcmp(key, "capacity_rows") == 0) *out = ix->capacity;
    else if (strcmp(key, "num_cus") == 0) *out = ix->num_cus;
    else if (strcmp(key, "device_bytes") == 0) {
        int64_t b = ix->capacity * (int64_t)ix->dpad * (int64_t)elem_size(ix->dtype) + ix->shadow_rows * (int64_t)ix->dpad * 2 +
                    ix->shadow8_rows * ((int64_t)dpad8_of(ix) + 4);
        for (const WorkSlot& w : ix->slots)
            b += w.bufs.qn_cap * 4 + w.bufs.partial_cap * 8 + w.bufs.keys_tmp_cap * 8 + w.bufs.hits_cap * 8 + w.bufs.bucket_cap * 8 +
                 w.bufs.qfrag_cap * 16 + w.bufs.fb_partial_cap * 8 + w.bufs.probe_cap * 8 + w.bufs.ivf_partial_cap * 8;
        *out = b;
    } else if (strcmp(key, "workspaces") == 0) {
        int64_t c = 0;
        for (const WorkSlot& w : ix->slots) c += w.used ? 1 : 0;
        *out = c;
    }
    else return fail(CODD_KNN_EINVAL, "unknown stat: %s", key);
    return CODD_KNN_OK;
}

}  // extern "C"
