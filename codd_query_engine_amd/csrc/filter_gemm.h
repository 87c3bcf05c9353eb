// filter_gemm.h — the large-batch leg of the search: a bf16 MFMA GEMM that never materialises the
// B x N score matrix.  It FILTERS: for every query it emits the rows whose approximate score
// clears a per-query threshold that provably keeps every true top-k row; the
// survivors are re-scored exactly (canonical fp32) by finalize_kernel.
//
// Data layout (both operands are stored in MFMA-fragment order, so every wave-level load is one
// contiguous 1 KiB piece and LDS reads are conflict-free without a swizzle); shown for the shipped
// v_mfma_f32_16x16x32_bf16 shape (shadow_piece_index / qfrag_piece_index are the definition; the
// 32x32x16 order is kept behind -DCODD_MFMA16=0):
//
//   shadow (bf16 copy of the corpus, written at ingest), in 16-byte pieces of 8 elements:
//       piece[((block*nsteps + s)*4 + (rs*2 + ks))*64 + (kq*16 + r)]
//                                   = C[row = 32*block + 16*rs + r][k = 64s + 32ks + 8kq + 0..7]
//   qfrag (bf16 query block, written by qfrag_kernel per search):
//       piece[(s*32 + qb*2 + ks)*64 + (kq*16 + c)] = Q[query = 16*qb + c][k = 64s + 32ks + 8kq + 0..7]
//
// One MFMA consumes, per lane (r|c = lane&15, kq = lane>>4), exactly one such piece of each operand;
// the k order inside an instruction is permuted identically on both sides, which a dot product does
// not notice.
//
// Workgroup = 8 waves = one tile of 256 corpus rows x 256 queries, K-step 64:
//   * wave w owns corpus rows [32w, 32w+32) of the tile and ALL 256 queries: 2 x 16 accumulator blocks
//     of 16x16 (128 VGPRs).  Its corpus fragments go HBM -> registers directly (each corpus byte
//     enters the CU once, is used by one wave: no LDS round trip), three K-steps deep in a
//     register ring so ~12 KiB per wave stays in flight;
//   * the query K-slice (32 KiB, L2-resident, shared by the 8 waves) is double-buffered in LDS in
//     stages of two slices, staged through registers (issue early, write late) so that every load in
//     the kernel is an ordinary counted load and __syncthreads() stays a bare s_barrier;
//   * one barrier per stage; 64 MFMAs + 32 ds_read_b128 per wave per K-step.
#pragma once
#include <type_traits>

#include "row_traits.h"
#include "wave_topk.h"

namespace codd {

#ifndef CODD_SHADOW_F16
#define CODD_SHADOW_F16 1    // the 2-byte shadow's element type: 1 = fp16 (11-bit significand: eps 0.0011 at 768-d), 0 = bf16 (8 bits: eps 0.0079).
                             // Same MFMA rate, same bytes; fp16 since round 3: this filter is what an index falls back to when the int8 one (eps ~0.02)
                             // cannot separate a query's neighbourhood from the rest of a dense cluster, and there a 7x tighter slack is worth more than
                             // anything else (profiles/r3/clustered_corpora.txt).  Unit-norm operands cannot overflow fp16; its subnormal grid is in the bound.
#endif
#if CODD_SHADOW_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;  // "shadow element x8" (name kept for the bf16 default)
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kTileRows = 256;
constexpr int kTileQ = 256;
constexpr int kStagePieces = 2048;  // 16-byte pieces of one query K-slice (256 q x 64 k bf16 = 32 KiB)
#ifndef CODD_RB
#define CODD_RB 1            // 32-row corpus blocks per wave: 1 -> 8 waves x 256 regs, 2 -> 4 waves x 512 regs
#endif
#ifndef CODD_QSPLIT
#define CODD_QSPLIT 1        // 2: two waves share each row group and take half of the query blocks each
#endif
constexpr int kRB = CODD_RB;
constexpr int kQSplit = CODD_QSPLIT;
constexpr int kRowGroups = 8 / kRB;                   // row groups of a 256-row tile
constexpr int kFilterWaves = kRowGroups * kQSplit;
constexpr int kFilterThreads = 64 * kFilterWaves;
constexpr int kQP = kStagePieces / kFilterThreads;  // query-slice pieces each thread stages per K-step
#ifndef CODD_MFMA16
#define CODD_MFMA16 1        // 1: v_mfma_f32_16x16x32_bf16 (16-row x 16-query blocks, K 32) instead of 32x32x16
#endif
// MFMA block geometry.  32x32x16: a wave's 32 rows are ONE row block, a 64-wide K-step is 4 sub-steps, a
// 32-query block per accumulator (16 regs).  16x16x32: TWO row blocks of 16, 2 sub-steps, 16-query blocks
// (4 regs).  Either way a wave-level operand piece is 64 lanes x 16 B = 1 KiB and an accumulator set is 128 regs.
constexpr int kMB = CODD_MFMA16 ? 16 : 32;         // rows (and queries) per MFMA block
constexpr int kAccRegs = CODD_MFMA16 ? 4 : 16;
constexpr int kKS = CODD_MFMA16 ? 2 : 4;           // MFMA K sub-steps per 64-wide K-step
constexpr int kRS = (32 / kMB) * kRB;              // row blocks per wave
constexpr int kQBper32 = 32 / kMB;                 // query blocks per 32 queries
static_assert(kFilterWaves <= 8, "at most 8 waves per workgroup");
#if CODD_MFMA16
typedef __attribute__((ext_vector_type(4))) float acc_t;
#else
typedef __attribute__((ext_vector_type(16))) float acc_t;
#endif
// which of the 4 corpus pieces of a (32-row block, K-step) feeds row block rs at sub-step ks
__host__ __device__ constexpr int a_piece(int rs_in_block, int ks) { return CODD_MFMA16 ? rs_in_block * 2 + ks : ks; }
// which piece of a query K-slice feeds query block qb at sub-step ks
__host__ __device__ constexpr int b_piece(int qb, int ks) { return CODD_MFMA16 ? qb * 2 + ks : qb * 4 + ks; }
// accumulator register r of a lane -> row offset inside the MFMA row block
__device__ __forceinline__ int acc_row(int r, int lane) {
    return CODD_MFMA16 ? 4 * (lane >> 4) + r : 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
}
// build-time experiment switches (defaults = the shipped configuration)
#ifndef CODD_EXPERIMENTS
#define CODD_EXPERIMENTS 0
#endif
#if !CODD_EXPERIMENTS && (defined(CODD_NO_EPILOGUE) || defined(CODD_EXP_SAME_TILE) || defined(CODD_EXP_NO_QSTAGE) || defined(CODD_EXP_NB) || \
                          defined(CODD_EXP_NO_HITS) || defined(CODD_EXP_NO_LDSREAD) || defined(CODD_EXP_NO_FLUSH) || defined(CODD_EXP_NO_BARRIER))
#error "the CODD_EXP_* / CODD_NO_EPILOGUE switches return wrong results or race: they exist only in -DCODD_EXPERIMENTS=1 builds (build_variant)"
#endif
#ifndef CODD_QS
#define CODD_QS 2            // 64-wide query K-slices per LDS stage = K-steps per workgroup barrier
#endif
#ifndef CODD_PIN_SCHEDULE
#define CODD_PIN_SCHEDULE 1  // sched_group_barrier shape of a K-step
#endif
#ifndef CODD_MFMA_PRIO
#define CODD_MFMA_PRIO 0     // s_setprio(1) around the MFMA cluster
#endif
#ifndef CODD_STATIC_PRIO
#define CODD_STATIC_PRIO 0   // waves 4..7 (the second wave of every SIMD) run at s_setprio 1 for the whole kernel
#endif
#ifndef CODD_NT_LOADS
#define CODD_NT_LOADS 1      // corpus fragments with the non-temporal cache policy (read-once stream)
#endif
#ifndef CODD_BLOCKED_TILES
#define CODD_BLOCKED_TILES 0 // 1: workgroup b walks a contiguous range of tiles instead of b, b+G, b+2G, ...
#endif
#ifndef CODD_NO_EPILOGUE
#define CODD_NO_EPILOGUE 0   // diagnostic only: skip the threshold test (results are wrong)
#endif
#ifndef CODD_EXP_SAME_TILE
#define CODD_EXP_SAME_TILE 0 // diagnostic only: every step re-reads tile 0 (L2-resident corpus)
#endif
#ifndef CODD_EXP_NO_QSTAGE
#define CODD_EXP_NO_QSTAGE 0 // diagnostic only: no query staging loads / LDS writes
#endif
#ifndef CODD_EXP_NB
#define CODD_EXP_NB 8        // diagnostic only: query blocks actually multiplied (8 = all)
#endif
#ifndef CODD_EXP_NO_HITS
#define CODD_EXP_NO_HITS 0   // diagnostic only: thresholds forced to +inf
#endif
#ifndef CODD_EXP_NO_LDSREAD
#define CODD_EXP_NO_LDSREAD 0 // diagnostic only: the query fragment is taken from a register, not from LDS
#endif
#ifndef CODD_EXP_NO_FLUSH
#define CODD_EXP_NO_FLUSH 0  // diagnostic only: the final hit list is dropped (results are wrong)
#endif
#ifndef CODD_EXP_NO_BARRIER
#define CODD_EXP_NO_BARRIER 0 // diagnostic only: no stage barriers (racy)
#endif
constexpr int kQS = CODD_QS;
constexpr int kLdsQBytes = 2 * kQS * kStagePieces * 16;
constexpr int kHitCap = kQS == 1 ? 4096 : 2048;  // per-workgroup LDS hit list (entries of 3 dwords)
#ifndef CODD_HITCNT_STRIDE
#define CODD_HITCNT_STRIDE 1  // dwords between two queries' global hit counters (32 = one 128-byte line each)
#endif
#ifndef CODD_FLUSH_AT
#define CODD_FLUSH_AT (kHitCap / 2)  // a workgroup empties its LDS hit list at the first tile end with more entries
#endif
constexpr int kHitCntStride = CODD_HITCNT_STRIDE;
#ifndef CODD_RING
#define CODD_RING 3
#endif
#ifndef CODD_STAGGER
#define CODD_STAGGER 0  // 1: waves 4..7 run their MFMA halves half a K-step behind waves 0..3 (one barrier per K-step)
#endif
#ifndef CODD_QDEPTH
#define CODD_QDEPTH 1   // 2: the query slice of step t+3 is requested during step t (one step more for the L2 round trip), 16 more VGPRs
#endif
constexpr int kQD = CODD_QDEPTH;
constexpr int kUnroll = CODD_RING * CODD_QDEPTH;  // steps per unrolled loop body: every register buffer index is static
constexpr int kRing = CODD_RING;        // corpus-fragment register ring (K-steps)
constexpr int kPrefetch = CODD_RING - 1;  // K-steps the corpus loads run ahead

enum { MODE_FILTER = 0, MODE_SAMPLE = 1, MODE_DUMP = 2 };
constexpr int kLdsWords = 576;  // dwords of per-workgroup bookkeeping behind the query slices (256 thresholds + counter, or 256 u64 keys)

// flags[] words shared with the host
enum { FLAG_WG_OVERFLOW = 0 /* statistics: a workgroup hit list filled up */, FLAG_NEED_FALLBACK = 1 /* any query queued */, FLAG_WORDS = 4 };

// 16-byte piece (8 consecutive elements, c8 = element/8) of `row` in the shadow / of query q in qfrag.
// 32x32x16: lane (h = lane>>5, r = lane&31) of sub-step kk holds k = 64s + 32h + 8kk + 0..7
// 16x16x32: lane (kq = lane>>4, r = lane&15) of sub-step ks holds k = 64s + 32ks + 8kq + 0..7
__host__ __device__ inline int64_t shadow_piece_index(int64_t row, int c8, int nsteps) {
    const int64_t block = row >> 5;
    const int s = c8 >> 3;
#if CODD_MFMA16
    const int rs = (int)(row >> 4) & 1, r = (int)(row & 15), ks = (c8 >> 2) & 1, kq = c8 & 3;
    return ((block * nsteps + s) * 4 + a_piece(rs, ks)) * 64 + (kq * 16 + r);
#else
    const int r = (int)(row & 31), h = (c8 >> 2) & 1, kk = c8 & 3;
    return ((block * nsteps + s) * 4 + kk) * 64 + (h * 32 + r);
#endif
}
__host__ __device__ inline int64_t qfrag_piece_index(int q, int c8) {
    const int s = c8 >> 3;
#if CODD_MFMA16
    const int qb = q >> 4, c = q & 15, ks = (c8 >> 2) & 1, kq = c8 & 3;
    return ((int64_t)s * 32 + b_piece(qb, ks)) * 64 + (kq * 16 + c);
#else
    const int nb = q >> 5, c = q & 31, h = (c8 >> 2) & 1, kk = c8 & 3;
    return (((int64_t)s * 8 + nb) * 4 + kk) * 64 + (h * 32 + c);
#endif
}

// two fp32 values -> two shadow elements (bf16, or fp16 under CODD_SHADOW_F16), round to nearest even
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if CODD_SHADOW_F16
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)lo, (_Float16)hi};  // v_cvt_f16_f32, RNE; |x| <= 1 here, no overflow
    return __builtin_bit_cast(uint32_t, v);
#endif
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    a = ((a & 0x7fffffffu) > 0x7f800000u) ? ((a >> 16) | 0x0040u) : ((a + 0x7fffu + ((a >> 16) & 1u)) >> 16);
    b = ((b & 0x7fffffffu) > 0x7f800000u) ? ((b >> 16) | 0x0040u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    return (a & 0xffffu) | (b << 16);
}

// qn: [B][dpad] fp32 normalised queries -> qfrag pieces (queries >= B are zero).  One thread per piece.
// Also zeroes the pass's control block (ctl_words dwords) so that no separate memset launch is needed.
__global__ __launch_bounds__(256) void qfrag_kernel(const float* __restrict__ qn, int B, int dpad, uint4* __restrict__ qfrag,
                                                    unsigned* __restrict__ ctl, int ctl_words) {
    const int npieces = kTileQ * (dpad >> 3);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < ctl_words) ctl[idx] = 0u;
    if (idx >= npieces) return;
    const int q = idx / (dpad >> 3), c8 = idx % (dpad >> 3);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (q < B) {
        const float4* p = reinterpret_cast<const float4*>(qn + (int64_t)q * dpad + c8 * 8);
        const float4 a = p[0], b = p[1];
        v = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
    qfrag[qfrag_piece_index(q, c8)] = v;
}

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{}), in order
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// append a workgroup's LDS hit list (entries: score bits, row, query) to the per-query global lists
// qs (optional, LDS): per-query factor still to be applied to the stored scores (filter_i8.h stores acc * rscale and leaves
// the query's scale to the flush)
__device__ __forceinline__ void flush_hits(const unsigned* lds_hits, unsigned m, int tid, u64* __restrict__ hits,
                                           unsigned* __restrict__ hit_cnt, int cap_q, const float* qs = nullptr) {
    for (unsigned e = tid; e < m; e += kFilterThreads) {
        const unsigned row = lds_hits[e * 3 + 1], q = lds_hits[e * 3 + 2];
        float v = __uint_as_float(lds_hits[e * 3 + 0]);
        if (qs) v *= qs[q];
        const unsigned slot = atomicAdd(&hit_cnt[q * kHitCntStride], 1u);
        if (slot < (unsigned)cap_q) hits[(int64_t)q * cap_q + slot] = make_key(v, row);
    }
}

// The same append for the list a workgroup is left with when it has run out of tiles (the common case: ~1000 hits,
// all 256 workgroups at once).  One device-scope atomic per hit made that moment ~0.12 ms of every launch, whatever
// the row count (profiles/r1/v5_hit_flush_ablation.txt); here the entries are first counted per query in LDS and every
// query reserves its whole range with ONE atomic.  cnt256 is a scratch of 256 words (the thresholds, dead by now).
#ifndef CODD_BALLOT_HITS
#define CODD_BALLOT_HITS 0   // 1: the epilogue reserves LDS hit slots per accumulator block (ballots) instead of per register; measured 2 % slower
#endif
#ifndef CODD_BINNED_INLOOP
#define CODD_BINNED_INLOOP 0  // 1: the flushes inside the loop reserve per-query ranges too (int8 kernel -1 %, bf16 kernel +10 % time: register allocation; off)
#endif
#ifndef CODD_BINNED_FLUSH
#define CODD_BINNED_FLUSH 1
#endif
__device__ __forceinline__ void flush_hits_binned(const unsigned* lds_hits, unsigned m, int tid, unsigned* cnt256,
                                                  u64* __restrict__ hits, unsigned* __restrict__ hit_cnt, int cap_q, const float* qs = nullptr) {
    constexpr int kPer = kHitCap / kFilterThreads;
    static_assert(kHitCap % kFilterThreads == 0, "hit list is a whole number of entries per thread");
    if (tid < 256) cnt256[tid] = 0u;
    __syncthreads();
    unsigned rank[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const unsigned e = (unsigned)tid + (unsigned)i * kFilterThreads;
        rank[i] = e < m ? atomicAdd(&cnt256[lds_hits[e * 3 + 2]], 1u) : 0u;
    }
    __syncthreads();
    if (tid < 256) {
        const unsigned c = cnt256[tid];
        cnt256[tid] = c ? atomicAdd(&hit_cnt[tid * kHitCntStride], c) : 0u;  // a poisoned counter stays poisoned
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const unsigned e = (unsigned)tid + (unsigned)i * kFilterThreads;
        if (e < m) {
            const unsigned q = lds_hits[e * 3 + 2];
            const unsigned slot = cnt256[q] + rank[i];
            float v = __uint_as_float(lds_hits[e * 3 + 0]);
            if (qs) v *= qs[q];
            if (slot < (unsigned)cap_q) hits[(int64_t)q * cap_q + slot] = make_key(v, lds_hits[e * 3 + 1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_filter_kernel<MODE, NBQ>   (NBQ = 32-query blocks actually multiplied: 1, 2, 4 or 8; a small
//   batch pays only for its own MFMAs and LDS traffic and the kernel turns into a pure HBM stream)
//   MODE_FILTER: every (query, row) with approx score >= thr[query] is appended to hits[query][]
//   MODE_SAMPLE: per (tile, query) the best (approx score, row) key is written to bucket_key[query][tile]
//   MODE_DUMP  : all scores to dump[query][row] (diagnostics / layout tests, small n only)
// Run-tile u (0 <= u < ntiles_run) is corpus tile u*tile_stride; workgroup b takes u = b, b+G, ...
// ---------------------------------------------------------------------------------------------
// EL = 1: the operands are int8 with one fp32 scale per corpus row (rscale) and per query (qscale): 128 elements per
// K-step instead of 64, v_mfma_i32_16x16x64_i8 (twice the bf16 rate, half the bytes), exact integer accumulation; a
// score is acc * rscale[row] * qscale[query].  The piece / ring / LDS geometry is byte-identical to the bf16 mode.
// RES = 1 (only launched when the row has at most 4 K-steps, i.e. the whole query block fits the 4 LDS slices): the
// query block is loaded into LDS once per workgroup and never re-staged — no staging loads, LDS writes or stage barriers
// in the loop (re-staging costs 15-19 % of the kernel, profiles/r1/v5_hit_flush_ablation.txt).
template <int MODE, int NBQ, int EL = 0, int RES = 0>
__global__ __launch_bounds__(kFilterThreads, kFilterWaves == 8 ? 2 : 1) void gemm_filter_kernel(
    const uint4* __restrict__ shadow, const uint4* __restrict__ qfrag, int64_t n, int nsteps, int64_t ntiles_run,
    int64_t tile_stride, const float* __restrict__ thr, u64* __restrict__ bucket_key, u64* __restrict__ hits,
    unsigned* __restrict__ hit_cnt, int cap_q, unsigned* __restrict__ flags, float* __restrict__ dump,
    const float* __restrict__ rscale = nullptr, const float* __restrict__ qscale = nullptr) {
    static_assert(EL == 0 || (CODD_MFMA16 && kKS == 2), "the int8 mode is written for the 16x16 MFMA geometry");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<EL == 1, i32x4, acc_t>::type accv_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* ldsQ = reinterpret_cast<uint4*>(smem);                        // 2 stages x kQS slices x 32 KiB
    unsigned* lds_w = reinterpret_cast<unsigned*>(smem + kLdsQBytes);    // [0..255] thr / bucket max, [256] hit count
    unsigned* lds_hits = lds_w + kLdsWords;                              // kHitCap x 3 dwords (FILTER only)
    u64* lds_k = reinterpret_cast<u64*>(lds_w);                          // SAMPLE: [0..255] best (score, row) key of the tile per query

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = kQSplit == 1 ? wave : wave % kRowGroups;  // row group of the tile this wave owns
    const int qh = kQSplit == 1 ? 0 : wave / kRowGroups;     // which share of the query blocks
    const int c = lane & (kMB - 1);    // query inside an MFMA query block
    const int qoff = qh * b_piece(NBQ * kQBper32 / kQSplit, 0) * 64;  // first LDS piece of this wave's query blocks
    const int qbase = qh * (NBQ * kQBper32 / kQSplit) * kMB + c;     // this lane's query in query block 0 of the wave

    const int64_t G = gridDim.x;
#if CODD_BLOCKED_TILES
    const int64_t per_wg = (ntiles_run + G - 1) / G;
    const int64_t first_u = (int64_t)blockIdx.x * per_wg, step_u = 1;
    const int64_t my_tiles = first_u >= ntiles_run ? 0 : (first_u + per_wg <= ntiles_run ? per_wg : ntiles_run - first_u);
#else
    const int64_t first_u = blockIdx.x, step_u = G;
    const int64_t my_tiles = ntiles_run > (int64_t)blockIdx.x ? (ntiles_run - blockIdx.x + G - 1) / G : 0;
#endif
    const int T = (int)(my_tiles * nsteps);  // K-steps of this workgroup (< 2^30: rows < 2^32, nsteps <= 64)
    if (T == 0) return;

    if (MODE == MODE_FILTER) {
        // int8: the test runs on acc * rscale[row], so the threshold carries the query's scale (a zero query has
        // scale 0: every score is 0 and its threshold becomes -inf or +inf by sign)
        if (tid < 256) lds_w[tid] = __float_as_uint(EL ? thr[tid] / qscale[tid] : thr[tid]);
        if (tid == 0) lds_w[256] = 0u;
    } else if (MODE == MODE_SAMPLE) {
        if (tid < 256) lds_k[tid] = 0ull;
    }

    constexpr int kSP = NBQ * 256;  // 16-byte pieces of a query slice that are actually staged
    constexpr int kQPn = kSP / kFilterThreads > 0 ? kSP / kFilterThreads : 1;
    constexpr int NQBall = NBQ * kQBper32;       // MFMA query blocks of the pass
    constexpr int NQB = NQBall / kQSplit;        // ... of this wave
    static_assert(NQBall % kQSplit == 0, "query blocks must split evenly");
    accv_t acc[kRS][NQB];
#pragma unroll
    for (int rs = 0; rs < kRS; ++rs)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int i = 0; i < kAccRegs; ++i) acc[rs][qb][i] = 0;

    // load cursor (runs kPrefetch steps ahead of the compute cursor)
    int64_t l_u = first_u;  // run-tile ordinal
    int l_s = 0;
    // always issues its 4 loads (a conditional load would make hipcc's vmcnt bookkeeping assume the
    // worst at every join): past the last step the cursor simply stays on the last valid slice
    int l_left = T;
    auto load_a = [&](uint4(&dst)[kRB][4]) {
#pragma unroll
        for (int rb = 0; rb < kRB; ++rb) {
            const int64_t block = (CODD_EXP_SAME_TILE ? 0 : l_u * tile_stride * 8) + wr * kRB + rb;
            const uint4* p = shadow + ((block * nsteps + l_s) * 4) * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#if CODD_NT_LOADS
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + kk * 64));
                dst[rb][kk] = make_uint4(t.x, t.y, t.z, t.w);
#else
                dst[rb][kk] = p[kk * 64];
#endif
            }
        }
        if (--l_left > 0) {
            if (++l_s == nsteps) { l_s = 0; l_u += step_u; }
        }
    };

    uint4 ring[kRing][kRB][4];
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) load_a(ring[i]);

    // The query operand is a cyclic stream of 64-wide K slices (slice of step t = t mod nsteps),
    // consumed through LDS stages of kQS slices: stage g holds the slices of steps [g*kQS, (g+1)*kQS).
    // Stage 0 is filled here; during step t each thread fetches its part of the slice of step t+kQS
    // and writes it into the other stage after its MFMAs; one barrier per STAGE, not per step.
    int q_s = 0;  // slice that the next staging load fetches
    // every thread stages (loads are unconditional, see load_a); when the slice has fewer pieces than
    // threads, two threads copy the same piece to the same place
    const int stid = kSP >= kFilterThreads ? tid : tid % kSP;
    typedef unsigned q4u __attribute__((ext_vector_type(4)));  // (native vectors: uint4 staging buffers end up as scratch allocas)
#pragma unroll
    for (int sub = 0; sub < (RES ? 2 * kQS : kQS); ++sub) {
        if (RES == 1 && sub >= nsteps) break;  // resident: slice s lives in LDS slice s for the whole kernel
        const uint4* src = qfrag + (int64_t)q_s * kStagePieces + stid;
        uint4* dst = ldsQ + sub * kStagePieces + stid;
        q4u tmp[kQPn];
#pragma unroll
        for (int j = 0; j < kQPn; ++j) tmp[j] = reinterpret_cast<const q4u*>(src)[j * kFilterThreads];
#pragma unroll
        for (int j = 0; j < kQPn; ++j) reinterpret_cast<q4u*>(dst)[j * kFilterThreads] = tmp[j];
        q_s = q_s + 1 == nsteps ? 0 : q_s + 1;
    }
    q4u qreg0[kQPn], qreg1[kQPn];
    if (kQD == 2) {  // the slice step 0 will write to LDS is already on its way
        const uint4* src = qfrag + (int64_t)q_s * kStagePieces + stid;
#pragma unroll
        for (int j = 0; j < kQPn; ++j) qreg1[j] = reinterpret_cast<const q4u*>(src)[j * kFilterThreads];
        q_s = q_s + 1 == nsteps ? 0 : q_s + 1;
    }
    __syncthreads();

#if CODD_STATIC_PRIO
    // the two waves of a SIMD otherwise run in lockstep (same barriers, age-based arbitration) and do
    // their non-MFMA work at the same time; a standing priority makes one of them take the matrix pipe
    // for its whole cluster while the other loads/stages, which staggers them by half a step
    if (__builtin_amdgcn_readfirstlane(tid) >= kFilterThreads / 2) __builtin_amdgcn_s_setprio(CODD_STATIC_PRIO);
#endif
    // ------------------------------------------------------------------------------------------------------------
    // Main loop.  An *interval* is the time between two workgroup barriers and covers one 64-wide K-step of work per
    // wave.  Query staging (loads of slice t+kQS at the top, LDS writes at the bottom) is tied to the interval for
    // every wave.  With CODD_STAGGER the second-dispatched half of the workgroup (waves 4..7 = the SIMD partners of
    // waves 0..3) runs its MFMA work half a K-step late: in interval t it multiplies the second K-half of step t-1,
    // then the first K-half of step t, and issues its corpus loads between the two.  The two waves of a SIMD then reach
    // their loads, waits and epilogues at different times instead of in lockstep (MI355X_MICROARCH.md, "two waves per
    // SIMD", item 9).  Slice hazards with 4 LDS slices: interval t reads slices t-1 and t and writes slice t+2.
    // ------------------------------------------------------------------------------------------------------------
    constexpr int kLag = (CODD_STAGGER && MODE == MODE_FILTER && kKS == 2 && !RES) ? 1 : 0;
    constexpr bool kNoStage = CODD_EXP_NO_QSTAGE || RES == 1;
    // RES = 2 (launched only for rows of exactly 6 K-steps: 768 int8 elements): slices 0 and 1 stay in LDS slices 0 and 1,
    // slices 2..5 alternate between LDS slices 2 and 3.  The unrolled body is one whole tile (6 steps), so which step
    // stages what is known at compile time (a runtime choice would make hipcc drain the corpus prefetch at every join):
    //   step 3 stages slice 4 -> LDS slice 2, step 4: 5 -> 3, step 5: 2 (of the next tile) -> 2, step 0: 3 -> 3,
    //   steps 1 and 2 stage nothing; barriers behind steps 2, 3, 4 and 5.  4 staged slices per tile instead of 6.
    constexpr int kBody = RES == 2 ? 6 : kUnroll;
    static_assert(RES != 2 || (kRing == 3 && kQD == 1 && kQS == 2), "the 6-step body assumes the default ring and stages");
    constexpr bool kBarrierEveryStep = kLag == 1;
    constexpr int kNqbRun = (CODD_EXP_NB < NBQ ? CODD_EXP_NB : NBQ) * kQBper32 / kQSplit;
    const int TI = T + kLag;  // intervals

    // MFMAs of K-halves [KS0, KS1) of the step whose corpus fragments sit in ring slot SLOT and whose query slice is at qs
    auto mfma_part = [&](auto SLOT, auto KS0, auto KS1, const uint4* qs) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
#pragma unroll
        for (int ks = decltype(KS0)::value; ks < decltype(KS1)::value; ++ks) {
#pragma unroll
            for (int qb = 0; qb < kNqbRun; ++qb) {
                const bf16x8 b = CODD_EXP_NO_LDSREAD ? __builtin_bit_cast(bf16x8, ring[slot][0][(ks + qb) & 3])
                                                     : __builtin_bit_cast(bf16x8, qs[b_piece(qb, ks) * 64]);
#pragma unroll
                for (int rs = 0; rs < kRS; ++rs) {
                    // row block rs lives in 32-row block rs / (32/kMB), sub-block rs % (32/kMB)
                    const bf16x8 a = __builtin_bit_cast(bf16x8, ring[slot][rs / kQBper32][a_piece(rs % kQBper32, ks)]);
                    if constexpr (EL == 1) {
#if CODD_MFMA16
                        acc[rs][qb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), acc[rs][qb], 0, 0, 0);
#endif
                    } else {
#if CODD_MFMA16 && CODD_SHADOW_F16
                    acc[rs][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[rs][qb], 0, 0, 0);
#elif CODD_MFMA16
                    acc[rs][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[rs][qb], 0, 0, 0);
#elif CODD_SHADOW_F16
                    acc[rs][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[rs][qb], 0, 0, 0);
#else
                    acc[rs][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[rs][qb], 0, 0, 0);
#endif
                    }
                }
            }
        }
    };
    // threshold test / bucket maxima / dump of the finished tile cu (run-tile ordinal), then clear the accumulators
    auto tile_epilogue = [&](int64_t cu) __attribute__((always_inline)) {
        if (!CODD_NO_EPILOGUE) {
            const int64_t tile = cu * tile_stride;
            const bool ragged = (tile + 1) * kTileRows > n;
#pragma unroll
            for (int rs = 0; rs < kRS; ++rs) {
                const int64_t row0 = tile * kTileRows + wr * (32 * kRB) + rs * kMB;  // + acc_row(r, lane)
                float rsc[kAccRegs];  // int8: this lane's row scales
#pragma unroll
                for (int r = 0; r < kAccRegs; ++r) rsc[r] = (EL && row0 + acc_row(r, lane) < n) ? rscale[row0 + acc_row(r, lane)] : 0.0f;
                if (MODE == MODE_FILTER) {
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) {
                        const float th = CODD_EXP_NO_HITS ? INFINITY : __uint_as_float(lds_w[qbase + qb * kMB]);
                        float v4[kAccRegs];
#pragma unroll
                        for (int r = 0; r < kAccRegs; ++r) v4[r] = EL ? (float)acc[rs][qb][r] * rsc[r] : (float)acc[rs][qb][r];
                        float m = v4[0];
#pragma unroll
                        for (int r = 1; r < kAccRegs; ++r) m = fmaxf(m, v4[r]);
                        if (__any(m >= th)) {
#if CODD_BALLOT_HITS
                            // one LDS atomic per block that has hits (lane 0 reserves the block's slots, every
                            // hit lane finds its own from the ballots) instead of one contended atomic per register
                            unsigned tot = 0;
#pragma unroll
                            for (int r = 0; r < kAccRegs; ++r)
                                tot += (unsigned)__popcll(__ballot(v4[r] >= th && row0 + acc_row(r, lane) < n));
                            unsigned base = 0;
                            if (lane == 0) base = atomicAdd(&lds_w[256], tot);
                            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                            for (int r = 0; r < kAccRegs; ++r) {
                                const bool hit = v4[r] >= th && row0 + acc_row(r, lane) < n;
                                const unsigned long long mk = __ballot(hit);
                                if (hit) {
                                    const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                                    if (slot < (unsigned)kHitCap) {
                                        lds_hits[slot * 3 + 0] = __float_as_uint(EL ? v4[r] * qscale[qbase + qb * kMB] : v4[r]);
                                        lds_hits[slot * 3 + 1] = (unsigned)(row0 + acc_row(r, lane));
                                        lds_hits[slot * 3 + 2] = (unsigned)(qbase + qb * kMB);
                                    } else {
                                        atomicOr(&hit_cnt[(qbase + qb * kMB) * kHitCntStride], 0x80000000u);  // see below
                                    }
                                }
                                base += (unsigned)__popcll(mk);
                            }
#else
#pragma unroll
                            for (int r = 0; r < kAccRegs; ++r) {
                                const float v = v4[r];
                                const int64_t row = row0 + acc_row(r, lane);
                                if (v >= th && row < n) {
                                    const unsigned slot = atomicAdd(&lds_w[256], 1u);
                                    if (slot < (unsigned)kHitCap) {
                                        lds_hits[slot * 3 + 0] = __float_as_uint(EL ? v * qscale[qbase + qb * kMB] : v);
                                        lds_hits[slot * 3 + 1] = (unsigned)row;
                                        lds_hits[slot * 3 + 2] = (unsigned)(qbase + qb * kMB);
                                    } else {
                                        // workgroup list full (> kHitCap/2 hits inside ONE tile: a dense cluster that many
                                        // queries point at): straight to the query's global list — slow (a returning global
                                        // atomic per hit) but complete, so the query needs no exact-scan fallback.  (Round 1
                                        // poisoned the query's counter here instead and sent it to the fallback: 168 ms per
                                        // batch on a corpus of 64 tight clusters.)
                                        unsigned gq = (unsigned)(qbase + qb * kMB);
                                        asm volatile("" : "+v"(gq));  // (opaque: hipcc otherwise hoists these addresses out of the K loop and spills them)
                                        const unsigned gslot = atomicAdd(&hit_cnt[gq * kHitCntStride], 1u);
                                        if (gslot < (unsigned)cap_q) hits[(int64_t)gq * cap_q + gslot] = make_key(EL ? v * qscale[gq] : v, (uint32_t)row);
                                    }
                                }
                            }
#endif
                        }
                    }
                } else if (MODE == MODE_SAMPLE) {
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) {
                        u64 best = 0ull;  // this lane's best (score, row) of the block; ties -> lower row, as everywhere
#pragma unroll
                        for (int r = 0; r < kAccRegs; ++r) {
                            const int64_t row = row0 + acc_row(r, lane);
                            if (!ragged || row < n) {
                                const u64 key = make_key(EL ? (float)acc[rs][qb][r] * rsc[r] : (float)acc[rs][qb][r], (uint32_t)row);
                                best = key > best ? key : best;
                            }
                        }
                        atomicMax(reinterpret_cast<unsigned long long*>(&lds_k[qbase + qb * kMB]), (unsigned long long)best);
                    }
                } else {
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                        for (int r = 0; r < kAccRegs; ++r) {
                            const int64_t row = row0 + acc_row(r, lane);
                            if (row < n)
                                dump[(int64_t)(qbase + qb * kMB) * n + row] =
                                    EL ? (float)acc[rs][qb][r] * rsc[r] * qscale[qbase + qb * kMB] : (float)acc[rs][qb][r];
                        }
                }
            }  // rs
        }
#pragma unroll
        for (int rs = 0; rs < kRS; ++rs)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int r = 0; r < kAccRegs; ++r) acc[rs][qb][r] = 0;
    };
    auto slice_ptr = [&](int step) -> const uint4* {  // LDS address of the query slice of a step (4 slices, cyclic)
        return ldsQ + ((step + 4) & (2 * kQS - 1)) * kStagePieces + lane + qoff;
    };
    static_assert(kQS == 2 || !kLag, "the staggered schedule is written for 4 LDS slices");

    int64_t c_u = first_u;  // compute cursor: the step a wave finishes in this interval (a lagging wave: step t-1)
    int c_s = 0;
    int64_t w_u = first_u;  // workgroup cursor: the step ALL waves have finished by the end of this interval
    int w_s = 0;
    // TI is walked in whole unrolled bodies: the padding intervals past it recompute the last slice into accumulators
    // nobody reads (`live` gates every side effect), which keeps the loop free of early exits and lets every load stay
    // unconditional
    auto run = [&](auto LAG) __attribute__((always_inline)) {
        constexpr int lag = decltype(LAG)::value;
        for (int t0 = 0; t0 < TI; t0 += kBody) {
            // one interval; IU (position inside the unrolled body) is a compile-time constant, so every register
            // buffer (corpus ring slot, query staging buffer) is chosen by the front end, not by an optimisation pass
            auto k_step = [&](auto IU) __attribute__((always_inline)) {
                constexpr int iu = decltype(IU)::value;
                constexpr int i = iu % kRing;
                const int t = t0 + iu;
                const int stage = t / kQS, sub = t % kQS;
                // RES = 2: iu IS the step inside the tile; what this step stages (slice, LDS slice) or nothing
                constexpr int kStageSlice = iu == 3 ? 4 : (iu == 4 ? 5 : (iu == 5 ? 2 : (iu == 0 ? 3 : -1)));
                constexpr int kStageSlot = iu == 3 ? 2 : (iu == 4 ? 3 : (iu == 5 ? 2 : 3));
                constexpr bool kStagesHere = RES == 2 ? kStageSlice >= 0 : !kNoStage;
                // query slice of step t+kQS: issue now, write to LDS at the end of the interval
                {
                    const uint4* src = qfrag + (int64_t)(RES == 2 ? (kStageSlice < 0 ? 0 : kStageSlice) : q_s) * kStagePieces + stid;
                    if (kStagesHere) {
#pragma unroll
                        for (int j = 0; j < kQPn; ++j) (iu % kQD ? qreg1 : qreg0)[j] = reinterpret_cast<const q4u*>(src)[j * kFilterThreads];
                    }
                    q_s = q_s + 1 == nsteps ? 0 : q_s + 1;
                }
                if constexpr (lag == 0) {
                    const bool live = t < T;
                    // corpus fragments for step t+2 AFTER the query loads: vmcnt retires in order, so the
                    // end-of-step wait for the (L2-served) query slice must not sit behind these HBM loads
                    load_a(ring[(i + kPrefetch) % kRing]);
#if CODD_MFMA_PRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
                    constexpr int kReadSlot6 = iu < 4 ? iu : iu - 2;  // RES = 2: slices 0..5 sit in LDS slices 0,1,2,3,2,3
                    mfma_part(std::integral_constant<int, i>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, kKS>{},
                              RES == 1 ? ldsQ + c_s * kStagePieces + lane + qoff
                                       : (RES == 2 ? ldsQ + kReadSlot6 * kStagePieces + lane + qoff : slice_ptr(t)));
#if CODD_MFMA_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
#if CODD_PIN_SCHEDULE && !CODD_EXP_NO_LDSREAD
                    // pin the step's shape: all 8 global loads (4 query, 4 corpus) first so they fly under the
                    // MFMAs; at most a few query fragments live (4 LDS reads up front, then one per MFMA);
                    // the LDS writes of the next query slice last
                    __builtin_amdgcn_sched_group_barrier(0x020, 4 * kRB + (kStagesHere ? kQPn : 0), 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int g = 0; g < kKS * kNqbRun - 4; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, kRS, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kRS, 0);
                    if (kStagesHere) __builtin_amdgcn_sched_group_barrier(0x200, kQPn, 0);
#endif
                    {
                        uint4* dstq = ldsQ + (RES == 2 ? kStageSlot : ((stage + 1) & 1) * kQS + sub) * kStagePieces + stid;
                        if (kStagesHere) {
#pragma unroll
                            for (int j = 0; j < kQPn; ++j)
                                reinterpret_cast<q4u*>(dstq)[j * kFilterThreads] = ((iu + kQD - 1) % kQD ? qreg1 : qreg0)[j];
                        }
                    }
                    if (live && c_s == nsteps - 1) tile_epilogue(c_u);
                    if (++c_s == nsteps) { c_s = 0; c_u += step_u; }
                } else {
                    // ---- lagging wave: second K-half of step t-1, epilogue, corpus loads, first K-half of step t ----
                    const bool live_prev = t >= 1 && t - 1 < T;
                    mfma_part(std::integral_constant<int, (i + kRing - 1) % kRing>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{},
                              slice_ptr(t - 1));
#if CODD_PIN_SCHEDULE && !CODD_EXP_NO_LDSREAD
                    __builtin_amdgcn_sched_group_barrier(0x020, (kNoStage ? 0 : kQPn), 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int g = 0; g < kNqbRun - 4; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, kRS, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kRS, 0);
#endif
                    if (t >= 1) {
                        if (live_prev && c_s == nsteps - 1) tile_epilogue(c_u);
                        if (++c_s == nsteps) { c_s = 0; c_u += step_u; }
                    } else {
                        // interval 0 multiplied nothing meaningful (there is no step -1): start from clean accumulators
#pragma unroll
                        for (int rs = 0; rs < kRS; ++rs)
#pragma unroll
                            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                                for (int r = 0; r < kAccRegs; ++r) acc[rs][qb][r] = 0;
                    }
                    load_a(ring[(i + kPrefetch) % kRing]);  // the slot the second half of step t-1 has just released
                    mfma_part(std::integral_constant<int, i>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, slice_ptr(t));
#if CODD_PIN_SCHEDULE && !CODD_EXP_NO_LDSREAD
                    __builtin_amdgcn_sched_group_barrier(0x020, 4 * kRB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int g = 0; g < kNqbRun - 4; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, kRS, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kRS, 0);
                    if (!kNoStage) __builtin_amdgcn_sched_group_barrier(0x200, kQPn, 0);
#endif
                    {
                        uint4* dstq = ldsQ + (((stage + 1) & 1) * kQS + sub) * kStagePieces + stid;
                        if (!kNoStage) {
#pragma unroll
                            for (int j = 0; j < kQPn; ++j)
                                reinterpret_cast<q4u*>(dstq)[j * kFilterThreads] = ((iu + kQD - 1) % kQD ? qreg1 : qreg0)[j];
                        }
                    }
                }

                // ---- workgroup bookkeeping (identical in both programs: the barriers must pair up) ----
                // the step every wave has finished by the end of this interval is t - kLag
                const bool wg_tile_end = t >= kLag && t - kLag < T && w_s == nsteps - 1;
                // stage boundary: the other stage is complete and this one is free to be overwritten.
                // A tile end synchronises too (its bookkeeping below needs every wave's epilogue done).
                if (kBarrierEveryStep || (sub == kQS - 1 && !CODD_EXP_NO_BARRIER && !RES) || (RES == 2 && iu >= 2) || wg_tile_end) __syncthreads();
                if (MODE == MODE_FILTER && wg_tile_end) {
                    // empty the workgroup's hit list once it is half full
                    const unsigned cnt = lds_w[256];
                    __syncthreads();  // everyone has read cnt before the next epilogue can move it
                    if (cnt > (unsigned)(CODD_FLUSH_AT)) {
#if CODD_BINNED_INLOOP
                        // (scratch: the 256 free words behind the thresholds and the counter)
                        flush_hits_binned(lds_hits, cnt < (unsigned)kHitCap ? cnt : (unsigned)kHitCap, tid, lds_w + 320, hits, hit_cnt, cap_q);
#else
                        flush_hits(lds_hits, cnt < (unsigned)kHitCap ? cnt : (unsigned)kHitCap, tid, hits, hit_cnt, cap_q);
#endif
                        __syncthreads();
                        if (tid == 0) lds_w[256] = 0u;
                        __syncthreads();
                    }
                }
                if (MODE == MODE_SAMPLE && wg_tile_end) {
                    // all 8 waves have folded this tile into lds_w: publish, reset, and fence the reset
                    // against the next tile's fold
                    if (tid < 256) {
                        // [query][bucket]: the threshold kernel reads rows of it.  (int8: the fold ran on acc * rscale; a
                        // scale >= 0 keeps the order, the query's scale is applied here)
                        u64 key = lds_k[tid];
                        if (EL && key) key = make_key(key_score(key) * qscale[tid], key_row(key));
                        bucket_key[(int64_t)tid * ntiles_run + w_u] = key;
                        lds_k[tid] = 0ull;
                    }
                    __syncthreads();
                }
                if (t >= kLag && ++w_s == nsteps) { w_s = 0; w_u += step_u; }
            };
            static_for<kBody>(k_step);
        }
    };
    if (kLag && __builtin_amdgcn_readfirstlane(wave) >= kFilterWaves / 2) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});

    if (MODE == MODE_FILTER) {
        __syncthreads();
        const unsigned cnt = lds_w[256];
        if (cnt > (unsigned)kHitCap && tid == 0) atomicOr(&flags[FLAG_WG_OVERFLOW], 1u);  // statistics only
        __syncthreads();  // everyone has read cnt (lds_w is about to be reused)
        if (CODD_EXP_NO_FLUSH) return;
        if (CODD_BINNED_FLUSH) flush_hits_binned(lds_hits, cnt < (unsigned)kHitCap ? cnt : (unsigned)kHitCap, tid, lds_w, hits, hit_cnt, cap_q);
        else flush_hits(lds_hits, cnt < (unsigned)kHitCap ? cnt : (unsigned)kHitCap, tid, hits, hit_cnt, cap_q);
    }
}

constexpr int kSurvChunk = 2048;  // hits examined per round; their survivors always fit the LDS list
#ifndef CODD_FIN_WAVES
#define CODD_FIN_WAVES 8
#endif
constexpr int kFinWaves = CODD_FIN_WAVES;      // waves of a finalize workgroup (one query, or one share of its hits): the kernel is a chain of
constexpr int kFinThreads = kFinWaves * kWave; // round trips (hit list, k anchor rows, survivor rows), so a query gets as many waves as pay

// finalize_body: what one workgroup of kFinThreads threads does for ONE query (or one share of its candidates): `my[0, total)`
// are the query's candidate keys (approximate score, local row; 0 = empty slot), qn_q its normalised fp32 vector.
//   eps1 : the slack between an approximate and an exact score — the whole of it, or (bmeta != nullptr, the int8 filter) its
//          block-independent part A(q), with bq = B(q): eps(q, block) = A + B e_block, e_block = bmeta[row / 32].y
//   out_*_q / part_keys_q : where this query's k results go (any may be null)
// Shared by finalize_kernel and by the last workgroup of small_batch_kernel (codd_knn.hip).
template <int DT, int NITER, int SLOTS>
__device__ __forceinline__ void finalize_body(const void* __restrict__ rows_, int dpad, const float* __restrict__ qn_q, const u64* __restrict__ my,
                                              unsigned total, unsigned part, unsigned nparts, int k, float eps1, float bq,
                                              const float2* __restrict__ bmeta, uint32_t row_base, u64* __restrict__ out_keys_q,
                                              float* __restrict__ out_dist_q, int64_t* __restrict__ out_rows_q, u64* __restrict__ part_keys_q,
                                              unsigned long long* __restrict__ stats, float* lo_out = nullptr) {
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    __shared__ u64 lds_list[kFinWaves * SLOTS * kWave];
    __shared__ unsigned lds_surv[kSurvChunk];
    __shared__ unsigned lds_n;
    __shared__ float lds_lo;
    __shared__ float lds_anchor[kFinWaves];

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // the query's fragments for the exact re-scoring: requested first, so the round trip overlaps step 1
    const int nchunks = dpad / E;
    float qf[NITER][E];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int j = lane + kWave * it;
#pragma unroll
        for (int e = 0; e < E; ++e) qf[it][e] = j < nchunks ? qn_q[(int64_t)j * E + e] : 0.0f;
    }

    // 1. k-th largest approximate key
    WaveTopK<SLOTS> L;
    L.init();
    for (unsigned i0 = wave * kWave; i0 < total; i0 += kFinThreads) {
        const unsigned i = i0 + lane;
        L.offer_lanes(i < total ? my[i] : 0ull, k, lane);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) lds_list[(wave * SLOTS + s) * kWave + lane] = L.v[s];
    __syncthreads();
    if (wave == 0) {
        for (int wv = 1; wv < kFinWaves; ++wv)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                u64 cand = lds_list[(wv * SLOTS + s) * kWave + lane];
                if (s * kWave + lane >= k) cand = 0ull;
                L.offer_lanes(cand, k, lane);
            }
        // rows of the k best approximate hits (or none when the list is not full: then every hit survives)
        if (lane == 0) lds_n = L.thr ? (unsigned)k : 0u;
        if (L.thr) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s * kWave + lane < k) lds_surv[s * kWave + lane] = key_row(L.v[s]);
        }
    }
    __syncthreads();
    const uint4* base = reinterpret_cast<const uint4*>(rows_);
    // canonical exact scores of up to four rows per wave step (the expression of the exact scan)
    auto rescore4 = [&](const unsigned (&rowid)[4], float (&sc)[4]) __attribute__((always_inline)) {
        float w[4][NITER][E];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint4* p = base + (int64_t)rowid[r] * nchunks + lane;
#pragma unroll
            for (int it = 0; it < NITER; ++it) {
                uint4 cch = make_uint4(0u, 0u, 0u, 0u);
                if (lane + kWave * it < nchunks) cch = p[kWave * it];
                RT::widen(cch, w[r][it]);
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
        for (int r = 0; r < 4; ++r) sc[r] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), 16 * r));
    };
    // 1b. anchor: those k rows are k distinct rows with EXACT scores >= L' := their smallest exact score, so the true
    // k-th best score is >= L' and every true top-k row has an approximate score >= L' - eps (one eps, not two: the
    // bound compares an approximation with an exact score, not two approximations)
    {
        const unsigned nk = lds_n;
        float worst = INFINITY;
        for (unsigned j = wave * 4; j < nk; j += 4 * kFinWaves) {
            unsigned rowid[4];
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rowid[r] = lds_surv[j + r < nk ? j + r : nk - 1];
            rescore4(rowid, sc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j + r < nk) worst = fminf(worst, sc[r]);
        }
        if (lane == 0) lds_anchor[wave] = worst;
        __syncthreads();
        if (tid == 0) {
            float l4 = lds_anchor[0];
#pragma unroll
            for (int wv = 1; wv < kFinWaves; ++wv) l4 = fminf(l4, lds_anchor[wv]);
            lds_lo = nk ? l4 - eps1 : -INFINITY;
            if (lo_out) *lo_out = lds_lo;
        }
        __syncthreads();
    }
    const float lo = lds_lo;

    WaveTopK<SLOTS> X;
    X.init();
    unsigned survivors = 0;

    // 2 + 3. round by round: survivors of the next kSurvChunk hits (approx >= a_k - 2 eps: no other row can reach
    // the exact top-k) are compacted into LDS and re-scored exactly, four rows per wave step, with the same
    // canonical expression as the scan.  No cap on the number of survivors: a dense cluster costs time, not exactness.
    const unsigned my_lo = (unsigned)((u64)total * part / nparts), my_hi = (unsigned)((u64)total * (part + 1) / nparts);  // this workgroup's share
    for (unsigned b0 = my_lo; b0 < my_hi; b0 += kSurvChunk) {
        if (tid == 0) lds_n = 0u;
        __syncthreads();
        const unsigned b1 = b0 + kSurvChunk < my_hi ? b0 + kSurvChunk : my_hi;
        for (unsigned i = b0 + tid; i < b1; i += kFinThreads) {
            const u64 key = my[i];
            const float slack_b = bmeta ? bq * bmeta[key_row(key) >> 5].y : 0.0f;
            if (key_score(key) + slack_b >= lo) lds_surv[atomicAdd(&lds_n, 1u)] = key_row(key);
        }
        __syncthreads();
        const unsigned ns = lds_n;
        survivors += ns;
        // eight rows per wave step: two independent groups of four, so that both groups' row reads are in flight
        // before the first is consumed (the loop is a chain of HBM round trips otherwise)
        for (unsigned j0 = wave * 8; j0 < ns; j0 += 8 * kFinWaves) {
            unsigned rowid8[2][4];
            float sc8[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) rowid8[h][r] = lds_surv[j0 + 4 * h + r < ns ? j0 + 4 * h + r : ns - 1];
            rescore4(rowid8[0], sc8[0]);
            rescore4(rowid8[1], sc8[1]);
#pragma unroll
            for (int hr = 0; hr < 8; ++hr) {
                const int r = hr & 3;
                const unsigned j = j0 + 4 * (hr >> 2);
                const unsigned (&rowid)[4] = rowid8[hr >> 2];
                const float sc = sc8[hr >> 2][r];
                if (j + r < ns) X.offer(make_key(sc, row_base + rowid[r]), k, lane);
            }
        }
        __syncthreads();  // lds_surv is refilled by the next round
    }
    if (tid == 0 && stats) {
        if (part == 0) atomicAdd(&stats[0], (unsigned long long)total);
        atomicAdd(&stats[1], (unsigned long long)survivors);
    }

    // 4. merge the waves' lists
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) lds_list[(wave * SLOTS + s) * kWave + lane] = X.v[s];
    __syncthreads();
    if (wave == 0) {
        for (int wv = 1; wv < kFinWaves; ++wv)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                u64 cand = lds_list[(wv * SLOTS + s) * kWave + lane];
                if (s * kWave + lane >= k) cand = 0ull;
                X.offer_lanes(cand, k, lane);
            }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int rank = s * kWave + lane;
            if (rank < k) {
                const u64 key = X.v[s];
                if (nparts > 1) {
                    part_keys_q[rank] = key;
                } else {
                    // the caller's (distance, row) outputs straight from here: no unpack launch behind the pass
                    if (out_keys_q) out_keys_q[rank] = key;
                    if (out_dist_q) out_dist_q[rank] = key ? 1.0f - key_score(key) : INFINITY;
                    if (out_rows_q) out_rows_q[rank] = key ? (int64_t)key_row(key) : (int64_t)-1;
                }
            }
        }
    }
    __syncthreads();  // (a caller that loops over queries reuses the shared lists)
}

template <int DT, int NITER, int SLOTS>
__global__ __launch_bounds__(kFinThreads) void finalize_kernel(const void* __restrict__ rows_, int dpad, const float* __restrict__ qn,
                                                       const u64* __restrict__ hits, const unsigned* __restrict__ hit_cnt,
                                                       int cap_q, unsigned* __restrict__ flags, int k, float two_eps,
                                                       uint32_t row_base, u64* __restrict__ out_keys,
                                                       unsigned* __restrict__ fb_count, unsigned* __restrict__ fb_list,
                                                       unsigned long long* __restrict__ stats, const float* __restrict__ two_eps_q = nullptr,
                                                       u64* __restrict__ part_keys = nullptr, float* __restrict__ out_dist = nullptr,
                                                       int64_t* __restrict__ out_rows = nullptr, const float2* __restrict__ bmeta = nullptr) {
    // bmeta (int8 filter; two_eps_q = qmeta + 256): the slack is evaluated per 32-row block, eps(q, block) = A(q) + B(q) e_block
    // (A at two_eps_q[256 + q], B at two_eps_q[512 + q], e_block = bmeta[row / 32].y): a hit survives iff
    // approx + B e_block >= L' - A.  Valid whichever kernel wrote the hit list (every kernel's list holds these rows).
    // gridDim.y > 1: the query's hit list is shared out between gridDim.y workgroups (contiguous shares), each
    // writes its own top-k to part_keys[(q * P + p) * k ..] and a merge launch follows.  With one or a few queries
    // and thousands of survivors (the int8 filter) one workgroup per query would do all the re-scoring on one CU.
    const int q = blockIdx.x;
    const unsigned total = hit_cnt[q * kHitCntStride];
    const unsigned part = blockIdx.y, nparts = gridDim.y;
    if (total > (unsigned)cap_q) {  // the candidate list was truncated: only the exact scan can answer this query
        if (threadIdx.x == 0 && part == 0) {
            fb_list[atomicAdd(fb_count, 1u)] = (unsigned)q;
            atomicOr(&flags[FLAG_NEED_FALLBACK], 1u);
        }
        return;
    }
    const float eps1 = bmeta ? two_eps_q[256 + q] : 0.5f * (two_eps_q ? two_eps_q[q] : two_eps);
    const float bq = bmeta ? two_eps_q[512 + q] : 0.0f;
    finalize_body<DT, NITER, SLOTS>(rows_, dpad, qn + (int64_t)q * dpad, hits + (int64_t)q * cap_q, total, part, nparts, k, eps1, bq, bmeta, row_base,
                                    out_keys ? out_keys + (int64_t)q * k : nullptr, out_dist ? out_dist + (int64_t)q * k : nullptr,
                                    out_rows ? out_rows + (int64_t)q * k : nullptr, part_keys ? part_keys + ((int64_t)q * nparts + part) * k : nullptr, stats);
}

// ---------------------------------------------------------------------------------------------
// anchor_thr_kernel: one wave per query.  The sample pass left, per bucket (sampled tile), the (approximate score, row)
// key of its best row.  The k best buckets name k DISTINCT rows; their EXACT scores (canonical expression) are all
// >= L := the smallest of them, hence the true k-th best score is >= L and every true top-k row has an approximate
// score >= L - eps.  thr[q] = L - eps(q); -inf when fewer than k buckets hold a row; +inf for padding queries.
// (Anchoring on exact scores costs k row reads per query and saves one eps of slack against the k-th largest
// *approximate* bucket maximum: a third fewer hits for the bf16 filter, 6x fewer for the int8 one.)
// ---------------------------------------------------------------------------------------------
constexpr int kAnchorWaves = 4;  // waves per query: the bucket scan and the k row reads are shared out (one wave: 19 us of a 430 us shard step)
template <int DT, int NITER, int SLOTS>
__global__ __launch_bounds__(kAnchorWaves * kWave) void anchor_thr_kernel(const u64* __restrict__ bucket_key, int64_t nbuckets, int B, int k,
                                                                          const void* __restrict__ rows_, int dpad, const float* __restrict__ qn, float eps,
                                                                          const float* __restrict__ two_eps_q, float* __restrict__ thr,
                                                                          float* __restrict__ thr0 = nullptr) {
    // thr0 (int8 filter, with two_eps_q = qmeta + 256): L - A(q), the block-independent part of the per-block threshold
    // L - A(q) - B(q) e_block that i8_tile_kernel evaluates (A at two_eps_q[256 + q])
    typedef RowTraits<DT> RT;
    constexpr int E = RT::E;
    __shared__ u64 lds_list[kAnchorWaves * SLOTS * kWave];  // the waves' lists; then [rank] of the merged one
    __shared__ float lds_worst[kAnchorWaves];
    __shared__ int lds_full;
    const int q = blockIdx.x, tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    if (q >= B) {
        if (tid == 0) {
            thr[q] = INFINITY;
            if (thr0) thr0[q] = INFINITY;
        }
        return;
    }
    const int nchunks = dpad / E;
    float qf[NITER][E];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int j = lane + kWave * it;
#pragma unroll
        for (int e = 0; e < E; ++e) qf[it][e] = j < nchunks ? qn[(int64_t)q * dpad + (int64_t)j * E + e] : 0.0f;
    }
    WaveTopK<SLOTS> L;
    L.init();
    for (int64_t i0 = (int64_t)wave * kWave; i0 < nbuckets; i0 += kAnchorWaves * kWave) {
        const int64_t i = i0 + lane;
        L.offer_lanes(i < nbuckets ? bucket_key[(int64_t)q * nbuckets + i] : 0ull, k, lane);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) lds_list[(wave * SLOTS + s) * kWave + lane] = L.v[s];
    __syncthreads();
    if (wave == 0) {
        for (int wv = 1; wv < kAnchorWaves; ++wv)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                u64 cand = lds_list[(wv * SLOTS + s) * kWave + lane];
                if (s * kWave + lane >= k) cand = 0ull;
                L.offer_lanes(cand, k, lane);
            }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) lds_list[s * kWave + lane] = L.v[s];  // rank s * 64 + lane
        if (lane == 0) lds_full = L.thr ? 1 : 0;
    }
    __syncthreads();
    if (!lds_full) {  // fewer than k buckets with a row: no threshold can be justified
        if (tid == 0) {
            thr[q] = -INFINITY;
            if (thr0) thr0[q] = -INFINITY;
        }
        return;
    }
    const uint4* base = reinterpret_cast<const uint4*>(rows_);
    // exact scores of the rows ranked [j, j+4) (ranks past k-1 repeat the last one)
    auto group = [&](int j, float (&sc)[4]) __attribute__((always_inline)) {
        float w[4][NITER][E];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rank = j + r < k ? j + r : k - 1;  // uniform
            const u64 key = lds_list[rank];
            const uint4* p = base + (int64_t)key_row(key) * nchunks + lane;
#pragma unroll
            for (int it = 0; it < NITER; ++it) {
                uint4 cch = make_uint4(0u, 0u, 0u, 0u);
                if (lane + kWave * it < nchunks) cch = p[kWave * it];
                RT::widen(cch, w[r][it]);
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
        for (int r = 0; r < 4; ++r) sc[r] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), 16 * r));
    };
    float worst = INFINITY;
    if (k <= 4 * kAnchorWaves) {  // the usual case (k = 10): one group of four rows per wave, one round trip
        if (4 * wave < k) {
            float s0[4];
            group(4 * wave, s0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * wave + r < k) worst = fminf(worst, s0[r]);
        }
    } else {
        for (int j = 8 * wave; j < k; j += 8 * kAnchorWaves) {  // two independent groups per step: both groups' row reads fly together
            float s0[4], s1[4];
            group(j, s0);
            group(j + 4, s1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (j + r < k) worst = fminf(worst, s0[r]);
                if (j + 4 + r < k) worst = fminf(worst, s1[r]);
            }
        }
    }
    if (lane == 0) lds_worst[wave] = worst;
    __syncthreads();
    if (tid == 0) {
        float wmin = lds_worst[0];
#pragma unroll
        for (int wv = 1; wv < kAnchorWaves; ++wv) wmin = fminf(wmin, lds_worst[wv]);
        thr[q] = wmin - (two_eps_q ? 0.5f * two_eps_q[q] : eps);
        if (thr0) thr0[q] = wmin - two_eps_q[256 + q];
    }
}

}  // namespace codd
