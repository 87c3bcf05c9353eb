// filter_i8.h — the int8 filter GEMM for batches of 65..256 queries on rows of 384 elements or more (3 K-steps),
// second generation.  Same operands, same layouts, same hit lists as gemm_filter_kernel<MODE, 8, EL = 1> in
// filter_gemm.h (which stays the kernel of every other shape); what changed is the schedule:
//
//   * round 1's loop compiled to `ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 2 MFMA` 32 times per K-step with ONE query
//     fragment register set: every MFMA pair paid a full LDS round trip (profiles/r2/i8_v1_isa_excerpt.txt; matrix pipe
//     38 % busy, LDS array 26 % busy, waves parked 56 % of their cycles).  Here the query fragments run through a ring
//     of kBD register sets and the order of a K-step is pinned;
//   * the query slices go L2 -> LDS by LDS-DMA (buffer_load ... lds), not through 16 staging registers and ds_write: with
//     128 accumulator and 48 corpus-ring registers the staging set pushed the wave over the 256 registers it has at two
//     waves per SIMD (hipcc spilled, and every scratch reload waits vmcnt(0));
//   * EVERY vector-memory operation of the loop is issued by inline asm and waited for by hand with counted vmcnt
//     (the queue retires in order; per interval it receives, in this order: 1 scale load, 4 DMA, 4 corpus loads).
//     hipcc sees no load in flight, so it inserts no wait of its own: with the DMA visible it ordered every LDS access
//     it could see behind the youngest DMA, and a false register dependency on an in-flight load cost a vmcnt(0)
//     inside the epilogue (measured: 5 us per tile).  The fragment reads of the MFMA phase are inline asm too, with
//     counted lgkmcnt; values flow from each wait asm to their consumers as in/out operands;
//   * waves 4..7 (the SIMD partners of waves 0..3) can run one K-step behind (CODD_I8_LAG), and a tile's epilogue is
//     deferred to the start of the wave's next interval;
//   * the epilogue tests a block pair on its largest accumulator first — max of the lane's 8 accumulators of (2 row blocks x
//     1 query block) times the block's scale against the query's threshold: 9 vector instructions per pair instead of 26.
//     The int8 shadow carries ONE scale per 32-row block (= per wave tile), so that pre-test is exact at pair level and only
//     pairs that hold a hit (about one in eight) take the per-value test — same expression as the first-generation kernel,
//     so the hit lists are the same — which runs straight-line (select + count, one append) unless a lane holds two hits;
//   * the sample mode folds the wave's 32 rows as packed (accumulator << 5 | 31 - row) ints: integer max, two lane swaps,
//     one (score, row) key per query and wave;
//   * row scales reach the epilogue through LDS (one small DMA per wave and interval), not through global loads inside a
//     conditional region;
//   * the workgroup's hit-list bookkeeping rides on the interval barrier: no extra barriers per tile;
//   * instantiations: NQB = 16 / 8 query blocks (129..256 / 65..128 queries); RES: rows of <= 4 K-steps keep the whole query
//     block in the four LDS slices (staged once per workgroup, no slice DMA per interval) and synchronise once per tile
//     instead of once per K-step (hit lists and sample keys double-buffered by tile parity).
//
// LDS: 4 query slices x 32 KiB | 1088 bookkeeping words | 2 row-scale buffers x (256 + 16) floats | hit list.
#pragma once
#include "filter_gemm.h"

#if !CODD_EXPERIMENTS && (defined(CODD_I8_EXP_NOEPI) || defined(CODD_I8_EXP_NODMA) || defined(CODD_I8_EXP_SAMETILE) || defined(CODD_I8_EXP_NOBARRIER) || \
                          defined(CODD_I8_EXP_NOBREAD) || defined(CODD_I8_EXP_NOHITS) || defined(CODD_I8_EXP_NOAPPEND) || defined(CODD_I8_EXP_NOFLUSH) || defined(CODD_I8_EXP_NOGLOBAL) || \
                          defined(CODD_I8_EXP_STAMPS))
#error "the CODD_I8_EXP_* switches return wrong results or race: they exist only in -DCODD_EXPERIMENTS=1 builds (build_variant)"
#endif

namespace codd {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kI8SliceBytes = 32768;  // one 128-wide K slice of the 256-query block
constexpr int kI8RsBufs = 2;          // row-scale buffers (tile ordinal & 1): a tile's scales are read (its epilogue, at the start of the next tile) before the tile after next requests its own
constexpr int kI8RsStride = 272;      // floats per buffer: 256 row scales + 8 per-wave maxima (+ pad)
#if !CODD_EXPERIMENTS && (defined(CODD_I8_BDEPTH) || defined(CODD_I8_EARLY_FRAGS) || defined(CODD_I8_LAG) || defined(CODD_I8_SPREAD_VM) || defined(CODD_I8_FUSE_EPI) || \
                          defined(CODD_I8_EARLY_A))
#error "CODD_I8_BDEPTH / CODD_I8_EARLY_FRAGS / CODD_I8_LAG / CODD_I8_EARLY_A ... are schedule experiments: only their defaults are under test; -DCODD_EXPERIMENTS=1 builds (build_variant) may set them"
#endif
#ifndef CODD_I8_BDEPTH
#define CODD_I8_BDEPTH 4              // query-fragment register sets in flight
#endif
constexpr int kBD = CODD_I8_BDEPTH;
#ifndef CODD_I8_EARLY_FRAGS
#define CODD_I8_EARLY_FRAGS 1         // the first fragment reads of a K-step go out before the interval's DMA / corpus-load instructions (0: behind them)
#endif
#ifndef CODD_I8_LAG
#define CODD_I8_LAG 0                 // 1: waves 4..7 run one K-step behind waves 0..3 (measured 12 % slower twice, profiles/r2/i8_tile_ablation.txt; experiment builds only)
#endif
#ifndef CODD_I8_SPREAD_VM
#define CODD_I8_SPREAD_VM 0           // pair program: 1 = the interval's vector-memory operations are issued one at a time behind MFMA groups instead of in a
                                      // block in front of them; 2 = ... the SIMD partners taking turns (waves 0..3 behind even groups, 4..7 behind odd ones)
#endif
#ifndef CODD_I8_EARLY_A
#define CODD_I8_EARLY_A 3             // static six-step program, bit mask: the corpus loads of interval 2 (bit 0) / 4 (bit 1) go out in the tail of the odd interval
                                      // in front of it, BEFORE the barrier (0: in the interval's own head, behind the barrier, with everything else)
#endif
#ifndef CODD_I8_FUSE_EPI
#define CODD_I8_FUSE_EPI 0            // 1: the tile-structured filter program tests tile i's accumulators INSIDE the first K-step of tile i + 1 (epi_pair in front
                                      // of the MFMAs that restart the pair).  Built, bit-equal, measured 1.7-3 % SLOWER (profiles/r3/i8_tile_ablation.txt): the
                                      // pre-tests' vector instructions slow the MFMA stream they sit in by as much as the standalone block costs
#endif

// ---- hand-issued memory operations (see the header) --------------------------------------------------------------
// HAZARD (found in round 3 as non-deterministic lost neighbours): gfx9 needs 5 wait states between a VALU instruction that
// WRITES an SGPR (v_readfirstlane, and v_readlane — which is how hipcc reloads the SGPRs this kernel spills to VGPR lanes,
// 2,400 times in the program) and a vector-memory instruction that READS that SGPR (descriptor, soffset).  hipcc pads its
// own instructions for this; it does not look inside inline asm.  A descriptor word produced right in front of one of
// these statements was read STALE: a K-step's corpus fragment came from a wrong address, non-deterministically, and whether
// it happened depended on register pressure (it appeared with one more live SGPR).  Every statement below therefore
// carries its own wait states in front of the memory instruction (the s_mov to m0 counts as one).
// buffer descriptor (4 SGPRs): base, 48-bit address | stride 0, bytes, gfx950 raw-buffer flags; out-of-range reads return 0
__device__ __forceinline__ i32x4 i8_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;  // (readfirstlane: the asm operands are SGPR tuples whatever hipcc's uniformity analysis concludes)
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
template <int OFF>
__device__ __forceinline__ void i8_load_b128_nt(u32x4& dst, int voff, i32x4 rsrc) {  // read-once stream: non-temporal
    // (s_nop 4: the HAZARD note above)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3 nt" : "=v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF));
}
// (m0 is on the clobber lists: hipcc reserves it for its own implicit uses — each is preceded by its own s_mov m0 — and warns that a
// reserved register "may not be preserved"; listing it is what keeps its scheduler from placing these statements between such a pair)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 64 lanes x 16 bytes: global (rsrc + voff + soff) -> LDS [lds_addr + 16 * lane]
__device__ __forceinline__ void i8_dma_b128(unsigned lds_addr, int voff, i32x4 rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rsrc),
                 "s"(__builtin_amdgcn_readfirstlane(soff))
                 : "memory", "m0");
}
// the same for operands that ARE scalar registers already (the static tile program: a v_readfirstlane hipcc does not fold costs a
// vector register for its source, and that one — loop-invariant — was spilled to scratch and reloaded with a vmcnt(0) per tile)
__device__ __forceinline__ void i8_dma_b128_s(unsigned lds_addr, int voff, i32x4 rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// 64 lanes x 4 bytes: global (rsrc + voff + soff) -> LDS [lds_addr + 4 * lane]
__device__ __forceinline__ void i8_dma_b32(unsigned lds_addr, int voff, i32x4 rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rsrc),
                 "s"(__builtin_amdgcn_readfirstlane(soff))
                 : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void i8_wait_vm(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128_asm(i32x4& dst, unsigned addr) {
#ifdef CODD_I8_EXP_NOBREAD
    asm volatile("; no read %0 %1 %2" : "=v"(dst) : "v"(addr), "n"(OFF));  // diagnostic: no LDS read at all
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}
template <int N>
__device__ __forceinline__ void lgkm_wait_asm(i32x4& v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

__device__ __forceinline__ float i8_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// vector-memory operations per interval, in issue order: scale DMA, kDmaPerIv slice DMA, kAPerIv corpus loads
// (kDmaPerIv = NQB / 4: a slice of 16 NQB queries is NQB / 2 chunks of 1 KiB per wave pair... 2 NQB KiB in all, 8 waves)
constexpr int kAPerIv = 4;
// bookkeeping words behind the slices: [0..127] pre-test thresholds, two bf16 per word (layout: the kernel's set-up); [256], [257] hit counts;
// [260..263] the waves' maxima of u_q; [320..575] query scales; [576..831] exact thresholds (thr0 / qscale), by query; [832..1087] scratch of the in-loop flush.
// SAMPLE: [0..511] = 256 u64 keys.
constexpr int kI8Words = 1344;   // ... [1088..1343] u_q = B(q) / qscale_q (FILTER)

// does the resident program apply? (nbq: 32-query blocks of the batch: 8 -> NQB = 16, 4 -> NQB = 8)
__host__ __device__ constexpr bool i8_tile_resident(int nsteps, int nbq) { return nsteps <= (nbq == 8 ? 4 : 8); }

__host__ __device__ constexpr size_t i8_lds_bytes(int mode) {
    return (size_t)4 * kI8SliceBytes + kI8Words * 4 + kI8RsBufs * kI8RsStride * 4 + (mode == MODE_FILTER ? (size_t)kHitCap * 12 : 0);
}

// STEPS3: the row has a multiple of 3 K-steps (the host picks the instantiation): tiles start at corpus-ring phase 0.
// NQB: query blocks of 16 the launch multiplies: 16 (129..256 queries) or 8 (65..128: half the MFMAs, half the slice bytes;
// the slices keep their 32 KiB slots and the bookkeeping its 256-query layout).
// RES: the whole query block fits the four LDS slices (rows of <= 4 K-steps; <= 8 K-steps with 8 query blocks, whose slices
// are half as big): every workgroup loads it once and no slice is re-staged per tile (the host picks it: i8_tile_resident();
// LAG builds keep the staged program).
// TS (tile structure; the host picks it): 0 = generic interval loop; 1 = rows whose K-steps are a multiple of 3 (tiles start at
// corpus-ring phase 0: the tile-structured program, "STEPS3"); 2 = ... a multiple of 6 (768 elements), staged slices: the same
// program with ONE workgroup barrier per TWO K-steps.  The four LDS slices hold the two slices being read and the two
// landing; the barrier behind every odd K-step hands both pairs over at once (the barrier behind an even K-step protected
// nothing that the next one does not: slot (t + 2) & 3, requested during step t, was last read during step t - 2).
// F16 (round 3): the same program over the 2-BYTE shadow (fp16; filter_gemm.h: identical piece order, a 128-byte K-step is 64 elements, two
// v_mfma_f32_16x16x32_f16 per block pair instead of two v_mfma_i32_16x16x64_i8 — byte for byte the same operand traffic and the same matrix time per
// byte).  The accumulators ARE the approximate scores (no scales, no quantisation error: the bound lives in thr[q] = L - eps, anchor_thr_kernel),
// `thr` holds one threshold per query, `bmeta` only has to be 256 readable bytes (the per-tile metadata DMA is kept so that the counted waits stay the
// same: what it fetches is never read), `rscale` and `qscale` are not read.  Filter pass only (the sample pass of the 2-byte filter stays gemm_filter_kernel's).
template <int MODE, int TS, int NQB = 16, bool RES = false, bool F16 = false>
__global__ __launch_bounds__(512, 2) void i8_tile_kernel(const uint4* __restrict__ shadow8, const uint4* __restrict__ qfrag8, int64_t n, int nsteps,
                                                         int64_t ntiles_run, int64_t tile_stride, const float* __restrict__ thr,
                                                         u64* __restrict__ bucket_key, u64* __restrict__ hits, unsigned* __restrict__ hit_cnt,
                                                         int cap_q, unsigned* __restrict__ flags, const float* __restrict__ rscale,
                                                         const float* __restrict__ qscale, const float2* __restrict__ bmeta = nullptr, float eb_scale = 1.0f) {
    // FILTER: `thr` holds thr0[q] = L(q) - A(q) and qscale[768 + q] = B(q): the bound is evaluated per 32-row block,
    // eps(q, block) = A(q) + B(q) * e_block with e_block = bmeta[block].y, the block's own quantisation error norm (its scale is
    // bmeta[block].x).  A row of block b is a candidate iff acc * scale_b * qscale_q >= thr0[q] - B(q) e_b, i.e. (in the units
    // the tests run in) acc * scale_b + u_q e_b >= thr0[q] / qscale_q with u_q = B(q) / qscale_q.  The pre-test uses U = max_q u_q
    // (one fused multiply-add, more permissive: sound), the per-value test the query's own u_q.
    // SAMPLE: rscale[row] (per row; NaN past the count), no thresholds.
    static_assert(MODE == MODE_FILTER || MODE == MODE_SAMPLE, "filter and sample passes only");
    constexpr bool STEPS3 = TS != 0;
    constexpr bool kPair = TS >= 2;
    // TS = 3: the pair program for rows of EXACTLY 6 K-steps (768 elements, the headline shape), every cursor of the tile loop a
    // compile-time constant (run_static6 below)
    constexpr bool kStatic6 = TS >= 3;          // (the name is from the six-step form; TS = 4: the same program over 12 K-steps)
    constexpr int NS = TS == 4 ? 12 : 6;        // K-steps per tile of the static program
    static_assert(TS >= 0 && TS <= 4 && !(kPair && (RES || CODD_I8_LAG)), "pair barriers: staged tile-structured program only");
    static_assert(!kStatic6 || (MODE == MODE_FILTER && !CODD_I8_FUSE_EPI && !CODD_I8_SPREAD_VM), "the static six-step program: filter pass, standalone epilogue");
    static_assert(NQB == 16 || NQB == 8, "256 or 128 queries");
    static_assert(!F16 || (MODE == MODE_FILTER && !RES), "fp16 operands: the staged filter pass only");
    typedef std::conditional_t<F16, f32x4, i32x4> acc4_t;
    constexpr int kDmaPerSlice = NQB / 4;                 // this wave's 1 KiB chunks of a slice
    constexpr int kDmaPerIv = RES ? 0 : kDmaPerSlice;     // slice DMA per interval
    constexpr int kOpsPerIv = 1 + kDmaPerIv + kAPerIv;    // vector-memory operations per interval (see above)
    static_assert(!(RES && CODD_I8_LAG), "the resident program has no lagging half");
    // The resident program synchronises ONCE per tile (nothing is staged cooperatively): the workgroup's shared state is
    // then double-buffered by tile parity — two half hit lists with their counters, two sets of sample keys — so that
    // what the barrier inside tile k hands over (tile k - 1's) is not written again before the barrier inside tile k + 1.
    constexpr int kLists = RES ? 2 : 1;
    constexpr unsigned kListCap = (unsigned)kHitCap / kLists;
    // LDS slot of a slice: 32 KiB; the resident program of 8 query blocks packs its 16 KiB slices (up to 8 K-steps fit)
    constexpr int kSlotBytes = RES && NQB == 8 ? kI8SliceBytes / 2 : kI8SliceBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* lds_w = reinterpret_cast<unsigned*>(smem + 4 * kI8SliceBytes);
    float* lds_rs = reinterpret_cast<float*>(lds_w + kI8Words);                  // row scales of the tiles in flight
    unsigned* lds_hits = reinterpret_cast<unsigned*>(lds_rs + kI8RsBufs * kI8RsStride);
    u64* lds_k = reinterpret_cast<u64*>(lds_w);                                  // SAMPLE: best (score, row) key of the tile per query
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds0 = (unsigned)(size_t)(lds_byte*)smem;                     // LDS byte address of the first query slice

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned n_rows = (unsigned)n;  // (row slots are 32-bit: keys carry them in their low word)

    // tile ordinals and strides are 32-bit in here (row slots are: 2^32 rows are 2^24 tiles) — the 64-bit cursors of round 2 and
    // the round-3 additions together pushed the kernel past its 102 scalar registers: hipcc parked the query slices' buffer
    // descriptor in lanes of a vector register and re-read it with four v_readlane in front of EVERY slice DMA (2,300 of them
    // in the unrolled program, 16 per K-step and wave)
    const int G = (int)gridDim.x;
    const int first_u = (int)blockIdx.x;
    const int tstride = (int)tile_stride;
    const int my_tiles = (int)ntiles_run > first_u ? ((int)ntiles_run - first_u + G - 1) / G : 0;
    const int T = my_tiles * nsteps;  // K-steps of this workgroup
    if (T == 0) return;

    if (MODE == MODE_FILTER) {
        if (tid < 256) {
            const float th = F16 ? thr[tid] : thr[tid] / qscale[F16 ? 0 : tid];  // the test runs on acc * rscale[row] (fp16: on the accumulator itself)
            // the pre-test (largest accumulator of a block pair x the block's scale) is only conclusive for a positive
            // threshold: thresholds <= 0 (and NaN: a zero query) always take the exact per-value test.  Stored rounded DOWN to
            // bf16 (truncation of a positive float; -inf and +inf are exact), query q = 16 qb + c in half (qb & 1) of word
            // [(((qb >> 1) >> 2) * 16 + c) * 4 + ((qb >> 1) & 3)]: a lane's 16 thresholds are two 16-byte reads, consecutive
            // lanes read consecutive 16 bytes (no bank conflict)
#ifdef CODD_I8_EXP_NOHITS
            const float pre = INFINITY;  // diagnostic: no pair ever passes the pre-test
#else
            const float pre = th > 0.0f ? th : -INFINITY;
#endif
            const int qb = tid >> 4, c = tid & 15, j = qb >> 1;
            reinterpret_cast<unsigned short*>(lds_w)[2 * (((j >> 2) * 16 + c) * 4 + (j & 3)) + (qb & 1)] = (unsigned short)(__float_as_uint(pre) >> 16);
            lds_w[576 + tid] = __float_as_uint(th);
            lds_w[320 + tid] = __float_as_uint(F16 ? 1.0f : qscale[F16 ? 0 : tid]);
            // u_q (0 for padding and zero queries: their scale is 0) and the workgroup's maximum of it, per wave here, folded
            // behind the prologue's barrier
            const float qs = F16 ? 1.0f : qscale[F16 ? 0 : tid];
            const float u = (!F16 && qs > 0.0f) ? qscale[F16 ? 0 : 768 + tid] / qs : 0.0f;
            lds_w[1088 + tid] = __float_as_uint(u);
            const float um = i8_wave_max(u);
            if (lane == 0) lds_w[260 + wave] = __float_as_uint(um);
        }
        if (tid < kLists) lds_w[256 + tid] = 0u;
    } else {
        if (tid < 256) {
            lds_k[tid] = 0ull;
            if (kLists == 2) lds_k[256 + tid] = 0ull;
        }
    }

    const int lane16 = lane * 16;
    const i32x4 rsrc_q = i8_rsrc(qfrag8, (unsigned)(nsteps * kI8SliceBytes));
    constexpr int step_bytes = 4096;                       // 4 pieces x 1 KiB: this wave's 32 rows x 128 elements
    const int tile_bytes = 8 * nsteps * step_bytes;

    float u_max = 0.0f;  // FILTER: max over the batch of B(q) / qscale_q (uniform; set behind the prologue's barrier)
    acc4_t acc[2][NQB];
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) acc[rs][qb] = acc4_t{0, 0, 0, 0};

    // ---- corpus fragments: HBM -> registers, ring of 3 K-steps, the cursor runs 2 steps ahead (1 for a lagging wave) ----
    int l_u = first_u;  // run-tile ordinal of the next step to load
    int l_s = 0, l_left = T;
    u32x4 ring[3][4];
    auto load_a = [&](u32x4(&dst)[4]) __attribute__((always_inline)) {
#ifdef CODD_I8_EXP_SAMETILE
        const char* base = reinterpret_cast<const char*>(shadow8) + (wave * nsteps + l_s) * step_bytes;  // diagnostic: corpus served by L2
#else
        const char* base = reinterpret_cast<const char*>(shadow8) + (int64_t)(l_u * tstride) * tile_bytes + (wave * nsteps + l_s) * step_bytes;
#endif
        const i32x4 r = i8_rsrc(base, 4096);
        i8_load_b128_nt<0>(dst[0], lane16, r);
        i8_load_b128_nt<1024>(dst[1], lane16, r);
        i8_load_b128_nt<2048>(dst[2], lane16, r);
        i8_load_b128_nt<3072>(dst[3], lane16, r);
        if (--l_left > 0) {  // past the last step the cursor stays on it: loads are unconditional
            if (++l_s == nsteps) { l_s = 0; l_u += G; }
        }
    };

    // the same in pieces (pair program: one vector-memory operation at a time, between MFMA groups)
    i32x4 a_rsrc = {0, 0, 0, 0};
    auto load_a_begin = [&]() __attribute__((always_inline)) {
#ifdef CODD_I8_EXP_SAMETILE
        const char* base = reinterpret_cast<const char*>(shadow8) + (wave * nsteps + l_s) * step_bytes;
#else
        const char* base = reinterpret_cast<const char*>(shadow8) + (int64_t)(l_u * tstride) * tile_bytes + (wave * nsteps + l_s) * step_bytes;
#endif
        a_rsrc = i8_rsrc(base, 4096);
    };
    auto load_a_end = [&]() __attribute__((always_inline)) {
        if (--l_left > 0) {
            if (++l_s == nsteps) { l_s = 0; l_u += G; }
        }
    };

    // ---- query slices: L2 -> LDS by LDS-DMA, slice of step t in LDS slice t & 3, requested two intervals ahead ----
    // wave w moves the 1 KiB chunks 8j + w (j < kDmaPerIv) of a slice: 64 lanes x 16 bytes each, contiguous on both sides
    // (pieces are ordered [query block][K half]: the first 2 NQB chunks of a 256-query slice are query blocks 0..NQB-1)
    int q_s = 0;
    auto stage_dma = [&](int slot) __attribute__((always_inline)) {
        const int soff = q_s * kI8SliceBytes + wave * 1024;
        const unsigned dst = lds0 + (unsigned)(slot * kSlotBytes + wave * 1024);
#pragma unroll
        for (int j = 0; j < kDmaPerSlice; ++j) i8_dma_b128(dst + j * 8192, lane16, rsrc_q, soff + j * 8192);
        q_s = q_s + 1 == nsteps ? 0 : q_s + 1;
    };
    auto stage_dma_piece = [&](int slot, int j, bool last) __attribute__((always_inline)) {
        const int soff = q_s * kI8SliceBytes + wave * 1024;
        const unsigned dst = lds0 + (unsigned)(slot * kSlotBytes + wave * 1024);
        i8_dma_b128(dst + j * 8192, lane16, rsrc_q, soff + j * 8192);
        if (last) q_s = q_s + 1 == nsteps ? 0 : q_s + 1;
    };

    // ---- row scales of run-tile ordinal u (this workgroup's tile number `ord`) -> LDS buffer ord & 1, by DMA too ----
    // one 256-byte DMA per wave and interval: the scales of rows [r0, r0 + 64) of the tile, r0 = min(32 wave, 192) — the
    // wave's own 32 rows among them (neighbouring waves write the same bytes twice).  Every interval of a tile repeats
    // the request (same bytes): the number of operations per interval stays fixed, which the counted waits rely on.
    // The host keeps rscale[r] = NaN for count <= r < the next multiple of 256: `acc * NaN >= thr` is false for every
    // threshold, so the epilogue needs no row < n test; tiles past the corpus (padding intervals) read zeros.
    const int lane4 = lane * 4;
    auto rs_dma = [&](int u, int ord) __attribute__((always_inline)) {
        if constexpr (MODE == MODE_FILTER) {
            // the tile's eight blocks' {scale, error norm}: dwords [2 w], [2 w + 1] of the buffer belong to wave w (one 256-byte DMA:
            // 32 blocks from the tile's first one — the allocation has the head room; every wave writes the same bytes)
            // (fp16 operands: no scales — the same 256 bytes every time, never read; the request keeps the operation count)
            const unsigned blk0 = F16 ? 0u : (unsigned)(u * tstride) * (unsigned)(kTileRows / 32);
            const unsigned nblk = F16 ? 1u : (n_rows + 31u) / 32u;
            i8_dma_b32(lds0 + (unsigned)(4 * kI8SliceBytes + kI8Words * 4 + ((ord & 1) * kI8RsStride) * 4), lane4,
                       i8_rsrc(bmeta + (blk0 < nblk ? blk0 : 0), blk0 < nblk ? 256u : 0u), 0);
        } else {
        const int64_t row0 = (int64_t)(u * tstride) * kTileRows;
        const int64_t bound = (n + kTileRows - 1) / kTileRows * kTileRows;
        const int64_t left = bound - row0;
        const int rows_here = left >= kTileRows ? kTileRows : (left > 0 ? (int)left : 0);
        const int r0 = wave * 32 < 192 ? wave * 32 : 192;
        i8_dma_b32(lds0 + (unsigned)(4 * kI8SliceBytes + kI8Words * 4 + ((ord & 1) * kI8RsStride + r0) * 4), lane4,
                   i8_rsrc(rscale + (left > 0 ? row0 : 0), (unsigned)(rows_here * 4)), r0 * 4);
        }
    };

    // ---- epilogues ----
    // FILTER: what a lane needs to test a tile's accumulators, set up once per tile (epi_begin); the test of one block pair
    // — (2 row blocks of the wave) x (query block qb) — is epi_pair.  The standalone epilogue runs the 16 pairs back to back;
    // the tile-structured program (CODD_I8_FUSE_EPI) runs pair qb right in front of the first MFMAs of the NEXT tile that
    // overwrite its accumulators (first K-step, K half 0, query block qb): all eight waves used to reach the epilogue
    // together and the matrix pipe idled for its whole length; fused, a wave that takes the rare per-value path (LDS atomic
    // round trip inside a branch) leaves its SIMD partner's MFMAs running.
    struct EpiCtx {
        unsigned par;              // which half of the double-buffered shared state the tile writes (uniform)
        unsigned wrow0;            // first row of the wave's 32-row block (uniform)
        float rsl;                 // its scale (uniform; NaN: the whole block lies past the count — every comparison fails)
        float inv_rsl;             // 1 / rsl (uniform)
        float eb;                  // the block's quantisation error norm (uniform)
        float ueb;                 // U * eb: what the pre-test adds to acc * rsl (uniform)
        u32x4 thw[NQB / 8];        // the lane's NQB pre-test thresholds, two bf16 per word (see the kernel's set-up)
    };
    auto epi_begin = [&](int cu, int ord, bool valid) __attribute__((always_inline)) {
        // lane coordinates re-derived behind an opaque asm: hipcc otherwise hoists every per-query-block address of this
        // body out of the interval loop and keeps dozens of registers of loop invariants alive across the MFMA phases
        EpiCtx e;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int c = l & 15;
        const int tile = cu * tstride;
        e.par = kLists == 2 ? (unsigned)(ord & 1) : 0u;
        e.wrow0 = (unsigned)tile * (unsigned)kTileRows + (unsigned)(wave * 32);
        // ONE scale per 32-row block (shadow8_from_rows_kernel), so the wave's rows share it and the pre-test on a pair's
        // largest accumulator is EXACT at pair level: it passes iff some value of the pair passes.  Rows past the count
        // carry NaN: the block's first row exists whenever any of its rows does, and the per-value test masks the others.
        const float rs0 = F16 ? 1.0f : lds_rs[(ord & 1) * kI8RsStride + 2 * wave], eb0 = F16 ? 0.0f : lds_rs[(ord & 1) * kI8RsStride + 2 * wave + 1];
        e.rsl = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(valid ? rs0 : __builtin_nanf(""))));
        e.inv_rsl = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(1.0f / (valid ? rs0 : __builtin_nanf("")))));
        e.eb = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(eb0))) * eb_scale;  // (eb_scale 0: the device-wide bound, thr = L - eps(q))
        e.ueb = e.eb * u_max;
#pragma unroll
        for (int j = 0; j < NQB / 8; ++j) e.thw[j] = *reinterpret_cast<const u32x4*>(lds_w + (j * 16 + c) * 4);
        return e;
    };
    auto epi_pair = [&](auto QB_, const EpiCtx& e) __attribute__((always_inline)) {
        constexpr int qb = decltype(QB_)::value;
        const acc4_t a0 = acc[0][qb], a1 = acc[1][qb];
        // (three v_max3 and one v_max: hipcc turns a balanced tree of two-operand max into five instructions)
        auto mx = [](auto x, auto y) __attribute__((always_inline)) { return x > y ? x : y; };   // (no NaN among finite products of finite operands)
        const auto m = mx(mx(mx(mx(mx(a0[0], a0[1]), a0[2]), mx(mx(a0[3], a1[0]), a1[1])), a1[2]), a1[3]);
        // no accumulator of the pair reaches the threshold when the largest one does not (same scale, rounding is
        // monotone, a non-positive accumulator is below a positive threshold anyway).  The pre-test threshold is the exact one
        // rounded DOWN to bf16 (two per register: 16 of them cost 8 registers instead of 16 at the point where the program
        // is tightest); the per-value test below uses the exact one.
        const unsigned w = e.thw[qb >> 3][(qb >> 1) & 3];
        const float thp = __uint_as_float((qb & 1) ? (w & 0xffff0000u) : (w << 16));
        bool pass;
        if constexpr (F16) pass = (float)m >= thp;   // the accumulator IS the approximate score
        else pass = __builtin_fmaf((float)m, e.rsl, e.ueb) >= thp;
        if (__builtin_expect(__any(pass), 0)) {
            int l = lane;
            asm volatile("" : "+v"(l));  // (lane coordinates derived BEHIND the opaque asm: hipcc otherwise computes them once at kernel start and parks them in scratch)
            const int c = l & 15, lg = l >> 4;
            const unsigned q = (unsigned)(qb * 16 + c);
            const unsigned row0 = e.wrow0 + (unsigned)(4 * lg);  // + 16 * rs + r
            const float th = __uint_as_float(lds_w[576 + q]) - __uint_as_float(lds_w[1088 + q]) * e.eb;  // the query's exact threshold for this block (thr0 / qscale - u_q e_b)
            // Branches are what this path pays for (no prediction: every taken one refills the wave's instruction buffer,
            // every exec-mask test waits for the compare), so the common case — no lane holds more than one hit among
            // its 8 rows — runs straight-line: the per-value tests select the lane's hit and count them, one append.
            auto append = [&](float val, int i) __attribute__((always_inline)) {
                const unsigned slot = atomicAdd(&lds_w[256 + e.par], 1u);
                const unsigned row = row0 + (unsigned)(16 * (i >> 2) + (i & 3));
                if (slot < kListCap) {
                    unsigned* h = lds_hits + (e.par * kListCap + slot) * 3;
                    h[0] = __float_as_uint(val);   // (the query's scale is applied by the flush)
                    h[1] = row;
                    h[2] = q;
                }
#ifndef CODD_I8_EXP_NOGLOBAL  // (diagnostic: hits past a full list are dropped)
                else {
                    // workgroup list full (a dense cluster many queries point at, more hits inside one tile than the
                    // list holds): straight to the query's global list.  Slow (a returning global atomic per hit) but
                    // complete: the query keeps its candidates and needs no fallback.
                    const unsigned gslot = atomicAdd(&hit_cnt[q * kHitCntStride], 1u);
                    if (gslot < (unsigned)cap_q) hits[(int64_t)q * cap_q + gslot] = make_key(val * __uint_as_float(lds_w[320 + q]), row);
                }
#endif
            };
            // The per-value test runs on the INTEGER accumulators against T, a lower bound of the smallest accumulator whose score
            // (float)a * rsl reaches th: (int)(th / rsl) - 2 — two accumulator levels of slack (1/90,000 of a threshold each on the
            // bench corpus) over the rounding of the division, so no hit of the float expression is lost and next to none is
            // added (the hit list may hold extra rows, never miss one).  One compare per value instead of convert + multiply +
            // compare: the eight waves of a workgroup are all in their epilogues at once, so its instruction count is wall time.
            // NaN threshold (a zero query): nothing passes, as with the float compare.  Rows past the count: masked in the last tile.
            typedef std::conditional_t<F16, float, int> val_t;
            val_t T;      // a value is a hit iff it is >= T
            val_t kMasked;
            if constexpr (F16) {
                T = th;   // (NaN: no comparison holds)
                kMasked = -INFINITY;
            } else {
                const float x = th * e.inv_rsl;
                T = x >= 2.0e9f ? INT_MAX : (x <= -2.0e9f ? INT_MIN + 1 : (int)x - 2);
                if (!(th == th)) T = INT_MAX;
                kMasked = INT_MIN;
            }
            val_t av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = i < 4 ? a0[i] : a1[i - 4];
            if (e.wrow0 + 32u > n_rows) {  // (uniform: the corpus's last 32-row block only)
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = row0 + (unsigned)(16 * (i >> 2) + (i & 3)) < n_rows ? av[i] : kMasked;
            }
            val_t sa = 0;
            int si = -1, cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool h = av[i] >= T;
                sa = h ? av[i] : sa;
                si = h ? i : si;
                cnt += h ? 1 : 0;
            }
            const float sv = F16 ? (float)sa : (float)sa * e.rsl;  // the first-generation kernel's expression (its rscale[row] IS the block's scale)
#ifdef CODD_I8_EXP_NOAPPEND
            asm volatile("" ::"v"(sv), "v"(si), "v"(cnt));  // diagnostic: the per-value test runs, nothing is appended
#else
            if (__builtin_expect(__any(cnt > 1), 0)) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (av[i] >= T) append(F16 ? (float)av[i] : (float)av[i] * e.rsl, i);
            } else if (si >= 0) {
                append(sv, si);
            }
#endif
        }
    };
    auto epilogue = [&](int cu, int ord) __attribute__((always_inline)) {
#ifdef CODD_I8_EXP_NOEPI
        if (true) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) { asm volatile("" ::"v"(acc[0][qb])); asm volatile("" ::"v"(acc[1][qb])); }
        } else if (false) {
#else
        if constexpr (MODE == MODE_FILTER) {
#endif
            const EpiCtx e = epi_begin(cu, ord, true);
            static_for<NQB>([&](auto QB_) __attribute__((always_inline)) { epi_pair(QB_, e); });
        } else {
            int c = lane & 15, lg = lane >> 4;
            asm volatile("" : "+v"(c), "+v"(lg));
            const int tile = cu * tstride;
            const unsigned par = kLists == 2 ? (unsigned)(ord & 1) : 0u;
            const float* rsb = lds_rs + (ord & 1) * kI8RsStride;
            const f32x4 rsc0 = *reinterpret_cast<const f32x4*>(rsb + wave * 32 + 4 * lg);
            const f32x4 rsc1 = *reinterpret_cast<const f32x4*>(rsb + wave * 32 + 16 + 4 * lg);
            // best (score, row) of the wave's 32 rows per query, folded into the tile's keys in LDS.  The 32 rows share one scale
            // (block-scaled shadow), so the order of their scores is the order of their accumulators: (accumulator, lower row
            // first) packs into one int — |accumulator| <= 127 * 127 * 4096 < 2^26, five bits for the row — and the fold across
            // the lane's 8 values and the 4 lanes that hold a query is integer max + two lane swaps; one lane per query builds
            // the key.  (INT_MIN: a row past n.)
            const bool ragged = (int64_t)(tile + 1) * kTileRows > n;
            const float rsl = fmaxf(fmaxf(fmaxf(rsc0[0], rsc0[1]), fmaxf(rsc0[2], rsc0[3])), fmaxf(fmaxf(rsc1[0], rsc1[1]), fmaxf(rsc1[2], rsc1[3])));
            // (the key is built by lanes 0..15, whose rows 0..3 of the block are valid whenever any row of it is)
            int code[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) code[i] = 31 - (4 * lg + 16 * (i >> 2) + (i & 3));
            const unsigned wrow0 = (unsigned)tile * (unsigned)kTileRows + (unsigned)(wave * 32);
            auto fold = [&](auto RAGGED) __attribute__((always_inline)) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    int m = INT_MIN;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        int p = (acc[i >> 2][qb][i & 3] << 5) | code[i];
                        if (decltype(RAGGED)::value && (int64_t)(wrow0 + (unsigned)(31 - code[i])) >= n) p = INT_MIN;
                        m = max(m, p);
                    }
                    const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)m, (unsigned)m, false, false);
                    m = max((int)s16[0], (int)s16[1]);
                    const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)m, (unsigned)m, false, false);
                    m = max((int)s32[0], (int)s32[1]);
                    if (lg == 0 && m != INT_MIN) {
                        const u64 key = make_key((float)(m >> 5) * rsl, wrow0 + (unsigned)(31 - (m & 31)));
                        atomicMax(reinterpret_cast<unsigned long long*>(&lds_k[par * 256 + qb * 16 + c]), (unsigned long long)key);
                    }
                }
            };
            if (ragged) fold(std::true_type{});
            else fold(std::false_type{});
        }
        if (!(STEPS3 && !CODD_I8_LAG)) {  // (the tile-structured program starts every tile from zero accumulators instead)
#pragma unroll
            for (int rs = 0; rs < 2; ++rs)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) acc[rs][qb] = acc4_t{0, 0, 0, 0};
        }
    };

    // MFMAs of one K-step: corpus fragments in ring slot SLOT, query slice at LDS byte address qaddr (+ lane * 16).  2 NQB
    // groups (K half ks, query block qb) of 2 MFMAs; the fragment of group g + kBD is requested when group g has
    // consumed its register set.  Before group g the reads of groups g+1 .. min(31, g + kBD - 1) are the only younger
    // LGKM operations and LDS returns in order, hence lgkmcnt(that many).  FIRST: the first K-step of a tile starts its
    // accumulators from zero (no clearing pass after the epilogue).
    // the first kBD fragment reads of a K-step: issued right behind the interval's barrier, BEFORE the interval's DMA and
    // corpus-load instructions, so that their LDS round trip runs beside that issue work instead of after it (every wave
    // of the workgroup is at this point at the same time: nothing else feeds the matrix pipe here)
    auto frag_prefetch = [&](i32x4(&b)[kBD], unsigned qaddr) __attribute__((always_inline)) {
        static_for<kBD>([&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            lds_read_b128_asm<(((g % NQB) * 2) + (g / NQB)) * 1024>(b[g], qaddr);
        });
    };
    // FUSED: the first K-step of a tile also carries the PREVIOUS tile's epilogue — pair qb is tested right in front of the
    // MFMAs that restart its accumulators from zero (K half 0, query block qb)
    // vm(g): vector-memory operations the pair program issues BEHIND group g's MFMAs (see the interval loop), nothing otherwise
    auto mfma_step = [&](auto SLOT, auto FIRST, auto FUSED, i32x4(&b)[kBD], unsigned qaddr, const EpiCtx* ectx, auto&& vm) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        constexpr bool first = decltype(FIRST)::value;
        constexpr bool fused = decltype(FUSED)::value;
        static_assert(!fused || first, "the fused epilogue belongs to a tile's first K-step");
        constexpr int kGroups = 2 * NQB;  // (K half, query block) groups of 2 MFMAs per K-step
        static_for<kGroups>([&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value, ks = g / NQB, qb = g % NQB;
            constexpr int younger = (g + kBD - 1 < kGroups - 1 ? g + kBD - 1 : kGroups - 1) - g;
            if constexpr (fused && ks == 0) {
                epi_pair(std::integral_constant<int, qb>{}, *ectx);
                __builtin_amdgcn_sched_barrier(0);
            }
            lgkm_wait_asm<younger>(b[g % kBD]);
            const i32x4 a0 = __builtin_bit_cast(i32x4, ring[slot][0 * 2 + ks]);
            const i32x4 a1 = __builtin_bit_cast(i32x4, ring[slot][1 * 2 + ks]);
            const acc4_t zero = {0, 0, 0, 0};
            if constexpr (F16) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                acc[0][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a0), __builtin_bit_cast(h8, b[g % kBD]), first && ks == 0 ? zero : acc[0][qb], 0, 0, 0);
                acc[1][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a1), __builtin_bit_cast(h8, b[g % kBD]), first && ks == 0 ? zero : acc[1][qb], 0, 0, 0);
            } else {
                acc[0][qb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[g % kBD], first && ks == 0 ? zero : acc[0][qb], 0, 0, 0);
                acc[1][qb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[g % kBD], first && ks == 0 ? zero : acc[1][qb], 0, 0, 0);
            }
            if constexpr (g + kBD < kGroups) {
                constexpr int g2 = g + kBD;
                lds_read_b128_asm<(((g2 % NQB) * 2) + (g2 / NQB)) * 1024>(b[g % kBD], qaddr);
            }
            vm(G_);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- the interval loop ----
    // interval t: waves 0..3 run step t, waves 4..7 step t - LAG; every wave requests its share of slice t + 2 and loads
    // the scales of step t's tile; one barrier.  Slices read: t & 3 and (t - 1) & 3; landing: (t + 1) & 3, (t + 2) & 3.
    // Issue order per interval: [a pending epilogue], scale DMA, slice DMA, corpus loads, MFMAs, barrier.  Waits (operations
    // younger than the one waited for, in queue order):
    //   corpus step loaded in interval t-2, at the MFMAs of t : everything of t-1 and of t                  = 2 kOpsPerIv
    //   DMA of interval t-1, at the barrier of t : the corpus loads of t-1 and everything of t              = kAPerIv + kOpsPerIv
    auto run = [&](auto LAG_) __attribute__((always_inline)) {
        constexpr int LAG = decltype(LAG_)::value;
        rs_dma(first_u, 0);  // (the first tile's scales)
        if constexpr (RES) {
            for (int s = 0; s < nsteps; ++s) stage_dma(s);  // the whole query block, once
        } else {
            stage_dma(0);
            stage_dma(1);
        }
        // corpus prologue: a lagging wave starts with step "-1" on an all-zero ring slot
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ring[2][kk] = u32x4{0u, 0u, 0u, 0u};
        load_a(ring[0]);
        if (LAG == 0) load_a(ring[1]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slices 0, 1, the first scales and corpus steps have landed
        if (MODE == MODE_FILTER) {
            const float um = fmaxf(fmaxf(__uint_as_float(lds_w[260]), __uint_as_float(lds_w[261])), fmaxf(__uint_as_float(lds_w[262]), __uint_as_float(lds_w[263])));
            u_max = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(um)));
        }

#ifdef CODD_I8_EXP_STAMPS
        // diagnostic build (guide: in-kernel stamps): per wave, shader cycles spent per phase, summed over its intervals.  s_memtime
        // returns through the scalar cache, so each stamp drains lgkmcnt: placed where the LDS queue is empty anyway, except the
        // one behind the corpus wait (4 fragment reads in flight there: the first MFMA group waits for the first of them anyway)
        unsigned long long st_pre = 0, st_vm = 0, st_mfma = 0, st_sync = 0, st_epi = 0, st_iv = 0;
        const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
        int pub_ord = 0;         // SAMPLE: next tile ordinal to publish
        if constexpr (kStatic6) {
            // ---- the static six-step tile program -----------------------------------------------------------------------------
            // The generic interval carries four cursors (load, slice, compute, workgroup) through compare-and-select chains and
            // rebuilds two buffer descriptors from 64-bit products: 74 scalar instructions per K-step and wave in front of 64
            // MFMAs (rocprofv3 SQ_INSTS_SALU), issued by all eight waves at the same moment — behind a barrier — with the matrix
            // pipe idle.  With 6 K-steps per tile everything is a constant of the unrolled interval index i: the slice of step
            // t + 2 is slice (i + 2) % 6 in LDS slot ((i + 2) & 3) ^ 2 (o & 1) [6 o + i = 2 o + i mod 4], the corpus step to load is
            // step (i + 2) % 6 of this tile (i < 4) or of the next one, whose base address is one 64-bit add per TILE away, and
            // the block metadata is requested once per tile (interval 0) — which makes the number of operations per interval
            // 9, 8, 8, 8, 8, 8 (1 + kDmaPerSlice + 4 in general) and the counted waits below.
            static_assert(LAG == 0, "no lagging half");
            constexpr int kOps0 = 1 + kDmaPerSlice + kAPerIv, kOpsN = kDmaPerSlice + kAPerIv;
            const unsigned ldsw = lds0 + (unsigned)(wave * 1024);   // this wave's share of a slice slot
            const int wq = wave * 1024;                              // ... and of a slice in the query buffer
            int u = first_u;                                          // run-tile ordinal of tile o
            const char* A_cur = reinterpret_cast<const char*>(shadow8) + (int64_t)(u * tstride) * tile_bytes + wave * (NS * step_bytes);
            const int64_t dA = (int64_t)(G * tstride) * tile_bytes;
            // LDS slot of the tile's interval i: (NS o + i) & 3 = (i & 3) ^ 2 (o & 1) for 6 K-steps per tile, i & 3 for 12: px = the bytes to flip
            unsigned px = 0u;
            constexpr unsigned kPxStep = (unsigned)((NS & 3) * kSlotBytes);
            unsigned tpar = 0u;                                       // tile parity: which of the two metadata buffers the tile uses
            // corpus loads "of interval I": step (I + 2) % 6 of this tile (I < 4) or the next one, into the ring slot interval I - 1 has
            // just consumed.  Intervals 1 and 3 issue their own in their heads and those of the EVEN interval behind them in their
            // tails, in front of their barriers (CODD_I8_EARLY_A): the heads of intervals 2 and 4 — which all eight waves run at the same
            // moment, right behind the barrier — are left with the slice DMA only, and the loads go out while the SIMD partners are
            // still staggered.  (Not interval 5 for the next tile's interval 0: the epilogue between them needs the third ring slot's
            // registers — with all three slots in flight it spilled 16 registers to scratch.)
            auto c_load = [&](auto I_, const char* A_next) __attribute__((always_inline)) {
                constexpr int i = decltype(I_)::value % NS;
                constexpr bool next_tile = decltype(I_)::value >= NS - 2;
                constexpr int li = ((i % 3) + 2) % 3;
                constexpr int st = (i + 2) % NS;
                const char* base = (next_tile ? A_next : A_cur) + st * step_bytes;
                const i32x4 r = i8_rsrc(base, 4096);
                i8_load_b128_nt<0>(ring[li][0], lane16, r);
                i8_load_b128_nt<1024>(ring[li][1], lane16, r);
                i8_load_b128_nt<2048>(ring[li][2], lane16, r);
                i8_load_b128_nt<3072>(ring[li][3], lane16, r);
            };
            auto s_interval = [&](auto I_, const char* A_next) __attribute__((always_inline)) {
                constexpr int i = decltype(I_)::value;
                constexpr int ci = i % 3;
                constexpr int st = (i + 2) % NS;                      // the step whose slice (and corpus fragments) are requested here
                constexpr bool odd = (i & 1) != 0;
                __builtin_amdgcn_sched_barrier(0);
                const unsigned qaddr = lds0 + (((unsigned)((i & 3) * kSlotBytes)) ^ px) + (unsigned)lane16;
                i32x4 b[kBD];
                if (CODD_I8_EARLY_FRAGS) frag_prefetch(b, qaddr);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i == 0) rs_dma(u, (int)tpar);          // (tile parity picks the scale buffer: ord & 1)
#ifndef CODD_I8_EXP_NODMA
                {
                    // (the two bases pass through an opaque asm per interval: hipcc otherwise hoists every "base + constant" of the unrolled
                    //  12-step tile out of the tile loop — 12 intervals x 8 of them — and spills 86 scalar registers to keep them)
                    unsigned ldsw_i = ldsw;
                    int wq_i = wq;
                    if constexpr (NS > 6) asm volatile("" : "+s"(ldsw_i), "+s"(wq_i));   // (the six-step program keeps them: 106 SGPRs, none spilled)
                    const unsigned dst = ldsw_i + (((unsigned)(((i + 2) & 3) * kSlotBytes)) ^ px);
#pragma unroll
                    for (int j = 0; j < kDmaPerSlice; ++j) i8_dma_b128_s(dst + j * 8192, lane16, rsrc_q, wq_i + st * kI8SliceBytes + j * 8192);
                }
#endif
                // (even intervals from 2 on: bit 0 of CODD_I8_EARLY_A for 2, 6, 10, bit 1 for 4, 8)
                auto early = [](int j) constexpr { return j >= 2 && j % 2 == 0 && (CODD_I8_EARLY_A & (j % 4 == 2 ? 1 : 2)) != 0; };
                constexpr bool early_here = early(i);                    // this interval's loads went out in the tail of the one before
                constexpr bool early_next = i + 1 < NS && early(i + 1);  // ... and the next one's go out in this one's tail
                if constexpr (!early_here) c_load(I_, A_next);
                // the corpus step requested two intervals ago; younger in the queue (issue order, kD = kDmaPerSlice, rs = the block
                // metadata DMA of interval 0):
                // (counted by walking the tile's issue order backwards from this interval's last operation to the loads of interval i - 2:
                //  head(j) = [rs (j == 0)] [kD slice DMA] [4 corpus loads unless early], tail(j) = [4: the early loads of j + 1])
                constexpr int kYounger = [&]() constexpr {
                    int n = 0;
                    // this interval's head, then backwards over tail(i-1), head(i-1), tail(i-2), head(i-2) until the loads of i - 2 are reached
                    n += (i == 0 ? 1 : 0) + kDmaPerSlice + (early(i) ? 0 : 4);
                    const int j1 = (i + NS - 1) % NS, j2 = (i + NS - 2) % NS;
                    if (early(i)) n += 4;                                   // tail(i-1): this interval's own early loads
                    n += (j1 == 0 ? 1 : 0) + kDmaPerSlice + (early(j1) ? 0 : 4);   // head(i-1)
                    if (early(j1)) n += 4;                                  // tail(i-2): the early loads of i - 1
                    if (early(j2)) {
                        // the loads of i - 2 sit in tail(i-3): behind them the whole head(i-2)
                        n += (j2 == 0 ? 1 : 0) + kDmaPerSlice;
                    }
                    return n;
                }();
                i8_wait_vm<kYounger>(ring[ci][0], ring[ci][1], ring[ci][2], ring[ci][3]);
                __builtin_amdgcn_sched_barrier(0);
                if (!CODD_I8_EARLY_FRAGS) frag_prefetch(b, qaddr);
                mfma_step(std::integral_constant<int, ci>{}, std::integral_constant<bool, i == 0>{}, std::false_type{}, b, qaddr, nullptr, [](auto) {});
                if constexpr (odd) {
                    if constexpr (early_next) {
                        __builtin_amdgcn_sched_barrier(0);
                        c_load(std::integral_constant<int, i + 1>{}, A_next);
                    }
                    // the slices of steps t + 1 and t + 2 (and, i == 1, the tile's block metadata) have landed: younger than this
                    // interval's slice DMA are only its corpus loads (and the next interval's, issued early)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(early_next ? 2 * kAPerIv : kAPerIv) : "memory");
                }
            };
            for (int o = 0; o < my_tiles; ++o) {
                const char* A_next = o + 1 < my_tiles ? A_cur + dA : A_cur;   // (past the last tile the loads stay on it: unconditional, never consumed)
                static_for<NS>([&](auto I_) __attribute__((always_inline)) { s_interval(I_, A_next); });
                // The workgroup's hit list is flushed HERE when it runs full: behind the tile's last barrier (three barriers after the
                // previous tile's epilogue: every wave's appends are in, the count is stable) and in front of this tile's epilogue — the one
                // point of the tile where only two of the three ring slots are in flight, so the flush code finds its registers (behind the
                // barrier of interval 1, with interval 2's loads issued early, it spilled 9 of them).  The rare branch ends in its own
                // barrier: the counter is back at 0 before any wave appends again.
                if (o >= 1) {
                    const unsigned cnt = lds_w[256];
                    if (cnt > (unsigned)(CODD_FLUSH_AT)) {
                        int tid_f = tid;
                        asm volatile("" : "+v"(tid_f));  // (or hipcc hoists &hit_cnt[tid] out of the tile loop)
                        flush_hits_binned(lds_hits, cnt < kListCap ? cnt : kListCap, tid_f, lds_w + 832, hits, hit_cnt, cap_q, reinterpret_cast<const float*>(lds_w + 320));
                        __syncthreads();
                        if (tid == 0) lds_w[256] = 0u;
                        __syncthreads();
                    }
                }
                epilogue(u, (int)tpar);
                u += G;
                A_cur = A_next;
                px ^= kPxStep;
                tpar ^= 1u;
            }
            // (the corpus loads of the last two intervals are never consumed: nothing of this wave may be in flight past this point)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
        int c_u = first_u;       // compute cursor (this wave's step s = t - LAG)
        int c_s = 0, c_ord = 0;
        int w_u = first_u;       // workgroup cursor (step t)
        int w_s = 0, w_ord = 0;
        bool pending = false;    // a finished tile whose epilogue has not run yet
        int p_u = 0;
        int p_ord = 0;
        const int TI = T + (CODD_I8_LAG ? 1 : 0);
        // one interval; IU (= t mod 3) picks the corpus ring slots statically.  EPI: a pending epilogue may run inside
        // (the generic loop); the tile-structured loop below runs it between intervals instead.
        // (A loop that is not unrolled, with a uniform switch around three static copies of [loads, wait, MFMAs], made
        // hipcc merge the 128 accumulator registers across the arms with copies: 700 bytes of scratch.)
        // BOOK: the interval may be a tile's second one (w_s == 1), behind whose barrier the workgroup's bookkeeping runs (hit-list
        // flush, sample keys): the tile-structured program knows statically which of its unrolled intervals that is, and the
        // flush code — 2 KB, with a handful of loop-invariant registers hipcc hoists out of the loop — exists once instead of six times
        // SYNC: 0 = a barrier behind every interval (the resident program: behind a tile's second interval only); pair program:
        // 1 = nothing behind this (even) interval, 2 = behind this (odd) one every DMA this wave has issued has landed, barrier.
        auto interval = [&](auto IU, auto EPI, auto FIRST, auto FUSED, auto BOOK, auto SYNC, int t, const EpiCtx* ectx) __attribute__((always_inline)) {
            constexpr int sync = decltype(SYNC)::value;
            constexpr int iu = decltype(IU)::value;
            constexpr int ci = (iu + 3 - LAG) % 3, li = (ci + 2) % 3;
            if (decltype(EPI)::value && pending) {
                epilogue(p_u, p_ord);
                pending = false;
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef CODD_I8_EXP_STAMPS
            const unsigned long long T0 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
#endif
            const unsigned qaddr = lds0 + (unsigned)((RES ? c_s : ((t - LAG) & 3)) * kSlotBytes + lane16);
            i32x4 b[kBD];
#ifdef CODD_I8_EXP_STAMPS
            unsigned long long T1 = 0, T2 = 0;
#endif
            if constexpr (sync != 0 && CODD_I8_SPREAD_VM) {
                // Pair program: the interval's vector-memory operations go out one at a time BEHIND MFMA groups instead of in a block
                // in front of them.  All eight waves reach an interval together, and a CU's address path takes one 1-KiB
                // wave-instruction per ~16 cycles: the block cost every wave 500-900 cycles per K-step with the matrix pipe idle
                // (profiles/r3/i8_tile_stamps.txt: 17 % of the launch).  Issue order (what the counted waits rely on):
                //   even interval t: 4 corpus loads (step t + 2), scale DMA, the slices of steps t + 2 AND t + 3 (2 kDmaPerSlice DMA);
                //   odd interval: 4 corpus loads, scale DMA.  Operation i goes behind group i * kStride.
                // Younger than the corpus loads of interval t - 2 at the top of interval t: the rest of t - 2 and all of t - 1
                //   = (1 + 2 kDmaPerSlice) + 5 (t even) = 1 + (5 + 2 kDmaPerSlice) (t odd);
                // younger than the slice DMA of even interval t - 1 at the barrier behind odd interval t: t's five operations.
                constexpr bool even = sync == 1;
                constexpr int kStride = (2 * NQB) / 16;
                constexpr int kOpsHere = 5 + (even ? 2 * kDmaPerSlice : 0);
                static_assert((kOpsHere - 1) * kStride + 1 < 2 * NQB, "every operation has its group");
                i8_wait_vm<6 + 2 * kDmaPerSlice>(ring[ci][0], ring[ci][1], ring[ci][2], ring[ci][3]);
                __builtin_amdgcn_sched_barrier(0);
                frag_prefetch(b, qaddr);
#ifdef CODD_I8_EXP_STAMPS
                __builtin_amdgcn_sched_barrier(0);
                T1 = T2 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
#endif
                mfma_step(std::integral_constant<int, ci>{}, FIRST, FUSED, b, qaddr, ectx, [&](auto G_) __attribute__((always_inline)) {
                    constexpr int g = decltype(G_)::value;
                    // SIMD partners (waves w and w + 4) take turns: a 1-KiB vector-memory instruction holds its wave's issue for ~60
                    // cycles, which the partner's MFMAs cover only if the partner is not issuing one of its own at the same moment:
                    // waves 0..3 issue behind the even groups, waves 4..7 behind the odd ones (CODD_I8_SPREAD_VM == 2)
                    constexpr int phase = (CODD_I8_SPREAD_VM == 2 && kStride == 2) ? g % kStride : 0;
                    if constexpr ((g - phase) % kStride == 0 && (g - phase) / kStride < kOpsHere) {
                        if (CODD_I8_SPREAD_VM == 2 && kStride == 2 && (wave >= 4) != (phase == 1)) return;
                        constexpr int i = (g - phase) / kStride;
                        if constexpr (i == 0) load_a_begin();
                        if constexpr (i == 0) i8_load_b128_nt<0>(ring[li][0], lane16, a_rsrc);
                        if constexpr (i == 1) i8_load_b128_nt<1024>(ring[li][1], lane16, a_rsrc);
                        if constexpr (i == 2) i8_load_b128_nt<2048>(ring[li][2], lane16, a_rsrc);
                        if constexpr (i == 3) { i8_load_b128_nt<3072>(ring[li][3], lane16, a_rsrc); load_a_end(); }
                        if constexpr (i == 4) rs_dma(w_u, w_ord);
#ifndef CODD_I8_EXP_NODMA
                        if constexpr (i >= 5) {
                            constexpr int d = i - 5, which = d / kDmaPerSlice, j = d % kDmaPerSlice;
                            stage_dma_piece((t + 2 + which) & 3, j, j == kDmaPerSlice - 1);
                        }
#endif
                    }
                });
            } else {
            if (CODD_I8_EARLY_FRAGS) frag_prefetch(b, qaddr);
            __builtin_amdgcn_sched_barrier(0);
            rs_dma(w_u, w_ord);
#ifndef CODD_I8_EXP_NODMA
            if constexpr (!RES) stage_dma((t + 2) & 3);
#endif
            load_a(ring[li]);
#ifdef CODD_I8_EXP_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            T1 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
#endif
            i8_wait_vm<2 * kOpsPerIv>(ring[ci][0], ring[ci][1], ring[ci][2], ring[ci][3]);
            __builtin_amdgcn_sched_barrier(0);
#ifdef CODD_I8_EXP_STAMPS
            T2 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (!CODD_I8_EARLY_FRAGS) frag_prefetch(b, qaddr);
            mfma_step(std::integral_constant<int, ci>{}, FIRST, FUSED, b, qaddr, ectx, [](auto) {});
            }
#ifdef CODD_I8_EXP_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long T3 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int s = t - LAG;
            if (s >= 0) {
                if (s < T && c_s == nsteps - 1) { pending = true; p_u = c_u; p_ord = c_ord; }
                if (++c_s == nsteps) { c_s = 0; c_u += G; ++c_ord; }
            }
#ifdef CODD_I8_EXP_NOBARRIER
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kAPerIv + kOpsPerIv) : "memory");  // diagnostic (racy)
#else
            // (the resident program: one barrier per tile, in its second interval — slices are read-only, corpus fragments
            // and row scales belong to the wave, the hit lists / sample keys are double-buffered by tile parity)
            if constexpr (sync == 2) {
                // the slices of steps t + 1 (requested one interval ago) and t + 2 (requested in this one) have landed; younger
                // than this interval's slice DMA are only its corpus loads
                // (spread issue: the odd interval's own five operations — corpus loads, scale DMA — are what is younger than the
                // even interval's slice DMA)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CODD_I8_SPREAD_VM ? 5 : kAPerIv) : "memory");
            } else if constexpr (sync == 0) {
                if (RES && w_s != 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kAPerIv + kOpsPerIv) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kAPerIv + kOpsPerIv) : "memory");
            }
#endif
#ifdef CODD_I8_EXP_STAMPS
            {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long T4 = __builtin_readcyclecounter();
                st_pre += T1 - T0; st_vm += T2 - T1; st_mfma += T3 - T2; st_sync += T4 - T3; st_iv += 1;
            }
#endif
            // every wave has folded tile w_ord - 1 when the barrier of the interval with w_s == 1 releases, and no wave
            // writes that tile's half of the shared state again before the barrier inside the next tile (staged program:
            // no epilogue runs in the next interval, nsteps >= 3): the counters read here are stable
            if (decltype(BOOK)::value && w_s == 1 && w_ord >= 1) {
                const unsigned par = kLists == 2 ? (unsigned)((w_ord - 1) & 1) : 0u;
                if (MODE == MODE_FILTER) {
                    const unsigned cnt = lds_w[256 + par];
                    if (cnt > (kLists == 2 ? kListCap / 2 : (unsigned)(CODD_FLUSH_AT))) {
                        // (per-query ranges reserved with one global atomic each: on clustered corpora a tile fills the list
                        // and every workgroup flushes every tile; one atomic per hit serialises on 256 counters)
                        int tid_f = tid;
                        asm volatile("" : "+v"(tid_f));  // (or hipcc hoists &hit_cnt[tid] out of the tile loop and parks the pointer in scratch)
                        flush_hits_binned(lds_hits + par * kListCap * 3, cnt < kListCap ? cnt : kListCap, tid_f, lds_w + 832, hits, hit_cnt, cap_q, reinterpret_cast<const float*>(lds_w + 320));
                        __syncthreads();
                        if (tid == 0) lds_w[256 + par] = 0u;
                    }
                } else if (pub_ord == w_ord - 1 && pub_ord < my_tiles) {
                    if (tid < 256) {
                        u64 key = lds_k[par * 256 + tid];
                        if (key) key = make_key(key_score(key) * qscale[tid], key_row(key));  // the fold ran on acc * rscale
                        bucket_key[(int64_t)tid * ntiles_run + (first_u + pub_ord * G)] = key;
                        lds_k[par * 256 + tid] = 0ull;
                    }
                    ++pub_ord;
                }
            }
            if (++w_s == nsteps) { w_s = 0; w_u += G; ++w_ord; }
        };
        if constexpr (LAG == 0 && !CODD_I8_LAG && STEPS3) {
            // rows whose K-steps are a multiple of the ring length (768 elements: 6): tiles start at ring phase 0, so the
            // epilogue sits BETWEEN the unrolled intervals and exists once in the program instead of three times (the
            // inlined copies, each 19 KB of code run once per tile, kept missing the instruction cache)
            int t = 0;
            constexpr bool kFuse = CODD_I8_FUSE_EPI && MODE == MODE_FILTER;
            const std::integral_constant<bool, kFuse> fuse{};
            const std::false_type no{};
            for (int o = 0; o < my_tiles; ++o) {
                EpiCtx ectx;
                if constexpr (kFuse) {
                    // tile o - 1 (this workgroup's previous one) is tested inside the first K-step below; the first tile has
                    // nothing behind it: the same code runs on NaN scales, which fail every comparison
                    ectx = epi_begin(pending ? p_u : c_u, pending ? p_ord : 0, pending);
                    pending = false;
                } else if (pending) {
#ifdef CODD_I8_EXP_STAMPS
                    const unsigned long long E0 = __builtin_readcyclecounter();
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    epilogue(p_u, p_ord);
                    pending = false;
#ifdef CODD_I8_EXP_STAMPS
                    __builtin_amdgcn_sched_barrier(0);
                    st_epi += __builtin_readcyclecounter() - E0;
#endif
                }
                // the tile's first K-step starts its accumulators from zero: no clearing pass in the epilogue
                const std::true_type yes{};
                const std::integral_constant<int, 0> i0{};
                const std::integral_constant<int, 1> i1{};
                const std::integral_constant<int, 2> i2{};
                if constexpr (kPair) {
                    // intervals 0..nsteps-1 of a tile: the odd ones end in the barrier (nsteps is a multiple of 6, so every tile
                    // starts on an even interval)
                    interval(i0, no, yes, fuse, no, i1, t, &ectx);
                    interval(i1, no, no, no, yes, i2, t + 1, nullptr);   // (w_s == 1: the bookkeeping, behind the barrier)
                    interval(i2, no, no, no, no, i1, t + 2, nullptr);
                    t += 3;
                    for (int s3 = 3; s3 < nsteps; s3 += 6) {
                        interval(i0, no, no, no, no, i2, t, nullptr);
                        interval(i1, no, no, no, no, i1, t + 1, nullptr);
                        interval(i2, no, no, no, no, i2, t + 2, nullptr);
                        t += 3;
                        if (s3 + 3 < nsteps) {
                            interval(i0, no, no, no, no, i1, t, nullptr);
                            interval(i1, no, no, no, no, i2, t + 1, nullptr);
                            interval(i2, no, no, no, no, i1, t + 2, nullptr);
                            t += 3;
                        }
                    }
                } else {
                    interval(i0, no, yes, fuse, no, i0, t, &ectx);
                    interval(i1, no, no, no, yes, i0, t + 1, nullptr);   // (w_s == 1: the bookkeeping)
                    interval(i2, no, no, no, no, i0, t + 2, nullptr);
                    t += 3;
                    for (int s3 = 3; s3 < nsteps; s3 += 3) {
                        interval(i0, no, no, no, no, i0, t, nullptr);
                        interval(i1, no, no, no, no, i0, t + 1, nullptr);
                        interval(i2, no, no, no, no, i0, t + 2, nullptr);
                        t += 3;
                    }
                }
            }
        } else {
            const std::false_type no{};
            for (int t0 = 0; t0 < TI; t0 += 3) {
                const std::integral_constant<int, 0> s0{};
                interval(std::integral_constant<int, 0>{}, std::true_type{}, no, no, std::true_type{}, s0, t0, nullptr);
                interval(std::integral_constant<int, 1>{}, std::true_type{}, no, no, std::true_type{}, s0, t0 + 1, nullptr);
                interval(std::integral_constant<int, 2>{}, std::true_type{}, no, no, std::true_type{}, s0, t0 + 2, nullptr);
            }
        }
        // The corpus loads of the last two intervals are never consumed: hipcc considers their destination registers free
        // from here on and hands them to the code below, while the loads are still in flight and will overwrite them.
        // Nothing of this wave may be in flight past this point.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CODD_I8_EXP_STAMPS
        if (MODE == MODE_FILTER && bucket_key && lane == 0) {
            // (bucket_key is unused by the filter pass: the stamps build's host passes a scratch buffer there; read back with
            // codd_knn_exp_read_stamps)
            unsigned long long* d = reinterpret_cast<unsigned long long*>(bucket_key) + ((int64_t)blockIdx.x * 8 + wave) * 8;
            d[0] = st_pre; d[1] = st_vm; d[2] = st_mfma; d[3] = st_sync; d[4] = st_epi; d[5] = st_iv;
            d[6] = __builtin_readcyclecounter() - st_begin; d[7] = (unsigned long long)my_tiles;
        }
#endif
        if (pending) epilogue(p_u, p_ord);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == MODE_SAMPLE && pub_ord < my_tiles && tid < 256) {
            u64 key = lds_k[(kLists == 2 ? (pub_ord & 1) * 256 : 0) + tid];
            if (key) key = make_key(key_score(key) * qscale[tid], key_row(key));
            bucket_key[(int64_t)tid * ntiles_run + (first_u + pub_ord * G)] = key;
        }
    };
    if (CODD_I8_LAG && wave >= 4) run(std::integral_constant<int, CODD_I8_LAG>{});
    else run(std::integral_constant<int, 0>{});

    if (MODE == MODE_FILTER) {
#pragma unroll
        for (int p = 0; p < kLists; ++p) {
            const unsigned cnt = lds_w[256 + p];
            if (cnt > kListCap && tid == 0) atomicOr(&flags[FLAG_WG_OVERFLOW], 1u);  // statistics only
            __syncthreads();  // everyone has read cnt (lds_w[0..255] is about to be reused as the flush's scratch)
#ifndef CODD_I8_EXP_NOFLUSH  // (diagnostic: the hits of the last tiles are dropped)
            flush_hits_binned(lds_hits + p * kListCap * 3, cnt < kListCap ? cnt : kListCap, tid, lds_w, hits, hit_cnt, cap_q, reinterpret_cast<const float*>(lds_w + 320));
#endif
            if (p + 1 < kLists) __syncthreads();  // (the scratch is reused by the other half's flush)
        }
    }
}

}  // namespace codd
