/*
 * oracle/knn_oracle.c — CPU restatement of the cosine top-k behind
 * `collection.query(...)` on Codd's search_relevant_metrics path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under codd_query_engine_amd/ may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / the timed CPU baseline.
 *
 * PARITY: UNPINNED against ChromaDB.  The arithmetic the reference runs lives in
 * the third-party `chromadb` package (unpinned: `>=0.4.0` in
 * /root/reference/codd_dal/pyproject.toml:11, `>=1.4.0` in
 * /root/reference/codd_lib/pyproject.toml:11, no lockfile; server image
 * chromadb/chroma:latest, /root/reference/docker-compose.yml:4), which is not
 * installed here and whose default embedder is fetched from the network.  The
 * reference's own tests hold no numeric vectors for this call.  What this file
 * restates is therefore the *contract at the reference's call sites*:
 *
 *   - metric = cosine            codd_dal/metrics/metrics_semantic_metadata_store.py:63-68
 *   - distance = 1 - similarity  codd_dal/metrics/metrics_semantic_metadata_store.py:336
 *   - ascending distance, one list per query, min(n, count) hits
 *                                codd_dal/metrics/metrics_semantic_metadata_store.py:314-329
 *   - n_results <= 100           codd_dal/metrics/metrics_semantic_metadata_store.py:24,308-312
 *
 * plus the published behaviour of the HNSW cosine space ChromaDB wraps
 * (hnswlib: rows are L2-normalised on insert and on query, distance =
 * 1 - <q,c> in fp32) — searched EXHAUSTIVELY here, i.e. the exact answer HNSW
 * approximates.  Ties resolve to the lower row index (build-defined; HNSW's
 * order is heap-dependent).
 *
 * To make "ids bit-exact" testable, the fp32 score is defined with a fixed
 * evaluation order (the CANONICAL SCORE): the row is cut into
 * 16-byte chunks (E = 4 fp32 / 8 bf16|fp16 elements); chunk j belongs to lane
 * j mod 64; every lane runs one fmaf chain over its chunks in increasing j and
 * over the E elements of a chunk in increasing order, starting from +0; the 64
 * lane sums are combined by a butterfly with strides 32,16,8,4,2,1 (IEEE add,
 * so every lane ends with the same value); finally +0.0f canonicalises -0.
 * The HIP kernels evaluate exactly this expression, so scores — not only ids —
 * are bit-identical between this file and the GPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 64

enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

/* ---------- scalar format helpers ---------- */

static inline float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u32_as_f32((uint32_t)h << 16); }

/* round-to-nearest-even, NaN stays NaN (what v_cvt_pk_bf16_f32 does) */
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_as_u32(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u32_as_f32(sign);
        /* subnormal: man * 2^-24 */
        float v = (float)man * (1.0f / 16777216.0f);
        return (sign ? -v : v);
    }
    if (exp == 31) return u32_as_f32(sign | 0x7f800000u | (man << 13));
    return u32_as_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* round-to-nearest-even fp32 -> fp16 (what v_cvt_f16_f32 does in RNE mode) */
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f32_as_u32(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t au = u & 0x7fffffffu;
    if (au > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);             /* NaN */
    if (au >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            /* >= 65520 -> inf */
    if (au < 0x33000001u) return (uint16_t)sign;                         /* < 2^-25 (+ulp) -> 0 */
    int32_t e = (int32_t)(au >> 23) - 127;
    uint32_t man = (au & 0x7fffffu) | 0x800000u;
    if (e < -14) {                                                       /* subnormal result */
        int shift = -14 - e + 13;                                        /* 14..24 */
        uint32_t r = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((man >> 13) & 0x3ffu);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;              /* may carry into exp: fine */
    return (uint16_t)(sign | r);
}

/* order-preserving map fp32 -> u32 (NaN is ranked as -inf) */
static inline uint32_t ord_f32(float s) {
    if (s != s) s = -INFINITY;
    uint32_t u = f32_as_u32(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unord_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return u32_as_f32(u);
}
/* packed key: larger = better (higher score, then LOWER row) */
static inline uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)ord_f32(score) << 32) | (uint64_t)(0xffffffffu - row);
}

int oracle_elems_per_chunk(int dtype) { return dtype == DT_F32 ? 4 : 8; }

void oracle_f32_to_bf16(const float* in, uint16_t* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = f32_to_bf16(in[i]);
}
void oracle_bf16_to_f32(const uint16_t* in, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = bf16_to_f32(in[i]);
}
void oracle_f32_to_f16(const float* in, uint16_t* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = f32_to_f16(in[i]);
}
void oracle_f16_to_f32(const uint16_t* in, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = f16_to_f32(in[i]);
}

/* ---------- the canonical score ---------- */

static inline float butterfly64(float* acc) {
    float t[LANES];
    for (int stride = 32; stride >= 1; stride >>= 1) {
        for (int l = 0; l < LANES; ++l) t[l] = acc[l] + acc[l ^ stride];
        memcpy(acc, t, sizeof(t));
    }
    return acc[0] + 0.0f;
}

/* q: fp32[dpad]; c: fp32[dpad] (already widened for 16-bit rows — widening is exact).
 * E = elements per 16-byte chunk of the STORED row (4 for fp32 rows, 8 for bf16/fp16). */
static float canon_dot(const float* q, const float* c, int dpad, int E) {
    float acc[LANES];
    for (int l = 0; l < LANES; ++l) acc[l] = 0.0f;
    const int nchunks = dpad / E;
    for (int j = 0; j < nchunks; ++j) {
        const int l = j & (LANES - 1);
        float a = acc[l];
        for (int e = 0; e < E; ++e) a = fmaf(q[j * E + e], c[j * E + e], a);
        acc[l] = a;
    }
    return butterfly64(acc);
}

float oracle_canon_dot(const float* q, const float* c, int dpad, int E) { return canon_dot(q, c, dpad, E); }

/* Row normalisation as the ingest kernel does it: n2 = canon_dot(x,x) with E=4
 * (the ingest input is always fp32), inv-free IEEE division by sqrtf(n2);
 * a zero (or non-finite-norm) row is stored as zeros -> score 0 -> distance 1. */
void oracle_normalize_rows(const float* in, float* out, int64_t n, int d, int dpad) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float* o = out + r * (int64_t)dpad;
        const float* x = in + r * (int64_t)d;
        for (int i = 0; i < dpad; ++i) o[i] = i < d ? x[i] : 0.0f;
        const float n2 = canon_dot(o, o, dpad, 4);
        const float nrm = sqrtf(n2);
        if (!(nrm > 0.0f) || !(nrm < INFINITY)) {
            for (int i = 0; i < dpad; ++i) o[i] = 0.0f;
        } else {
            for (int i = 0; i < dpad; ++i) o[i] = o[i] / nrm;
        }
    }
}

/* ---------- bounded top-k list of keys (descending) ---------- */

static inline void topk_insert(uint64_t* list, int k, uint64_t key) {
    if (key <= list[k - 1]) return;
    int i = k - 1;
    while (i > 0 && list[i - 1] < key) { list[i] = list[i - 1]; --i; }
    list[i] = key;
}

static void widen_row(const void* rows, int dtype, int64_t r, int dpad, float* out) {
    if (dtype == DT_F32) {
        memcpy(out, (const float*)rows + r * dpad, sizeof(float) * (size_t)dpad);
    } else if (dtype == DT_BF16) {
        const uint16_t* p = (const uint16_t*)rows + r * dpad;
        for (int i = 0; i < dpad; ++i) out[i] = bf16_to_f32(p[i]);
    } else {
        const uint16_t* p = (const uint16_t*)rows + r * dpad;
        for (int i = 0; i < dpad; ++i) out[i] = f16_to_f32(p[i]);
    }
}

/*
 * Exhaustive canonical-score top-k.
 *   rows     : n x dpad stored rows (dtype), already normalised (the index's storage)
 *   queries  : B x dpad fp32, already normalised
 *   out_keys : B x k packed keys, descending, 0 = empty slot
 *   row_base : added to the row index inside the key (global row of a shard)
 */
int oracle_search_keys(const void* rows, int dtype, int64_t n, int dpad,
                       const float* queries, int B, int k, uint32_t row_base,
                       uint64_t* out_keys) {
    if (k < 1 || B < 1 || dpad % 64 != 0) return -1;
    const int E = oracle_elems_per_chunk(dtype);
    memset(out_keys, 0, sizeof(uint64_t) * (size_t)B * (size_t)k);
#pragma omp parallel
    {
        uint64_t* local = (uint64_t*)calloc((size_t)B * (size_t)k, sizeof(uint64_t));
        float* wide = (float*)malloc(sizeof(float) * (size_t)dpad);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            widen_row(rows, dtype, r, dpad, wide);
            for (int b = 0; b < B; ++b) {
                const float s = canon_dot(queries + (int64_t)b * dpad, wide, dpad, E);
                topk_insert(local + (int64_t)b * k, k, make_key(s, row_base + (uint32_t)r));
            }
        }
#pragma omp critical
        {
            for (int b = 0; b < B; ++b)
                for (int i = 0; i < k; ++i)
                    if (local[(int64_t)b * k + i]) topk_insert(out_keys + (int64_t)b * k, k, local[(int64_t)b * k + i]);
        }
        free(local);
        free(wide);
    }
    return 0;
}

/* keys -> (distance = 1 - score in fp32, row or -1) */
void oracle_unpack_keys(const uint64_t* keys, int64_t count, float* out_dist, int64_t* out_rows) {
    for (int64_t i = 0; i < count; ++i) {
        if (keys[i] == 0) { out_dist[i] = INFINITY; out_rows[i] = -1; continue; }
        const float s = unord_f32((uint32_t)(keys[i] >> 32));
        out_dist[i] = 1.0f - s;
        out_rows[i] = (int64_t)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu));
    }
}

int oracle_search(const void* rows, int dtype, int64_t n, int dpad,
                  const float* queries, int B, int k,
                  float* out_dist, int64_t* out_rows) {
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)B * (size_t)k);
    int rc = oracle_search_keys(rows, dtype, n, dpad, queries, B, k, 0u, keys);
    if (rc == 0) oracle_unpack_keys(keys, (int64_t)B * k, out_dist, out_rows);
    free(keys);
    return rc;
}

/* top-k of an arbitrary key multiset (the shard merge): in = B x m, out = B x k */
void oracle_merge_keys(const uint64_t* in, int B, int m, int k, uint64_t* out) {
    memset(out, 0, sizeof(uint64_t) * (size_t)B * (size_t)k);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < m; ++i)
            if (in[(int64_t)b * m + i]) topk_insert(out + (int64_t)b * k, k, in[(int64_t)b * m + i]);
}

/* fp64 scores of one query against all rows (tolerance checks, not ranking) */
void oracle_scores_f64(const void* rows, int dtype, int64_t n, int dpad, const float* q, double* out) {
#pragma omp parallel
    {
        float* wide = (float*)malloc(sizeof(float) * (size_t)dpad);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            widen_row(rows, dtype, r, dpad, wide);
            double s = 0.0;
            for (int i = 0; i < dpad; ++i) s += (double)q[i] * (double)wide[i];
            out[r] = s;
        }
        free(wide);
    }
}

/*
 * Throughput-oriented CPU scan used ONLY as bench.py's cpu_baseline ("port"):
 * same answer up to fp32 summation order (8-way split accumulators that the
 * compiler vectorises), fp32 rows only, all OpenMP threads, row blocks so a
 * block of rows is reused from cache across the B queries.
 */
int oracle_search_fast_f32(const float* rows, int64_t n, int dpad,
                           const float* queries, int B, int k,
                           float* out_dist, int64_t* out_rows) {
    if (k < 1 || B < 1) return -1;
    uint64_t* keys = (uint64_t*)calloc((size_t)B * (size_t)k, sizeof(uint64_t));
    const int64_t RB = 64;
#pragma omp parallel
    {
        uint64_t* local = (uint64_t*)calloc((size_t)B * (size_t)k, sizeof(uint64_t));
#pragma omp for schedule(dynamic, 16)
        for (int64_t r0 = 0; r0 < n; r0 += RB) {
            const int64_t r1 = r0 + RB < n ? r0 + RB : n;
            for (int b = 0; b < B; ++b) {
                const float* q = queries + (int64_t)b * dpad;
                for (int64_t r = r0; r < r1; ++r) {
                    const float* c = rows + r * dpad;
                    float a[16];
                    for (int u = 0; u < 16; ++u) a[u] = 0.0f;
                    for (int i = 0; i < dpad; i += 16)
                        for (int u = 0; u < 16; ++u) a[u] += q[i + u] * c[i + u];
                    float s = 0.0f;
                    for (int u = 0; u < 16; ++u) s += a[u];
                    topk_insert(local + (int64_t)b * k, k, make_key(s, (uint32_t)r));
                }
            }
        }
#pragma omp critical
        {
            for (int64_t i = 0; i < (int64_t)B * k; ++i)
                if (local[i]) topk_insert(keys + (i / k) * k, k, local[i]);
        }
        free(local);
    }
    oracle_unpack_keys(keys, (int64_t)B * k, out_dist, out_rows);
    free(keys);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
