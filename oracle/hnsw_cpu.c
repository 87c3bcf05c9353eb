/*
 * hnsw_cpu.c — a from-scratch CPU HNSW, TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the package).
 *
 * Why it exists: the reference's k-NN is ChromaDB's, i.e. an HNSW graph (hnswlib) configured at
 * /root/reference/codd_dal/metrics/metrics_semantic_metadata_store.py:63-68 with
 *     "hnsw:space": "cosine", "hnsw:construction_ef": 200, "hnsw:search_ef": 100, "hnsw:M": 16.
 * chromadb / hnswlib are not installed here and cannot be (no network), so bench.py's cpu_baseline leg cannot time
 * the reference's own index.  This file restates the PUBLISHED algorithm (Malkov & Yashunin, "Efficient and robust
 * approximate nearest neighbor search using Hierarchical Navigable Small World graphs", Alg. 1-5, with hnswlib's
 * choices: level multiplier 1/ln(M), M0 = 2M links on layer 0, neighbour selection by the heuristic of Alg. 4 without
 * extendCandidates / keepPruned, cosine space = inner product of L2-normalised vectors, distance 1 - <q, c>) with the
 * reference's parameters, so that the baseline is like for like in ALGORITHM: approximate graph search on the host's
 * cores, reported with its recall@10 against the exact answer next to the exact brute-force port.
 * It is not hnswlib's code and makes no claim about hnswlib's constant factors ("parity unpinned" for speed).
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp)      Binding: oracle/hnsw_cpu.py (ctypes)
 */
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float dist;  /* 1 - <q, c>: smaller = closer */
    int32_t id;
} cand_t;

typedef struct {
    int n, dim, M, M0, efc;
    const float* data;  /* [n][dim], L2-normalised, owned by the caller */
    int* level;         /* [n] */
    int32_t* link0;     /* [n][M0 + 1]: count, then neighbours (layer 0) */
    int32_t** linkup;   /* [n] -> [level][M + 1] or NULL */
    omp_lock_t* lock;   /* [n] */
    omp_lock_t global;
    int entry, maxlevel;
} hnsw_t;

static inline float dist_ip(const float* a, const float* b, int d) {
    /* AVX2 + FMA, four independent accumulators (the Makefile builds with -mavx2 -mfma, as for the exact port) */
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    int i = 0;
    for (; i + 32 <= d; i += 32) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24), s3);
    }
    for (; i + 8 <= d; i += 8) s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
    s0 = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
    __m128 lo = _mm_add_ps(_mm256_castps256_ps128(s0), _mm256_extractf128_ps(s0, 1));
    lo = _mm_add_ps(lo, _mm_movehl_ps(lo, lo));
    lo = _mm_add_ss(lo, _mm_shuffle_ps(lo, lo, 1));
    float s = _mm_cvtss_f32(lo);
    for (; i < d; ++i) s += a[i] * b[i];
    return 1.0f - s;
}

/* ---- binary heaps of cand_t ------------------------------------------------------------------------------------ */
typedef struct {
    cand_t* a;
    int n, cap;
    int maxheap; /* 1: largest dist on top (result set), 0: smallest on top (candidate queue) */
} heap_t;

static void heap_init(heap_t* h, int cap, int maxheap) {
    h->a = (cand_t*)malloc((size_t)cap * sizeof(cand_t));
    h->n = 0;
    h->cap = cap;
    h->maxheap = maxheap;
}
static inline int heap_before(const heap_t* h, cand_t x, cand_t y) { return h->maxheap ? x.dist > y.dist : x.dist < y.dist; }
static void heap_push(heap_t* h, cand_t c) {
    if (h->n == h->cap) {
        h->cap *= 2;
        h->a = (cand_t*)realloc(h->a, (size_t)h->cap * sizeof(cand_t));
    }
    int i = h->n++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!heap_before(h, c, h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = c;
}
static cand_t heap_pop(heap_t* h) {
    cand_t top = h->a[0], last = h->a[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = -1;
        if (l >= h->n) break;
        m = (r < h->n && heap_before(h, h->a[r], h->a[l])) ? r : l;
        if (!heap_before(h, h->a[m], last)) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (h->n > 0) h->a[i] = last;
    return top;
}

/* per-thread scratch: visited epochs and two heaps */
typedef struct {
    uint32_t* seen;
    uint32_t epoch;
    heap_t cands, top;
    cand_t* tmp;
    int tmp_cap;
} scratch_t;

static void scratch_init(scratch_t* s, int n, int ef) {
    s->seen = (uint32_t*)calloc((size_t)n, sizeof(uint32_t));
    s->epoch = 0;
    heap_init(&s->cands, 4 * ef + 64, 0);
    heap_init(&s->top, ef + 64, 1);
    s->tmp_cap = ef + 64;
    s->tmp = (cand_t*)malloc((size_t)s->tmp_cap * sizeof(cand_t));
}
static void scratch_free(scratch_t* s) {
    free(s->seen);
    free(s->cands.a);
    free(s->top.a);
    free(s->tmp);
}

static inline int32_t* links_of(const hnsw_t* h, int id, int lvl) { return lvl == 0 ? h->link0 + (size_t)id * (h->M0 + 1) : h->linkup[id] + (size_t)(lvl - 1) * (h->M + 1); }

/* Alg. 2 SEARCH-LAYER: ef closest to q on layer lvl starting from ep; result left in s->top (max-heap). locked: take node locks */
static void search_layer(const hnsw_t* h, const float* q, int ep, float ep_dist, int ef, int lvl, scratch_t* s, int locked) {
    if (++s->epoch == 0) {
        memset(s->seen, 0, (size_t)h->n * sizeof(uint32_t));
        s->epoch = 1;
    }
    s->cands.n = 0;
    s->top.n = 0;
    cand_t e = {ep_dist, ep};
    heap_push(&s->cands, e);
    heap_push(&s->top, e);
    s->seen[ep] = s->epoch;
    int32_t nb[512];
    while (s->cands.n > 0) {
        cand_t c = heap_pop(&s->cands);
        if (c.dist > s->top.a[0].dist && s->top.n >= ef) break;
        int cnt;
        if (locked) omp_set_lock(&((hnsw_t*)h)->lock[c.id]);
        const int32_t* l = links_of(h, c.id, lvl);
        cnt = l[0];
        memcpy(nb, l + 1, (size_t)cnt * sizeof(int32_t));
        if (locked) omp_unset_lock(&((hnsw_t*)h)->lock[c.id]);
        for (int i = 0; i < cnt; ++i) {
            const int v = nb[i];
            if (s->seen[v] == s->epoch) continue;
            s->seen[v] = s->epoch;
            const float dv = dist_ip(q, h->data + (size_t)v * h->dim, h->dim);
            if (s->top.n < ef || dv < s->top.a[0].dist) {
                cand_t nv = {dv, v};
                heap_push(&s->cands, nv);
                heap_push(&s->top, nv);
                if (s->top.n > ef) heap_pop(&s->top);
            }
        }
    }
}

static int cmp_cand(const void* a, const void* b) {
    const cand_t *x = (const cand_t*)a, *y = (const cand_t*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/* Alg. 4 SELECT-NEIGHBORS-HEURISTIC on `c` (sorted ascending by distance to the base point): keep a candidate only if it is
 * closer to the base than to every neighbour kept so far.  Returns how many were kept (written to the front of c). */
static int select_heuristic(const hnsw_t* h, cand_t* c, int n, int M) {
    if (n <= M) return n;
    int kept = 0;
    for (int i = 0; i < n && kept < M; ++i) {
        int good = 1;
        for (int j = 0; j < kept; ++j) {
            const float dij = dist_ip(h->data + (size_t)c[i].id * h->dim, h->data + (size_t)c[j].id * h->dim, h->dim);
            if (dij < c[i].dist) {
                good = 0;
                break;
            }
        }
        if (good) c[kept++] = c[i];
    }
    return kept;
}

static void connect(hnsw_t* h, int id, int lvl, cand_t* sel, int nsel, scratch_t* s) {
    const int cap = lvl == 0 ? h->M0 : h->M;
    omp_set_lock(&h->lock[id]);
    int32_t* mine = links_of(h, id, lvl);
    mine[0] = nsel;
    for (int i = 0; i < nsel; ++i) mine[1 + i] = sel[i].id;
    omp_unset_lock(&h->lock[id]);
    for (int i = 0; i < nsel; ++i) {
        const int v = sel[i].id;
        omp_set_lock(&h->lock[v]);
        int32_t* l = links_of(h, v, lvl);
        int cnt = l[0], dup = 0;
        for (int j = 0; j < cnt; ++j) dup |= l[1 + j] == id;
        if (!dup) {
            if (cnt < cap) {
                l[1 + cnt] = id;
                l[0] = cnt + 1;
            } else {
                /* shrink: the current links plus the new one, re-selected by the heuristic around v */
                cand_t* t = s->tmp;
                const float* pv = h->data + (size_t)v * h->dim;
                for (int j = 0; j < cnt; ++j) {
                    t[j].id = l[1 + j];
                    t[j].dist = dist_ip(pv, h->data + (size_t)l[1 + j] * h->dim, h->dim);
                }
                t[cnt].id = id;
                t[cnt].dist = sel[i].dist;
                qsort(t, (size_t)cnt + 1, sizeof(cand_t), cmp_cand);
                const int k = select_heuristic(h, t, cnt + 1, cap);
                l[0] = k;
                for (int j = 0; j < k; ++j) l[1 + j] = t[j].id;
            }
        }
        omp_unset_lock(&h->lock[v]);
    }
}

static void insert(hnsw_t* h, int id, scratch_t* s) {
    const float* q = h->data + (size_t)id * h->dim;
    const int lvl = h->level[id];
    omp_set_lock(&h->global);
    int ep = h->entry, top = h->maxlevel;
    if (ep < 0) {
        h->entry = id;
        h->maxlevel = lvl;
        omp_unset_lock(&h->global);
        return;
    }
    const int holds_global = lvl > top;  /* a new top level is installed under the global lock, as hnswlib does */
    if (!holds_global) omp_unset_lock(&h->global);
    float d = dist_ip(q, h->data + (size_t)ep * h->dim, h->dim);
    for (int l = top; l > lvl; --l) {  /* greedy descent, ef = 1 */
        int changed = 1;
        while (changed) {
            changed = 0;
            omp_set_lock(&h->lock[ep]);
            const int32_t* ln = links_of(h, ep, l);
            int32_t nb[512];
            const int cnt = ln[0];
            memcpy(nb, ln + 1, (size_t)cnt * sizeof(int32_t));
            omp_unset_lock(&h->lock[ep]);
            for (int i = 0; i < cnt; ++i) {
                const float dv = dist_ip(q, h->data + (size_t)nb[i] * h->dim, h->dim);
                if (dv < d) {
                    d = dv;
                    ep = nb[i];
                    changed = 1;
                }
            }
        }
    }
    for (int l = lvl < top ? lvl : top; l >= 0; --l) {
        search_layer(h, q, ep, d, h->efc, l, s, 1);
        int n = s->top.n;
        if (n > s->tmp_cap) n = s->tmp_cap;
        cand_t* c = (cand_t*)malloc((size_t)n * sizeof(cand_t));
        memcpy(c, s->top.a, (size_t)n * sizeof(cand_t));
        qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
        ep = c[0].id;
        d = c[0].dist;
        const int k = select_heuristic(h, c, n, h->M);
        connect(h, id, l, c, k, s);
        free(c);
    }
    if (holds_global) {
        h->entry = id;
        h->maxlevel = lvl;
        omp_unset_lock(&h->global);
    }
}

/* ---- C ABI ------------------------------------------------------------------------------------------------------ */
void* hnsw_build(const float* data, int n, int dim, int M, int ef_construction, uint64_t seed, int threads) {
    if (n < 1 || dim < 1 || M < 2 || M > 128) return NULL;
    hnsw_t* h = (hnsw_t*)calloc(1, sizeof(hnsw_t));
    h->n = n;
    h->dim = dim;
    h->M = M;
    h->M0 = 2 * M;
    h->efc = ef_construction;
    h->data = data;
    h->entry = -1;
    h->maxlevel = -1;
    h->level = (int*)malloc((size_t)n * sizeof(int));
    h->link0 = (int32_t*)calloc((size_t)n * (h->M0 + 1), sizeof(int32_t));
    h->linkup = (int32_t**)calloc((size_t)n, sizeof(int32_t*));
    h->lock = (omp_lock_t*)malloc((size_t)n * sizeof(omp_lock_t));
    omp_init_lock(&h->global);
    const double mult = 1.0 / log((double)M);
    uint64_t x = seed ? seed : 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < n; ++i) {
        omp_init_lock(&h->lock[i]);
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;  /* xorshift64: deterministic levels */
        const double u = ((double)(x >> 11) + 1.0) / 9007199254740993.0;
        int l = (int)(-log(u) * mult);
        if (l > 30) l = 30;
        h->level[i] = l;
        if (l > 0) h->linkup[i] = (int32_t*)calloc((size_t)l * (M + 1), sizeof(int32_t));
    }
    if (threads < 1) threads = omp_get_max_threads();
    /* the first rows go in serially (a sane top of the hierarchy), the rest in parallel */
    const int serial = n < 256 ? n : 256;
    {
        scratch_t s;
        scratch_init(&s, n, ef_construction);
        for (int i = 0; i < serial; ++i) insert(h, i, &s);
        scratch_free(&s);
    }
#pragma omp parallel num_threads(threads)
    {
        scratch_t s;
        scratch_init(&s, n, ef_construction);
#pragma omp for schedule(dynamic, 64)
        for (int i = serial; i < n; ++i) insert(h, i, &s);
        scratch_free(&s);
    }
    return h;
}

/* out_ids / out_dist: [nq][k], ascending distance, -1 / +inf padded */
void hnsw_search(void* handle, const float* queries, int nq, int k, int ef_search, int32_t* out_ids, float* out_dist, int threads) {
    hnsw_t* h = (hnsw_t*)handle;
    const int ef = ef_search > k ? ef_search : k;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        scratch_t s;
        scratch_init(&s, h->n, ef);
#pragma omp for schedule(dynamic, 4)
        for (int qi = 0; qi < nq; ++qi) {
            const float* q = queries + (size_t)qi * h->dim;
            int ep = h->entry;
            float d = dist_ip(q, h->data + (size_t)ep * h->dim, h->dim);
            for (int l = h->maxlevel; l > 0; --l) {
                int changed = 1;
                while (changed) {
                    changed = 0;
                    const int32_t* ln = links_of(h, ep, l);
                    for (int i = 0; i < ln[0]; ++i) {
                        const float dv = dist_ip(q, h->data + (size_t)ln[1 + i] * h->dim, h->dim);
                        if (dv < d) {
                            d = dv;
                            ep = ln[1 + i];
                            changed = 1;
                        }
                    }
                }
            }
            search_layer(h, q, ep, d, ef, 0, &s, 0);
            int n = s.top.n;
            cand_t* c = (cand_t*)malloc((size_t)n * sizeof(cand_t));
            memcpy(c, s.top.a, (size_t)n * sizeof(cand_t));
            qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
            for (int j = 0; j < k; ++j) {
                out_ids[(size_t)qi * k + j] = j < n ? c[j].id : -1;
                out_dist[(size_t)qi * k + j] = j < n ? c[j].dist : INFINITY;
            }
            free(c);
        }
        scratch_free(&s);
    }
}

void hnsw_free(void* handle) {
    hnsw_t* h = (hnsw_t*)handle;
    if (!h) return;
    for (int i = 0; i < h->n; ++i) {
        omp_destroy_lock(&h->lock[i]);
        free(h->linkup[i]);
    }
    omp_destroy_lock(&h->global);
    free(h->level);
    free(h->link0);
    free(h->linkup);
    free(h->lock);
    free(h);
}

int hnsw_max_level(void* handle) { return ((hnsw_t*)handle)->maxlevel; }
