/*
 * codd_knn.h — C ABI of the MI355X-native cosine top-k engine that stands where
 * ChromaDB stands on Codd's `search_relevant_metrics` path.
 *
 * The reference has no FFI of its own: its seam is Python duck typing on the
 * chromadb client object injected into MetricsSemanticMetadataStore
 * (/root/reference/codd_dal/metrics/metrics_semantic_metadata_store.py:43-57).
 * Each entry point below names the chromadb call it takes over; string ids,
 * documents and metadata never cross this boundary (the Python façade
 * codd_query_engine_amd/knn_client.py keeps id <-> row slot and metadata on the
 * host, exactly the part of chromadb that is not arithmetic).
 *
 * Conventions
 *   - every function returns 0 on success and a negative CODD_KNN_E* code on
 *     failure; it never throws, aborts or exits.  codd_knn_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - "dev_" pointers are device (HBM) addresses on the index's GPU, "host_"
 *     pointers are ordinary host memory.  The caller owns every buffer it passes;
 *     the index owns its row storage and its workspaces.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *     are asynchronous with respect to the host unless stated otherwise.
 *   - thread-safety: the search entry points (codd_knn_search, _search_keys,
 *     _ivf_search, _approx_scores) may be called from several host threads and on
 *     several streams of one index: the index keeps one workspace per stream (up to
 *     4; a fifth stream takes over the least recently used one, ordered behind its
 *     previous owner on the device) and serialises only the enqueueing.  Searches on
 *     different streams then overlap on the GPU.  upsert/reserve/load/ivf_install
 *     are exclusive: no other call on the index may be in flight.
 *   - rows are stored L2-normalised, zero padded to a multiple of 64 elements.
 *     score = <q/|q|, c/|c|> evaluated in fp32 in the canonical order of
 *     DESIGN.md §3; distance = 1 - score (fp32); ties -> lower row.
 */
#ifndef CODD_KNN_H
#define CODD_KNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CODD_KNN_DTYPE_F32 0
#define CODD_KNN_DTYPE_BF16 1
#define CODD_KNN_DTYPE_F16 2

/* `hnsw:space` of get_or_create_collection (store.py:63-68); only cosine exists on the path */
#define CODD_KNN_METRIC_COSINE 0

#define CODD_KNN_MAX_K 128        /* reference caps n_results at 100 (store.py:24) */
#define CODD_KNN_MAX_BATCH 1024   /* queries per codd_knn_search call */

#define CODD_KNN_OK 0
#define CODD_KNN_EINVAL (-22)     /* bad argument */
#define CODD_KNN_ENOMEM (-12)     /* device allocation failed */
#define CODD_KNN_EDEVICE (-5)     /* HIP runtime error */
#define CODD_KNN_ENOTSUP (-95)    /* shape/dtype outside what the kernels cover */

typedef struct codd_knn_index codd_knn_index;

/* library / build identification: "codd_knn <semver> gfx950 shadow=<bf16|f16> mfma=<shape>" */
const char* codd_knn_version(void);
const char* codd_knn_last_error(void);

/*
 * Replaces: client.get_or_create_collection(name=..., metadata={"hnsw:space":"cosine",...})
 *           (store.py:60-69) — the arithmetic half: an empty device-resident row store.
 * dim: embedding width (<= 4096; f32 rows <= 1024 in this release).
 */
int codd_knn_create(codd_knn_index** out, int device, int dim, int dtype, int metric);
int codd_knn_destroy(codd_knn_index* index);

/* Grow row storage to hold at least `rows` row slots (contents preserved). Synchronous. */
int codd_knn_reserve(codd_knn_index* index, int64_t rows);

/*
 * Replaces: collection.upsert(documents=[...], metadatas=[...], ids=[...]) (store.py:236-238)
 *           — the vector half.  Writes `n` fp32 vectors (row stride `dim`) into the row
 *           slots chosen by the host id-map; a slot may be new (append) or existing
 *           (overwrite).  normalize != 0 scales each vector to unit L2 norm first
 *           (always what the façade asks for; 0 is for callers that already did).
 *           Storage grows as needed.  The host variant stages through a bounded device
 *           buffer (kept by the index between calls) and is synchronous; the device variant
 *           writes slots [first_slot, first_slot+n) asynchronously on `stream`: it is ordered on
 *           the device behind every search and every earlier upsert already enqueued on other
 *           streams of this index (two writers on two streams run in call order), and every later
 *           search, upsert or codd_knn_copy_rows_f32 on another stream waits for it on the device
 *           (no host synchronisation either way).  The `dev_vecs` buffer must stay valid until
 *           `stream` has run the call.
 */
int codd_knn_upsert_host(codd_knn_index* index, const int64_t* host_slots, const float* host_vecs,
                         int64_t n, int normalize);
int codd_knn_upsert_device(codd_knn_index* index, int64_t first_slot, const float* dev_vecs,
                           int64_t n, int normalize, void* stream);

/*
 * Counterpart of codd_knn_read_rows for loading a persisted index (the role of the Chroma
 * server's docker volume, docker-compose.yml:8-9): `host_rows` are n stored rows exactly as
 * codd_knn_read_rows returned them (storage dtype, padded width, already normalised); they are
 * copied into slots [first_slot, first_slot+n).  Their norms are checked on the device: if any row is
 * not a unit vector (within the storage type's rounding) the index behaves as after an upsert with
 * normalize = 0 — both filters off, every search an exact scan ("all_normalized" stat = 0).  Synchronous.
 */
int codd_knn_load_rows(codd_knn_index* index, int64_t first_slot, const void* host_rows, int64_t n);

/* Number of row slots in use (highest written slot + 1) — collection.count(). */
int codd_knn_count(const codd_knn_index* index, int64_t* out);
int codd_knn_dim(const codd_knn_index* index, int* dim, int* padded_dim, int* dtype);

/* Copy stored rows [first, first+n) back to the host, in storage dtype, padded width.
 * (persistence + tests).  Synchronous. */
int codd_knn_read_rows(const codd_knn_index* index, int64_t first, int64_t n, void* host_out);

/*
 * Replaces: collection.query(query_texts=[q], n_results=n) (store.py:314-316) — the k-NN
 *           half, batched.  dev_queries: B x dim fp32, raw (normalised here).
 *           dev_dist : B x k fp32, ascending distance (1 - score), +inf padded.
 *           dev_rows : B x k int64 row slots, -1 padded (fewer than k rows stored).
 */
int codd_knn_search(codd_knn_index* index, const float* dev_queries, int B, int k,
                    float* dev_dist, int64_t* dev_rows, void* stream);

/*
 * Shard-local half of a row-sharded search: same as codd_knn_search but returns packed
 * keys, key = (orderable_u32(score) << 32) | (0xFFFFFFFF - (row_base + row)), descending,
 * 0 = empty slot — so that the cross-rank merge is an integer top-k (codd_knn_merge_keys).
 */
int codd_knn_search_keys(codd_knn_index* index, const float* dev_queries, int B, int k,
                         uint32_t row_base, uint64_t* dev_keys, void* stream);

/*
 * Top-k of B lists of m packed keys each (the all-gathered [B][G*k] shard partials).
 * Any of dev_keys_out / dev_dist / dev_rows may be NULL.  `device` as in codd_knn_create.
 */
int codd_knn_merge_keys(int device, const uint64_t* dev_keys_in, int B, int m, int k,
                        uint64_t* dev_keys_out, float* dev_dist, int64_t* dev_rows, void* stream);

/*
 * The same merge straight from the buffer an all_gather of the ranks' [B][k_in] partials delivers:
 * dev_keys_in[(g * B + q) * k_in + j], g < G.  No transpose pass in between.
 */
int codd_knn_merge_shards(int device, const uint64_t* dev_keys_in, int G, int B, int k_in, int k,
                          uint64_t* dev_keys_out, float* dev_dist, int64_t* dev_rows, void* stream);

/*
 * The raw approximate (bf16 MFMA) scores of B <= 256 queries against every stored row:
 * dev_scores[q * count + row], q < 256 (rows of padding queries are zero).  Used by the IVF build
 * (row -> nearest centroid) and by tests that check the MFMA operand layouts in isolation.
 */
int codd_knn_approx_scores(codd_knn_index* index, const float* dev_queries, int B,
                                 float* dev_scores, void* stream);

/*
 * Coarse-IVF with exact scores (BASELINE config 5: the small-batch, HBM-bound regime on corpora
 * where even one pass over the shard is too slow).  Not on the reference's path — ChromaDB's own
 * index is HNSW (store.py:63-68) — but the same trade: approximate candidate generation, exact
 * distances.  The caller clusters the rows (codd_query_engine_amd/ivf.py: spherical k-means whose
 * assignment GEMM is codd_knn_approx_scores on a centroid index) and hands over
 *   dev_centroids [nlist][dim] fp32, dev_perm [count] (row slots grouped by list),
 *   dev_offsets [nlist+1] (list l = perm[offsets[l] .. offsets[l+1])).
 * install copies the rows into list order (HBM: one more copy of the rows) and builds the coarse
 * index; any later upsert makes the layout stale (ivf_search then fails with EINVAL).
 * ivf_search: nprobe (<= 128) best lists per query by exact centroid score, canonical exact scores
 * over those lists, top-k with ORIGINAL row slots; with nprobe == nlist the result is bit-identical
 * to codd_knn_search.  Any of dev_keys / dev_dist / dev_rows may be NULL.
 */
/* Stored rows [first, first+n) widened to fp32, device to device, asynchronously on `stream` (the IVF build's
 * input).  Ordered like a search: behind every upsert enqueued before it on any stream, and a later upsert waits
 * for it — which is why the index is not const here (ABI change in round 3). */
int codd_knn_copy_rows_f32(codd_knn_index* index, int64_t first, int64_t n, float* dev_out, void* stream);
int codd_knn_ivf_install(codd_knn_index* index, const float* dev_centroids, int nlist,
                         const int64_t* dev_perm, const int64_t* dev_offsets, void* stream);
int codd_knn_ivf_search(codd_knn_index* index, const float* dev_queries, int B, int k, int nprobe,
                        uint32_t row_base, uint64_t* dev_keys, float* dev_dist, int64_t* dev_rows,
                        void* stream);

/*
 * Tuning / introspection (never needed for correctness).
 *   options: "scan_blocks_per_cu" (1..8); "filter" (0/1: MFMA filter path for large batches);
 *            "filter_min_batch" (9), "filter_min_rows" (1: batches >= filter_min_batch always
 *            filter when the corpus has >= 2k sample tiles), "filter_min_rows_small" (100000:
 *            smaller batches filter when rows * B reaches it): when the filter path is taken;
 *            "sample_div" (40: about 1/40 of the tiles set the per-query thresholds, never
 *            fewer than one tile per CU once the corpus has two rounds of tiles), "sample_tiles"
 *            (4096: upper bound on that number), "hit_cap" (131072: per-query
 *            candidate capacity; overflow falls back to the exact scan);
 *            "all_normalized" (write 0 only: the caller knows of stored rows that are not unit
 *            vectors, e.g. a persisted index whose manifest says so: filters off for good);
 *            "shadow8" (1: batches of <= "shadow8_max_batch" (256) queries are filtered through an
 *            int8 copy of the corpus, 1 byte per element, derived lazily from the stored rows and
 *            kept up to date incrementally; results stay exact; an index whose worst row quantises
 *            badly — error norm above 0.04 — keeps the bf16 filter; so does, for the next
 *            "shadow8_cooldown" (256) searches, an index whose int8 passes leave more than
 *            "shadow8_max_surv" (4000) survivors per query or send queries to the fallback: dense
 *            clusters — unless the 2-byte passes are seen to leave at least half as many), "sample_div8" (28:
 *            the int8 filter's thresholds come from a sample of 1/28 of the row tiles) and
 *            "sample_rounds8" (2: ... of at least that many tiles per compute unit for batches of
 *            more than 32 queries, up to a quarter of the corpus; performance only),
 *            "resident_q" (1: rows of <= 512 int8 elements keep the query block in LDS for the
 *            whole launch), "i8v2" (2: batches of 65..256 queries on rows of 384 or
 *            more elements take the second-generation int8 kernel, csrc/filter_i8.h; 1: only rows of
 *            more than 512 elements; 0: never), "i8v2_half" (1: batches of 65..128 queries take that kernel's
 *            8-query-block instantiation; 0: the first-generation kernel),
 *            "i8_pair" (2: that kernel synchronises once per two K-steps on rows of 768 / 1536 ... elements, and rows of exactly 768 take its static form; 1: without the static form; 0: every
 *            K-step), "per_block" (7, bit mask: the int8 bound uses each 32-row block's own quantisation error instead of
 *            the corpus's worst — bit 0 in the tile kernel, bit 1 in finalize (bit 2 is accepted and ignored: the
 *            tile kernel's filter pass always fetches the block metadata by one LDS-DMA per tile); 0 = the device-wide bound everywhere), "fuse_fallback" (1: batches above 64 queries answer candidate-list
 *            overflows inside the finalize launch; 0: a launch of their own), "small_batch_max" (0; 1: a single query on
 *            <= 2M-row int8 shadows is answered by ONE launch — measured slower than the three-launch chain, see
 *            DESIGN.md §12, hence off), "f16_tile" (1: the 2-byte filter of 129..256 queries on rows of 384 / 768 / 1152 ... elements runs
 *            csrc/filter_i8.h's tile program on fp16 operands; 0: the first-generation kernel), "ivf_share" (1: codd_knn_ivf_search scans a probed list once for all queries of the
 *            batch that probe it, from 1,024 (query, list) pairs on; 0: once per pair),
 *            "debug_fail_shadow_alloc" (tests: the next N allocations of the 2-byte shadow fail);
 *            "profile" = N keeps N (start, stop) HIP-event pairs, one per heavy-kernel launch,
 *            recorded on the launch stream (0 = off; resets the log)
 *   stats  : "searches", "scan_launches", "last_scan_blocks", "filter_passes",
 *            "fallback_queries", "filter_hits", "filter_survivors", "capacity_rows",
 *            "device_bytes", "num_cus", "workspaces" (stream workspaces in use), "shadow8_builds", "shadow8_passes", "i8v2_passes",
 *            "shadow16_builds" (the bf16 shadow is built lazily, by the first search that needs it), "all_normalized",
 *            "shadow8_cooldowns", "shadow8_eps_r_micro", "shadow8_wide_blocks" (32-row blocks whose quantisation error is above
 *            0.04: tolerated up to 1 % of the blocks), "shadow16_alloc_failures", "small_batch_passes", "f16_tile_passes", and per kernel K in {scan, filter, sample, finalize}:
 *            "events:K", "time_ns:K" (sum of the recorded launches; syncs on the last event)
 */
int codd_knn_set_option(codd_knn_index* index, const char* key, int64_t value);
int codd_knn_get_stat(const codd_knn_index* index, const char* key, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* CODD_KNN_H */
