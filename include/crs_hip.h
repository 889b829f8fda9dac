/* crs_hip.h -- C ABI of libcrs_hip.so: the MI355X (gfx950) embed -> index -> retrieve hot path.
 *
 * The reference (zahraamselim/compressed-rag-suite) is pure Python; the arithmetic of this path
 * lives in third-party wheels it calls (sentence-transformers / ChromaDB).  Each entry point
 * below names the reference call site it replaces.  All pointers marked "dev" are device (HBM)
 * addresses; no torch types cross this boundary.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Every function returns 0 on success or a negative CRS_E* code;
 * crs_last_error() returns a thread-local message for the last failure.
 *
 * Nothing here ever falls back to the CPU: a missing GPU is an error.
 */
#ifndef CRS_HIP_H
#define CRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRS_OK 0
#define CRS_EINVAL (-1)   /* bad argument (shape, alignment, k out of range) */
#define CRS_ENOSPC (-2)   /* workspace too small */
#define CRS_EHIP (-3)     /* HIP runtime error (message has hipGetErrorString) */

#define CRS_MAX_K 64      /* largest k one scan launch selects exactly */

/* Slab element types (the vector store's row format in HBM). */
#define CRS_SLAB_F16 0    /* fp16 unit rows                                   */
#define CRS_SLAB_I8 1     /* int8 rows + one fp32 scale per row (s = max|x|/127) */

const char* crs_last_error(void);
int crs_abi_version(void);

/* Slab rows are zero-padded: fp16 rows to a multiple of 128 elements, int8 rows to a multiple of
 * 256 (both = whole 256-byte groups for the LDS swizzle).  crs_row_elems returns the padded row
 * length for an embedding dimension (fp16: 384 -> 384, 100 -> 128; int8: 768 -> 768, 384 -> 512);
 * crs_padded_dim(dim) == crs_row_elems(dim, CRS_SLAB_F16).  Queries searched against a slab use
 * the slab's row length. */
int crs_row_elems(int dim, int slab_type);
int crs_padded_dim(int dim);

/* ---- index build: replaces collection.add(embeddings=...) -- rag/indexing.py:114-119 ------
 * Converts `n` fp32 embedding rows (dev, row stride `dim`) into slab rows starting at row
 * `row0` of `slab` (dev, row stride crs_padded_dim(dim) elements).  Rows are L2-normalised in
 * fp32 first (x / max(||x||, 1e-12)): ChromaDB's cosine space does the same, and it is the
 * identity on already-normalised encoder output (rag/embedding.py:69).  For CRS_SLAB_I8,
 * `scales` (dev fp32, one per slab row) receives max|x|/127 of the normalised row.
 * `shadow_f32` (dev, stride `dim`, may be NULL) receives the normalised fp32 rows for exact
 * re-scoring.
 * `row_err_max` (dev, ONE fp32 that the caller zeroed when it created the slab; may be NULL) is raised to the
 * largest |stored row - normalised fp32 row|_2 seen so far (the stored row being fp16, or int8 * scale):
 * the measured row term of the exactness certificate below (ABI 3). */
int crs_slab_append_f32(const float* emb_dev, int64_t n, int dim, int slab_type, void* slab_dev,
                        float* scales_dev, float* shadow_f32_dev, int64_t row0, float* row_err_max_dev, void* stream);

/* Query side of the same conversion: fp32 [nq, dim] -> normalised fp16 [nq, crs_row_elems(dim, slab_type)]. */
int crs_queries_to_f16(const float* q_dev, int nq, int dim, int slab_type, void* q16_dev, void* stream);

/* ---- search: replaces collection.query(query_embeddings, n_results) -- rag/indexing.py:171-176
 * Exact cosine top-k of `nq` fp16 queries against `n_rows` slab rows on the current device.
 *   q16_dev   fp16 [nq, pdim]  (pdim = crs_row_elems(dim, slab_type)), unit rows, zero padded
 *   slab_dev  fp16 or int8 [n_rows, pdim]; scales_dev fp32 [n_rows] for CRS_SLAB_I8 else NULL
 *             (int8: the kernel moves each query to 16-bit fixed point, see csrc/scan_i8.hip)
 *   k         1..CRS_MAX_K; if k > n_rows the tail slots are (-inf, -1)
 *   id_base   added to row indices (the shard's first global row)
 *   out_scores fp32 [nq, k] cosine, descending; out_ids int64 [nq, k]; ties -> lower id first
 * Workspace: crs_scan_workspace_bytes() bytes of device memory, contents don't-care. */
int crs_scan_workspace_bytes(int nq, int dim, int k, int64_t n_rows, size_t* bytes);
int crs_cosine_topk(const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                    const float* scales_dev, int64_t n_rows, int k, int64_t id_base,
                    void* workspace_dev, size_t workspace_bytes, float* out_scores_dev,
                    int64_t* out_ids_dev, void* stream);

/* ---- multi-shard merge (new: the reference is single process) ------------------------------
 * Merges `nlists` partial results laid out [nlists, nq, k_in] (what an RCCL all-gather of the
 * per-shard crs_cosine_topk outputs delivers) into the global top-k_out per query; slots with
 * id < 0 are ignored.  Same ordering rule. */
int crs_merge_topk(const float* scores_dev, const int64_t* ids_dev, int nlists, int nq, int k_in,
                   int k_out, float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* Exact fp32 re-score of candidates: out[i, j] = <q32[i, :], shadow[ids[i, j], :]> for ids >= 0
 * (id_base subtracted first), then each row re-sorted by (score desc, id asc). Used when the
 * store keeps an fp32 shadow and over-fetches (recall vs an exact fp32 ranking). */
int crs_rescore_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows,
                    int64_t id_base, int k, float* scores_dev, int64_t* ids_dev, void* stream);

/* The scoring half alone, for candidate lists of ANY length k (VectorStore.search with top_k > CRS_MAX_K orders them with a
 * device sort): scores[i, j] = <q32[i, :], shadow[ids[i, j] - id_base, :]>; entries with ids < 0 get -inf, entries of other
 * shards are left as they are. */
int crs_score_rows_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows, int64_t id_base, int k,
                       const int64_t* ids_dev, float* scores_dev, void* stream);

/* Over-fetch + exact re-rank in one launch (Recall@10 = 1.0 against the fp32 ranking the
 * reference's ChromaDB collection keeps, rag/indexing.py:114-119).  cand_ids [nq, k_in] are the rows a
 * crs_cosine_topk call with k = k_in >= k_out found in the fp16 / int8 slab; each is re-scored as the fp32
 * dot product <q32[i], shadow[id - id_base]>, the k_in candidates are ranked (score desc, id asc; ids < 0
 * or outside this shard are empty) and the best k_out leave as out_scores fp32 [nq, k_out] / out_ids int64
 * [nq, k_out] (empty slots: -inf, -1).  One workgroup per query; k_out <= k_in <= CRS_MAX_K. */
int crs_refine_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows,
                   int64_t id_base, const int64_t* cand_ids_dev, int k_in, int k_out,
                   float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ---- exactness certificate + escalation (ABI 3; csrc/exact.hip holds the derivation) ---------------------
 * The reference's store keeps and ranks fp32 rows (rag/indexing.py:114-119,171-176); north_star asks for
 * identical doc-id top-k sets.  crs_refine_f32_cert is crs_refine_f32 plus a per-query PROOF that the re-ranked
 * k_out are the fp32 top-k_out of all n_rows rows, not only of the k_in fetched ones:
 *     status[i] = 0   certified: k_out-th fp32 score > (k_in-th slab score) + eps_i, where eps_i bounds
 *                     |slab score - fp32 score| for query i over every row of the shard (from |q16_i - q32_i|_2,
 *                     measured here, and row_err_max = the value crs_slab_append_f32 tracked; pass a negative
 *                     number to use the analytic worst case crs_exact_row_error_bound instead)
 *     status[i] = 1   not certified (near-ties deeper than the over-fetch, e.g. near-duplicate chunks)
 * cand_scores [nq, k_in] are the slab scores crs_cosine_topk returned with cand_ids; q16 [nq, crs_row_elems] is
 * the query block that scan read, q32 [nq, dim] the unit fp32 queries.
 * crs_escalate_exact then makes every status-1 query exact on the same stream, with no host round trip: one more
 * sweep of the slab lists every row whose slab score is >= (k_out-th fp32 score so far) - eps_i (no row of the true
 * top-k can score lower), and the list is re-ranked in fp32; out_scores / out_ids of those queries are overwritten,
 * certified queries are left alone, and the one kernel's blocks all leave at once when nothing is to do (so the call can sit in a
 * captured graph; the lists are re-ranked by the last block through the sweep -- release fence + counter, no spinning).  A list longer than `cap` rows leaves status[i] = 2: call again with a larger cap
 * (<= CRS_EXACT_MAX_CAP).  Workspace: crs_exact_workspace_bytes(nq, cap) bytes, written by crs_refine_f32_cert and
 * consumed by crs_escalate_exact (same nq, cap). */
#define CRS_EXACT_MAX_CAP 13312
size_t crs_exact_workspace_bytes(int nq, int cap);
float crs_exact_row_error_bound(int dim, int slab_type);
int crs_refine_f32_cert(const float* q32_dev, const void* q16_dev, int nq, int dim, int slab_type, const float* shadow_dev,
                        int64_t n_rows, int64_t id_base, const int64_t* cand_ids_dev, const float* cand_scores_dev, int k_in,
                        int k_out, float row_err_max, float* out_scores_dev, int64_t* out_ids_dev, int32_t* status_dev,
                        void* exact_ws_dev, size_t exact_ws_bytes, int cap, void* stream);
int crs_escalate_exact(const float* q32_dev, const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                       const float* scales_dev, const float* shadow_dev, int64_t n_rows, int64_t id_base, int k_out,
                       float* out_scores_dev, int64_t* out_ids_dev, int32_t* status_dev, void* exact_ws_dev,
                       size_t exact_ws_bytes, int cap, void* stream);

/* ---- one-collective exchange (ONE all-gather per query batch) ------------------
 * A rank's per-shard result travels as one contiguous "wire block":
 *     [ ids int64 [nq, k] | scores fp32 [nq, k] | pad to 8 bytes ]        crs_wire_bytes(nq, k) bytes
 * crs_cosine_topk / crs_refine_f32 write it in place (out_ids = block, out_scores = block +
 * crs_wire_scores_offset(nq, k)); an all-gather of the blocks gives [nlists][crs_wire_bytes] and
 * crs_merge_topk_wire ranks it like crs_merge_topk. */
size_t crs_wire_bytes(int nq, int k);
size_t crs_wire_scores_offset(int nq, int k);
int crs_merge_topk_wire(const void* wire_dev, int nlists, int nq, int k_in, int k_out,
                        float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* Which scan kernel (and launch geometry) crs_cosine_topk would use for these sizes on the current
 * device, as text, e.g. "scan_tb_kernel<384,32,4,0> streams=768 qblocks=1 kp=5 + merge + refine".
 * For bench.py / profiles only; writes at most `cap` bytes including the terminator. */
int crs_scan_plan_describe(int nq, int dim, int k, int64_t n_rows, int slab_type, char* buf, size_t cap);

/* A HIP stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (hipExtStreamCreateWithCUMask; on this part consecutive
 * mask bits go round the XCDs).  The throughput engine (rag/_engine.py) gives its query-encoder lanes such streams: the latency-bound
 * 38-launch encoder chain then shares a few CUs with the corpus sweep instead of touching all of them, and the sweep's dynamic tile
 * schedule routes around those CUs.  No reference counterpart (the reference runs one query at a time on one stream).
 * Destroy with crs_stream_destroy. */
int crs_stream_create_cu_masked(int first_cu, int n_cus, void** stream_out);
int crs_stream_destroy(void* stream);

/* Timing hook for bench.py: runs `iters` back-to-back crs_cosine_topk launches bracketed by
 * hipEvents on `stream` and returns the mean milliseconds of ONE launch pair (scan + merge)
 * in *ms_total and of the scan kernel alone in *ms_scan. */
int crs_time_cosine_topk(const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                         const float* scales_dev, int64_t n_rows, int k, void* workspace_dev,
                         size_t workspace_bytes, float* out_scores_dev, int64_t* out_ids_dev,
                         void* stream, int iters, float* ms_total, float* ms_scan);

#ifdef __cplusplus
}
#endif
#endif /* CRS_HIP_H */
