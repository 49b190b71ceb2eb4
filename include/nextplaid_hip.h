/*
 * nextplaid_hip.h -- C ABI of the MI355X-native PLAID search path for next-plaid.
 *
 * This is the drop-in boundary: the entry points a `hip` cargo feature of the next-plaid crate
 * would bind (extern "C") to serve MmapIndex::{load, search, search_batch} from a gfx950 GPU
 * instead of the crate's CPU path.  Plain pointers and sizes only; no C++/torch types.
 * Each entry cites the reference interface it replaces (paths under /root/reference/next-plaid/src).
 *
 * Conventions
 *  - Every function returning int returns an np_status; 0 = ok.  Codes map onto the crate's
 *    error enum (error.rs:9-66): 1 IndexLoad, 2 Search, 3 Shape, 4 Codec, 5 Io, 6 DeviceUnavailable
 *    (the Rust wrapper falls back to its CPU path unless NEXT_PLAID_FORCE_GPU, mirroring
 *    cuda.rs:105-144), 7 OutOfMemory, 8 InvalidArgument.  Nothing aborts or throws across the ABI.
 *  - np_hip_last_error() is thread-local and valid until the next call on that thread.
 *  - An np_index is immutable after open; all search entry points are re-entrant and may be
 *    called concurrently on one shared handle from many threads (the crate shares &MmapIndex
 *    across tokio workers, next-plaid-api/src/state.rs:413-416).  Each call checks a private
 *    stream + workspace out of a small pool.
 *  - The caller allocates every output buffer; no allocation crosses the boundary.
 *  - Doc ids are GLOBAL i64 ids (position in the concatenation of doclens.*.json) even when the
 *    handle holds one document shard.
 */
#ifndef NEXTPLAID_HIP_H
#define NEXTPLAID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NP_ABI_VERSION 6

typedef enum np_status {
  NP_OK = 0,
  NP_ERR_INDEX_LOAD = 1,         /* Error::IndexLoad  (error.rs:37) */
  NP_ERR_SEARCH = 2,             /* Error::Search     (error.rs:17) */
  NP_ERR_SHAPE = 3,              /* Error::Shape      (error.rs:29) */
  NP_ERR_CODEC = 4,              /* Error::Codec      (error.rs:41) */
  NP_ERR_IO = 5,                 /* Error::Io / Json  (error.rs:21,25) */
  NP_ERR_DEVICE_UNAVAILABLE = 6, /* no usable gfx950 device / HIP runtime failure */
  NP_ERR_OUT_OF_MEMORY = 7,
  NP_ERR_INVALID_ARGUMENT = 8
} np_status;

typedef struct np_index np_index; /* opaque: device-resident index (or one document shard of it) */

/* Options for opening an index.  Zero-initialise, then set what you need. */
typedef struct np_open_opts {
  int32_t device;       /* HIP device ordinal (one process per GPU: pass LOCAL_RANK) */
  int32_t shard_rank;   /* this handle holds documents [N*rank/count, N*(rank+1)/count) */
  int32_t shard_count;  /* 0 or 1 = whole index */
  int32_t n_contexts;   /* concurrent search calls served without blocking (default 2) */
  int32_t max_batch;    /* workspace is sized for this many queries per call (default 64);
                           larger batches are processed in slices */
  int32_t max_query_tokens; /* per-query token cap used to size workspaces (default 64;
                           grows automatically, this is only the initial reservation) */
  int64_t workspace_bytes;  /* soft cap for per-context scratch (0 = default: the HBM left free by the resident index,
                               shared by the contexts, between 2 and 16 GiB) */
} np_open_opts;

/* Mirrors SearchParameters (search.rs:26-69).  batch_size is unused by search and omitted. */
typedef struct np_search_params {
  int32_t top_k;                  /* search.rs:34 */
  int32_t n_full_scores;          /* search.rs:32 */
  int32_t n_ivf_probe;            /* search.rs:36 */
  int32_t centroid_batch_size;    /* search.rs:41: K > this (and > 0) selects the batched-probe
                                     semantics of search.rs:140-254 */
  float centroid_score_threshold; /* search.rs:47 */
  int32_t has_threshold;          /* 0 = None */
  int32_t precision;              /* exact MaxSim stage (S1-S5 are always exact f32):
                                     0 = exact-f32 MFMA on decompressed rows (strict parity mode)
                                     1 = QC-reuse form, bf16 MFMA on the residual term
                                     2 = QC-reuse form, split-bf16 (hi/lo) MFMA: f32-class accuracy
                                     3 = bf16 MFMA on decompressed rows (plain bf16 MaxSim) */
} np_search_params;

typedef struct np_info {            /* accessors of index.rs:1290-1312 */
  int64_t num_documents;            /* whole index */
  int64_t num_embeddings;           /* whole index (metadata.json) */
  int64_t num_partitions;           /* K */
  int32_t embedding_dim;
  int32_t nbits;
  double avg_doclen;
  int64_t shard_doc_begin, shard_doc_end; /* documents held by this handle */
  int64_t shard_embeddings;         /* tokens held by this handle */
  int64_t device_bytes;             /* HBM held by the index (without workspaces) */
  int32_t device;
  int32_t abi_version;
  int64_t workspace_bytes;          /* ABI v5: the LIVE per-context scratch budget (np_open_opts.workspace_bytes, or the
                                       default, which shrinks under memory pressure and grows back; 0 from probe_dir) */
} np_info;

/* Per-call stage timings (HIP events on the call's stream) and work counters.  Optional. */
typedef struct np_stats {
  float ms_total;        /* first launch -> results ready */
  float ms_centroid;     /* S1  Q.C^T (MFMA) + group maxima */
  float ms_probe;        /* S2  top-nprobe per token, threshold, cell list */
  float ms_candidates;   /* S3  posting-list union (bitmap) + compaction; since round 5 also the hot level's thresholds and plane
                            rows (they follow the candidate count), since round 6 the zeroth filter level (three sweeps of the
                            posting lists + the exact bound of its S0 list) */
  float ms_approx;       /* S4  approximate scores (codes x QC gather) */
  float ms_select;       /* S5  top n_full_scores/4 by approximate score */
  float ms_exact;        /* S6  decompress + MaxSim (MFMA) */
  float ms_topk;         /* S7  final top-k */
  int64_t n_cells;       /* probed cells after threshold, summed over the batch */
  int64_t n_ivf_ids;     /* posting-list entries read */
  int64_t n_candidates;  /* unique candidate documents */
  int64_t n_cand_tokens; /* sum of candidate doc lengths (codes read by S4) */
  int64_t n_exact_docs;  /* documents exact-scored */
  int64_t n_exact_tokens;/* tokens decompressed by S6 */
  int64_t n_cand_codes;  /* u8 table rows gathered by the S4 filter (both levels); without the filter: f32 rows */
  int32_t n_queries;
  int32_t n_rounds;      /* candidate-pool rounds (1 unless the batch's candidates overflowed workspace_bytes) */
  int64_t n_survivors;   /* candidates that passed the S4 upper-bound filter and got an exact approximate score */
  int64_t n_cand_dcodes; /* distinct (document, code) pairs of the candidates (<= n_cand_tokens) */
  int64_t n_level2;      /* two-level filter: documents that took the exact u8 bound after the hot bound */
  float ms_hot_level;    /* ABI v5: the first filter level's launch alone (approx_hotp_kernel / approx_hot_kernel of round 0;
                            part of ms_approx; 0 when the two-level filter does not apply) */
  int32_t reserved0;
  int64_t n_level0;      /* ABI v6: candidates the zeroth filter level (per-document sums of the probed cells' gains, S3) handed to
                            the filter; n_candidates stays the size of the posting-list union; 0 when the level did not run */
} np_stats;

/* ---- runtime ------------------------------------------------------------------------------ */

/* Number of usable gfx950 devices (0 if none / no HIP runtime).  Replaces the role of
 * cuda::get_global_context().is_some() (cuda.rs:105-144) for the search path. */
int np_hip_device_count(void);

/* Thread-local description of the last error on this thread ("" if none). */
const char* np_hip_last_error(void);

/* ABI v6: the library's NP_ABI_VERSION, readable BEFORE any struct crosses the boundary.  np_info and np_stats are
 * caller-allocated and have grown (v5: np_info.workspace_bytes, np_stats.ms_hot_level): a host compiled against an older
 * header would have bytes written past its structs before it could read np_info.abi_version.  A host binds this first and
 * refuses a library whose version differs from the header it was built with; np_hip_struct_size lets it check the two
 * layouts it allocates (which: 0 = np_info, 1 = np_stats, 2 = np_search_params, 3 = np_open_opts; -1 for an unknown id). */
int np_hip_abi_version(void);
int64_t np_hip_struct_size(int32_t which);

/* ---- index lifecycle ------------------------------------------------------------------------ */

/* MmapIndex::load (index.rs:1026-1139): reads the crate's on-disk index directory unchanged
 * (metadata.json, centroids.npy, bucket_weights.npy, ivf.npy, ivf_lengths.npy, doclens.N.json,
 * N.codes.npy, N.residuals.npy; NPY v1/v2) and makes it resident in HBM.  The merged_*.npy caches
 * are not needed: chunks are concatenated in the same order. */
int np_hip_index_open(const char* index_dir, const np_open_opts* opts, np_index** out);

/* Same index, built from host arrays in the on-disk dtypes instead of files (what
 * MmapIndex holds after load: index.rs:995-1016).  doc ids in `ivf` are global; with sharding
 * the arrays may cover only the shard's documents (doc_begin = first global id) or the whole
 * index (doc_begin = 0, the shard range is cut out here).  bucket_cutoffs may be NULL. */
typedef struct np_index_arrays {
  int64_t num_documents_total;  /* N of the whole index */
  int64_t doc_begin;            /* global id of doc_lengths[0] */
  int64_t num_docs;             /* entries in doc_lengths */
  int64_t num_centroids;        /* K */
  int32_t dim, nbits;
  const float* centroids;       /* [K, dim] */
  const float* bucket_weights;  /* [2^nbits] */
  const int64_t* ivf;           /* concatenated posting lists, global doc ids ascending per list */
  const int32_t* ivf_lengths;   /* [K] */
  const int64_t* doc_lengths;   /* [num_docs] */
  const int64_t* codes;         /* [sum doc_lengths] */
  const uint8_t* residuals;     /* [sum doc_lengths, dim*nbits/8] */
} np_index_arrays;
int np_hip_index_from_arrays(const np_index_arrays* arrays, const np_open_opts* opts, np_index** out);

/* Seeded synthetic corpus generated directly in HBM (bench / large-scale tests; the generator
 * spec is next-plaid_amd/next_plaid_amd/synth.py, bit-identical).  centroids / bucket_weights
 * are host arrays.  Sharding as in np_open_opts. */
typedef struct np_synth_spec {
  int64_t num_docs;             /* whole corpus */
  int64_t num_centroids;
  int32_t dim, nbits;
  int32_t doc_len_min, doc_len_max;
  int32_t n_topics, rand256;
  uint64_t seed;
  const float* centroids;       /* [K, dim] */
  const float* bucket_weights;  /* [2^nbits] */
  const int32_t* len_table;     /* optional: document length = len_table[hash(doc) % len_table_size] (a quantile table,
                                   e.g. the clipped LogNormal of MS MARCO passages) instead of uniform [doc_len_min, doc_len_max] */
  int32_t len_table_size;       /* 0 = none */
} np_synth_spec;
int np_hip_index_synth(const np_synth_spec* spec, const np_open_opts* opts, np_index** out);

/* Copies the shard held by a handle back to host arrays in the on-disk dtypes (ivf ids global).
 * Pass NULL for any array not wanted.  Sizes come from np_hip_index_info / np_hip_index_ivf_size. */
int np_hip_index_export(const np_index* index, int64_t* doc_lengths, int64_t* codes, uint8_t* residuals,
                        int64_t* ivf, int32_t* ivf_lengths);
int64_t np_hip_index_ivf_size(const np_index* index);

/* Host only (no device): writes host arrays as an index DIRECTORY in the crate's on-disk format -- the canonical file
 * set of write_index_from_encoded_chunks (index.rs:373-528): centroids / bucket_cutoffs / bucket_weights / avg_residual /
 * cluster_threshold .npy, plan.json, per chunk of <= chunk_docs documents {i}.metadata.json, doclens.{i}.json,
 * {i}.codes.npy, {i}.residuals.npy, then ivf.npy, ivf_lengths.npy and metadata.json.  NPY 1.0 headers padded to 64 bytes
 * (mmap.rs:1176-1250), every file via temporary name + fsync + rename (utils.rs:16-60).  arrays->ivf / ivf_lengths may
 * be NULL: the posting lists (ascending unique document ids per centroid, index.rs:479-504) are then built from the
 * codes.  arrays must hold the whole index (doc_begin = 0).  With np_hip_encode_tokens this is the index-build path:
 * encode on the GPU, write here, and MmapIndex::load (the crate's or np_hip_index_open) reads the result. */
typedef struct np_write_opts {
  int64_t chunk_docs;           /* documents per chunk; 0 = 50 000 (IndexConfig.batch_size, index.rs:92) */
  const float* bucket_cutoffs;  /* [2^nbits - 1] or NULL (file not written; search does not read it) */
  const float* avg_residual;    /* [dim] or NULL = zeros (not used by search arithmetic) */
  float cluster_threshold;      /* update path only (update.rs:372) */
} np_write_opts;
int np_hip_index_write_dir(const char* index_dir, const np_index_arrays* arrays, const np_write_opts* opts);

/* Kernel-selection knobs (no reference counterpart).  Each knob is read from the environment once, at open, as
 * NP_<UPPER-CASE NAME>, and can be changed on a live handle with this call (sweep tools, kernel-variant parity tests);
 * both paths clamp through one table.  Knobs: "s4_mode" 0..8, "s4_minb" >= 1, "s4_nbx" 8..512, "s4_swz" 0/1,
 * "s4_filter" 0/1, "s4_hot" 0..500 (per-mille of hot centroids in the first filter level; 0 = single-level filter),
 * "s4_planes" 0/1 (first level in bit planes; read at open too: it sets the list-block cap), "s4_pexp" 5..40, "s4_lpd" 2/4,
 * "s4_qm" 0/1, "s4_pnbx" 8..512, "s4_warm" 0..1000 (per-mille of centroids whose rows the exact filter level still gathers
 * for the S2 lists; 1000 = every row; 0 = the default: by query length and mean distinct-code count), "s4_hot_auto" >= 0 (candidates
 * per query up to which "s4_hot" applies as given; beyond it the share falls with the count^(-1/3); 0 = always as given),
 * "ub_ncut" 1..512, "s3_bisect" 0/1, "s3_gain" 0/1/2 (zeroth filter level in S3: 0 = off -- read at open too: its range table is not built --, 2 = whenever it
 * applies, 1 = the default: 2 with a run / skip policy fed by the previous batches' pruning; with a centroid_score_threshold it starts skipped),
 * "s3_gain_mult" 1..16, "s3_gain_direct" 0..64, "s1_split" 0/1 (the only knob that changes values: see INTEGRATION.md),
 * "s3_slices" 0/1, "ub_nt" 0..2, "ub_steal" >= 1, "ub_nbx" 8..256, "ub_direct" 0..16, "ub_static" 0/1, "hot_static" 0/1, "s6_xcd" 0/1, "s6_tiles" 0/1, "s6_lds" 0..2, "gemm_cpw" 1/2, "exact_rowmax" 0/1.
 * Results are identical for every setting except "s1_split"; not synchronised with concurrent searches.  Unknown name:
 * NP_ERR_INVALID_ARGUMENT.  (A library built with -DNP_DIAGNOSTICS also accepts "s4_probe" 0..7, a phase-skipping timing
 * probe whose results are invalid; production builds reject the name and never read it from the environment.) */
int np_hip_index_tune(np_index* index, const char* name, int32_t value);

void np_hip_index_close(np_index* index);               /* Drop for MmapIndex */
int np_hip_index_info(const np_index* index, np_info* out); /* index.rs:1290-1312 */

/* ---- search ------------------------------------------------------------------------------------ */

/* MmapIndex::search_batch (index.rs:1279-1287 -> search.rs:643-675); MmapIndex::search
 * (index.rs:1258-1265) is the B = 1 case.
 *   queries        row-major f32, all queries' token rows concatenated: [q_tok_offsets[B], dim]
 *   q_tok_offsets  B+1 prefix offsets (query i owns rows [off[i], off[i+1]))
 *   subset         optional pre-filter doc ids (search.rs:350-382,434-437); subset_len < 0 = None,
 *                  subset_len == 0 = empty subset (every result empty)
 *   out_ids/out_scores  [B * top_k], query i at [i*top_k, i*top_k + out_counts[i]); scores descending
 *   out_counts     [B]
 * Host pointers; H2D/D2H copies are inside the call.
 * Geometry: every index the crate writes with embedding_dim <= 128 is searchable -- nbits 1, 2, 4, 8 (codec.rs:161-166),
 * any dim with dim * nbits % 8 == 0.  `dim` here, np_info.embedding_dim / nbits, decompressed rows, exported and encoded
 * residual rows are always in the geometry of the index FILES; inside, rows are stored at the next of the four kernel
 * widths (32 / 64 / 96 / 128, zero-padded) and 1-bit buckets as 2-bit segments.  A wider index opens (info, decompress,
 * export work) and search returns NP_ERR_SHAPE: the wrapper's signal to take the CPU path. */
int np_hip_search_batch(const np_index* index, const float* queries, const int32_t* q_tok_offsets,
                        int32_t B, int32_t dim, const np_search_params* params,
                        const int64_t* subset, int64_t subset_len,
                        int64_t* out_ids, float* out_scores, int32_t* out_counts, np_stats* stats);

/* Same call with every buffer already resident in HBM on `index`'s device and the work enqueued on
 * `stream` (a hipStream_t; NULL = the context's own stream).  Returns after enqueueing; the caller
 * synchronises the stream.  `stats` (host) is filled only by np_hip_search_batch.  This is what a
 * multi-GPU host and bench.py use (queries resident, results consumed on device by the merge). */
int np_hip_search_batch_device(const np_index* index, const float* d_queries,
                               const int32_t* d_q_tok_offsets, const int32_t* h_q_tok_offsets,
                               int32_t B, int32_t dim, const np_search_params* params,
                               const int64_t* d_subset, int64_t subset_len,
                               int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts,
                               void* stream);

/* ---- document-sharded search (one process per GPU) -------------------------
 * Phase A runs S1-S5 on the local shard and leaves, per query, the shard's best
 * n_sel = min(n_full_scores, max(n_full_scores/4, top_k)) candidates as 64-bit rank keys in
 * d_sel_keys[B * n_sel] (descending; key = orderable(approx score) << 32 | ~global_doc_id; 0 pads).
 * The host all-gathers the keys over RCCL, takes each query's global n_sel-th key as the cut, and
 * phase B exact-scores only the local candidates with key >= d_cut[b], returning the local top-k
 * as (score, id, key) triples.  np_hip_merge_topk then merges the G shards' triples into the
 * final top-k with the reference's tie rules (exact score desc, then approx rank).  With
 * shard_count == 1 and d_cut == NULL the result equals np_hip_search_batch. */
int np_hip_search_phase_a(const np_index* index, const float* d_queries, const int32_t* d_q_tok_offsets,
                          const int32_t* h_q_tok_offsets, int32_t B, int32_t dim,
                          const np_search_params* params, const int64_t* d_subset, int64_t subset_len,
                          const uint32_t* d_elig_global, uint64_t* d_sel_keys, void* stream, void** call_state);
/* With a `subset` the dense path restricts the probe to the centroids that occur in the subset's documents and
 * scales n_ivf_probe by their number (search.rs:350-382).  A document shard only sees its own documents, so for a
 * result identical to the unsharded search the host ORs the shards' bitmaps: np_hip_subset_eligible writes this
 * shard's bitmap (np_hip_elig_words() u32 words), the host all-gathers them, np_hip_or_bitmaps combines
 * d_all[G][words] and the result goes into phase A as d_elig_global (NULL = use the local bitmap: unsharded). */
int64_t np_hip_elig_words(const np_index* index);
int np_hip_subset_eligible(const np_index* index, const int64_t* d_subset, int64_t subset_len, uint32_t* d_elig_bits,
                           void* stream);
int np_hip_or_bitmaps(const np_index* index, const uint32_t* d_all, int32_t G, int64_t words, uint32_t* d_out,
                      void* stream);
int np_hip_search_phase_b(const np_index* index, void* call_state, const uint64_t* d_cut,
                          int64_t* d_out_ids, float* d_out_scores, uint64_t* d_out_keys,
                          int32_t* d_out_counts, void* stream);
/* Ends a phase-A/phase-B call and returns its context to the pool (always call it). */
void np_hip_search_end(const np_index* index, void* call_state);
/* n_sel for given params (size of one query's slice of d_sel_keys). */
int32_t np_hip_n_sel(const np_search_params* params);
/* Global cut from G gathered key lists: d_all_keys[G][B][n_sel] -> d_cut[B]. */
int np_hip_select_cut(const np_index* index, const uint64_t* d_all_keys, int32_t G, int32_t B,
                      int32_t n_sel, uint64_t* d_cut, void* stream);
/* Merge G shards' local top-k triples ([G][B][top_k], counts [G][B]) into out [B][top_k]. */
int np_hip_merge_topk(const np_index* index, const int64_t* d_ids, const float* d_scores,
                      const uint64_t* d_keys, const int32_t* d_counts, int32_t G, int32_t B,
                      int32_t top_k, int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts,
                      void* stream);

/* The same merge over one packed record per rank: rank g's record starts at d_records + g * record_bytes and holds
 * ids [B*top_k] i64 at 0, keys [B*top_k] u64 at off_keys, scores [B*top_k] f32 at off_scores, counts [B] i32 at
 * off_counts (one all-gather instead of four). */
int np_hip_merge_packed(const np_index* index, const void* d_records, int64_t record_bytes, int64_t off_keys,
                        int64_t off_scores, int64_t off_counts, int32_t G, int32_t B, int32_t top_k,
                        int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts, void* stream);

/* ---- the whole sharded protocol in one call, RCCL below the ABI (np_dist.hip) -----------------------------------
 * One process per GPU.  Every rank opens its document shard (np_open_opts.shard_rank / shard_count = rank / nranks),
 * rank 0 draws a 128-byte id (np_hip_comm_unique_id = ncclGetUniqueId) and hands it to the others by whatever
 * channel the host has (the crate: its own RPC / a file; bench.py: a torch.distributed broadcast), every rank calls
 * np_hip_comm_create (= ncclCommInitRank; collective).  np_hip_search_batch_sharded then runs phase A ->
 * ncclAllGather(keys) -> cut -> phase B -> ncclAllGather(packed top-k) -> merge on `stream` and leaves the GLOBAL
 * top-k (identical to the unsharded search, subsets included) in the output buffers of EVERY rank.  All ranks must
 * call it with the same queries / params / subset in the same order.  librccl is loaded at first use
 * (dlopen "librccl.so.1"; NEXTPLAID_RCCL_LIB overrides); nranks == 1 with id128 == NULL needs no RCCL at all.
 * A communicator serialises its calls; use one per concurrent stream.
 *
 * Failure of ONE rank never blocks the others: every exchange record carries a status word, a rank whose local work
 * fails (its workspace does not fit, a launch error) still takes part in both all-gathers with empty data and returns its
 * own error at once; on the other ranks the batch comes back ABANDONED -- every out_counts[i] = -1 (NP_COUNT_ABANDONED), a
 * count no healthy batch produces: a host MUST treat a negative count as a failed batch, whether or not it polls
 * np_hip_comm_status -- and np_hip_comm_status, valid once `stream` is synchronised, names the failed rank and its np_status.  With the hosted transport below the
 * gathered bytes pass through the host, so every rank returns NP_ERR_SEARCH from the call itself (after gather 1). */
typedef struct np_comm np_comm;
#define NP_COUNT_ABANDONED (-1)
int np_hip_comm_unique_id(void* id128);
int np_hip_comm_create(const np_index* index, const void* id128, int32_t rank, int32_t nranks, np_comm** out);
void np_hip_comm_destroy(np_comm* comm);
/* Reads and clears the failure word of the communicator's last batches: *failed_rank = -1 and *code = 0 if every batch
 * since the last call was healthy, else the first failed rank and its np_status.  Call after synchronising the stream. */
int np_hip_comm_status(np_comm* comm, int32_t* failed_rank, int32_t* code);
/* ABI v6: what a communicator really is, for a bench line or a health endpoint.  *transport = NP_COMM_LOCAL (one rank, no
 * collective library at all), NP_COMM_RCCL (ncclCommInitRank succeeded) or NP_COMM_HOSTED; *nranks = the size the
 * communicator was created with; *rccl_ranks = ncclCommCount of the RCCL communicator (the number of ranks RCCL itself
 * sees: equals nranks on a healthy communicator), 0 for the other transports or a librccl without ncclCommCount.  Any
 * pointer may be NULL. */
#define NP_COMM_LOCAL 0
#define NP_COMM_RCCL 1
#define NP_COMM_HOSTED 2
int np_hip_comm_info(np_comm* comm, int32_t* transport, int32_t* nranks, int32_t* rccl_ranks);
/* The same protocol over a transport the HOST brings (MPI, gloo, shared memory, the crate's own RPC) instead of RCCL:
 * for hosts without librccl, for ranks that share one GPU (RCCL refuses two ranks on a device), and for the tests that
 * run the shipped multi-rank code path on a one-GPU box.  `all_gather(ctx, send, recv, bytes)` is called on the calling
 * thread with HOST pointers: it must place rank r's `bytes` bytes at recv + r * bytes for every rank and return 0.  The
 * library stages the (<= 0.6 MB) records through pinned memory and synchronises the stream around the callback, so a
 * hosted communicator costs two stream synchronisations per batch; results are identical to the RCCL transport.
 * flags: NP_COMM_DEFERRED_STATUS = do not inspect the gathered status words on the host (propagate a failure on the
 * device like the RCCL transport does; np_hip_comm_status reports it). */
typedef int (*np_all_gather_host_fn)(void* ctx, const void* send, void* recv, int64_t bytes);
#define NP_COMM_DEFERRED_STATUS 1
int np_hip_comm_create_hosted(const np_index* index, int32_t rank, int32_t nranks, np_all_gather_host_fn all_gather,
                              void* ctx, int32_t flags, np_comm** out);
int np_hip_search_batch_sharded(const np_index* index, np_comm* comm, const float* d_queries,
                                const int32_t* d_q_tok_offsets, const int32_t* h_q_tok_offsets, int32_t B, int32_t dim,
                                const np_search_params* params, const int64_t* d_subset, int64_t subset_len,
                                int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts, void* stream);

/* Host-only validation of an index directory: parses and checks every file exactly as np_hip_index_open
 * does (MmapIndex::load, index.rs:1026-1139; NPY headers mmap.rs:659-749; fast-plaid dtypes mmap.rs:1780-1808)
 * without touching a device, and reports the index geometry (out->device = -1).  Same error codes/messages as
 * np_hip_index_open. */
int np_hip_index_probe_dir(const char* index_dir, np_info* out);

/* ---- adjacent rows ----------------------------------------------------- */

/* N2: MmapIndex::get_document_embeddings / decompress_documents (index.rs:1159-1245): decompressed,
 * L2-normalised f32 embeddings of the given global doc ids, concatenated; out_lengths[i] = tokens of
 * doc i (0 for ids outside this shard).  out_embeddings may be NULL to query lengths only. */
int np_hip_decompress_documents(const np_index* index, const int64_t* doc_ids, int64_t n_docs,
                                float* out_embeddings, int64_t out_capacity_rows, int64_t* out_lengths);

/* N3: index-time encode of a flat batch of token embeddings against this index's codec -- replaces
 * ResidualCodec::compress_into_codes + compress_and_residuals + quantize_residuals (codec.rs:297-411,
 * index.rs:17-40,289-371 encode_index_chunk; the reference's CUDA path cuda.rs:185-237,353-653).
 * out_codes[t] = nearest centroid by dot product (last index among equal maxima, non-finite scores below
 * every finite one: Iterator::max_by(cmp_f32_for_max)); out_packed[t][dim*nbits/8] = residual buckets in the
 * on-disk bit layout.  bucket_cutoffs holds 2^nbits - 1 floats (bucket_cutoffs.npy).  Host pointers. */
int np_hip_encode_tokens(const np_index* index, const float* embeddings, int64_t n_tokens, int32_t dim,
                         const float* bucket_cutoffs, int64_t* out_codes, uint8_t* out_packed);

/* N4: /rerank MaxSim on caller-supplied embeddings (next-plaid-api/src/handlers/rerank.rs:57-94 compute_maxsim,
 * :139-170 scoring + sort).  query [n_query_tokens][dim]; documents concatenated, document i owns rows
 * doc_tok_offsets[i] .. doc_tok_offsets[i+1].  out_scores[n_docs] in input order; out_order (nullable) =
 * document indices sorted by descending score, stable.  NP_ERR_INVALID_ARGUMENT with the handler's message for
 * "No documents provided" and "Rerank score contains non-finite value".  Needs no index; host pointers. */
int np_hip_rerank_maxsim(int32_t device, const float* query, int32_t n_query_tokens, int32_t dim,
                         const float* doc_embeddings, const int64_t* doc_tok_offsets, int64_t n_docs,
                         float* out_scores, int64_t* out_order);

/* Stage-level debug access for parity tests: runs S1-S5 for ONE query and copies out the probed
 * cells (ascending), candidate doc ids (ascending, global), their approximate scores, and the
 * selected docs in approx-rank order with their exact scores.  Capacities are in elements;
 * counts are returned in n_*.  Any pointer may be NULL. */
int np_hip_debug_trace(const np_index* index, const float* query, int32_t n_tokens, int32_t dim,
                       const np_search_params* params, const int64_t* subset, int64_t subset_len,
                       int64_t* cells, int64_t cap_cells, int64_t* n_cells,
                       int64_t* cand, float* approx, int64_t cap_cand, int64_t* n_cand,
                       int64_t* sel, float* sel_exact, int64_t cap_sel, int64_t* n_sel);

#ifdef __cplusplus
}
#endif
#endif /* NEXTPLAID_HIP_H */
