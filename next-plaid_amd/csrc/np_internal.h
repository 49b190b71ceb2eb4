// np_internal.h -- shared declarations of libnextplaid_hip.so (not part of the C ABI).
//
// Data layout in HBM.  One np_index = one document shard:
//   centroids   f32 [K][dim]            replicated on every shard   (centroids.npy)
//   wlut        f32 [2^nbits]           permuted bucket weights: wlut[s] = bucket_weights[bitrev_nbits(s)]
//                                       (folds codec.rs:168-214's two LUTs into one)
//   codes       u16 [T] (K <= 65536) or u32 [T]   centroid id per token   (N.codes.npy, i64 on disk, range-checked at load)
//   residuals   u8  [T][pd]             packed buckets, unchanged   (N.residuals.npy)
//   ucodes      u16 / u32 [U]           derived at open: the documents' sorted DISTINCT-code lists: one fixed-stride block per
//                                       document (16-byte header + codes; the stride holds 99.9 % of the lists) followed by an
//                                       overflow region for longer lists; the doc_meta record carries the list's offset either
//                                       way.  S4 reads these.  ulen i32 [n_docs] = list lengths
//   inv_norm    f32 [T]                 derived at open: 1/||centroid + residual|| per token (S6 QC-reuse form)
//   tok_pos     u16 [T]                 derived at open: codes / residuals / inv_norm keep each document's tokens ORDERED BY
//                                       CODE (MaxSim is a max over tokens: order-free), tok_pos = the original position
//                                       (decompress_documents / export restore the on-disk order)
//   doc_offsets i64 [n_docs+1]          prefix sum of doclens       (index.rs:1106-1110)
//   ivf         u32 [ivf_size]          shard-local doc ids         (ivf.npy re-based)
//   ivf_offsets i64 [K+1]                                           (index.rs:1089-1094)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <string>
#include <vector>
#include "../../include/nextplaid_hip.h"

namespace np {

// A code array of the index (token codes, or the documents' distinct-code lists): u16 entries when every centroid id
// fits 16 bits (K <= 65536), u32 otherwise.  `wide` is a kernel argument, so the branch is wave-uniform.
struct CodeArr {
  const void* p;
  int wide;
  __device__ __forceinline__ uint32_t operator[](int64_t i) const {
    return wide ? static_cast<const uint32_t*>(p)[i] : (uint32_t)static_cast<const uint16_t*>(p)[i];
  }
};

// ---- errors --------------------------------------------------------------------------------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();
void clear_error();

struct Status {
  int code;
  Status(int c = NP_OK) : code(c) {}
  bool ok() const { return code == NP_OK; }
};

#define NP_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      np::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (e__ == hipErrorOutOfMemory) ? NP_ERR_OUT_OF_MEMORY : NP_ERR_DEVICE_UNAVAILABLE; \
    }                                                                                        \
  } while (0)

#define NP_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != NP_OK) return rc__; \
  } while (0)

// ---- host view of an index (on-disk dtypes), produced by the loader or handed in by the caller
struct HostChunk {
  const int64_t* codes = nullptr;     // [n_tokens]
  const uint8_t* residuals = nullptr; // [n_tokens][pd]
  int64_t n_tokens = 0;
};

struct HostIndex {
  int64_t num_documents_total = 0, num_embeddings_total = 0;
  double avg_doclen = 0.0;
  int64_t doc_begin = 0;               // global id of doc_lengths[0]
  int64_t K = 0;
  int32_t dim = 0, nbits = 0;
  const float* centroids = nullptr;
  const float* bucket_weights = nullptr;
  const int64_t* ivf = nullptr;        // global doc ids
  int64_t ivf_size = 0;
  const int32_t* ivf_lengths = nullptr;
  std::vector<int64_t> doc_lengths;    // docs [doc_begin, doc_begin + size)
  std::vector<HostChunk> chunks;       // token data of those docs, in order
  // keeps mmaps / owned buffers alive
  std::vector<std::pair<void*, size_t>> maps;
  std::vector<std::vector<char>> owned;
  ~HostIndex();
};

// np_loader.cpp: MmapIndex::load's file parsing (index.rs:1026-1139, mmap.rs:659-749)
int load_index_dir(const char* dir, HostIndex* out);

// ---- device buffers ------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);  // grows (never shrinks); contents are NOT preserved
  void release();
  template <class T> T* as() const { return (T*)p; }
};

struct Workspace;  // np_search.hip

struct Context {
  hipStream_t stream = nullptr;
  hipEvent_t ev[10] = {};
  Workspace* ws = nullptr;
  bool busy = false;
  uint64_t last_use = 0;   // acquisition stamp: idle contexts are handed out least-recently-used first
};

// Kernel-selection knobs.  Read from the environment ONCE, at index open (NP_<NAME>, e.g. NP_S4_MODE); np_hip_index_tune()
// changes them on a live handle for sweep tools and the kernel-variant parity tests; both go through set_tuning()'s one
// clamp table.  Every setting produces identical results.
struct Tuning {
  int s4_mode = 4;       // 0 approx_kernel; 1..4 approx_xcd_kernel with 8/4/2/1 phases; 5..8 approx_stream_kernel (the survivor lists
                         // of the filter are short: one phase measured best; 2 was best on unfiltered 18 k-document lists)
  int s4_minb = 8;       // smallest batch that takes the per-XCD kernels
  int s4_nbx = 128;      // workgroups per XCD
  int s4_swz = 1;        // ds_swizzle vs ds_bpermute code broadcast
  int s3_bisect = 1;     // S3: a bitmap range reads only its part of every (ascending) posting list instead of sweeping all of it
  int s3_slices = 1;     // S3: document-bitmap ranges built in LDS (mark_slices_kernel) instead of atomicOr in memory
  int s3_gain = 1;       // zeroth filter level (gain_sweep_kernel): per-document sums of the probed cells' gains prune the candidates
                         // before any list block is read -- where no centroid_score_threshold is set (with one, the removed cells
                         // lift the bound's floor above the cut: tools/sim/s3_gain_sim.py).  0 off (read at OPEN too: the range
                         // table of the posting lists is not built), 2 whenever it applies, 1 (default) = 2 with a run / skip policy:
                         // the level costs ~2 ms per batch at 10 M documents whatever it prunes, so while the candidates it removed
                         // in the handle's recent batches would not have cost the filter that much (K = 2^19: the bound's floor
                         // reaches the cut, 85 % of the candidates stay) it is skipped for 31 batches and tried again
  int s3_gain_mult = 3;  // ... S0 = the s3_gain_mult x n_sel candidates with the largest bound take the exact bound first (tau0)
  int s3_gain_direct = 16;   // ... workgroups per query of that launch (0 = one query per XCD at a time, like the S2 list)
  int s4_filter = 1;     // u8 upper-bound filter ahead of the exact f32 approximate scores
  int s4_hot = 60;       // per-mille of the centroids that are "hot" for a query in the first filter level (0 = single-level
                         // filter).  More hot centroids: fewer documents left to the exact bound (S1 + S2: 2.79 M / 2.20 M / 1.81 M of
                         // 11.9 M at 60 / 100 / 150) but more plane rows and walk steps per document in the hot kernel.  With the
                         // floored exact level (s4_warm) behind it, 10 M documents: 40 / 60 / 80 / 100 / 150 -> 18.3 / 18.6 / 18.4 /
                         // 17.8 / 16.6 k queries/s (round 3, byte maxima and every row at the exact level: 100 was best)
  int s4_hot_auto = 200000;   // candidates per query up to which s4_hot applies as given; beyond, the share falls with the
                         // candidate count^(-1/3), never below 8 per mille (hot_levels_kernel; 0 = s4_hot always).  The REST
                         // API's default regime (t_cs = None, nprobe 8: 2 M candidates per query at 10 M documents) wants ~30
  int s4_planes = 1;     // first filter level in bit-plane form (approx_hotp_kernel: 8 planes per hot centroid, OR + weighted popcount
                         // instead of 32 byte maxima per table row); 0 = approx_hot_kernel.  Read at OPEN too: with it the list blocks
                         // may be up to 512 bytes (corpora with long distinct-code lists), which approx_hot_kernel cannot stage
  int s4_lpd = 2;        // approx_hotp_kernel: lanes per document -- 4: claims of 16 documents (half the LDS rows per wave: more waves per
                         // CU hide the latency of the block loads), 2: claims of 32.  512-byte blocks always take 4
  int s4_pnbx = 96;      // ... workgroups per XCD (3 per CU at 42 KB of LDS each with 2 lanes per document; 160 = 5 per CU with 4)
  int s4_qm = 1;         // ... a lane's hot codes as a position mask in registers (1) or compacted in place by LDS writes (0)
  int s4_pexp = 15;      // plane levels: t_j = Lambda + span * (j / 8)^(s4_pexp / 10); 10 = uniform (hot_levels_kernel)
  int s4_warm = 0;       // ... per-mille of the centroids whose rows the exact level still gathers for the S2 list (the rest: floored
                         // at Lambda2; approx_ub_kernel FLOOR); 1000 = every row; 0 = by query length and list length (np_search.hip)
  int ub_ncut = 96;      // workgroups per query of the cut kernels (each appends 2048 records per step: latency-bound per step)
  int ub_direct = 8;     // workgroups per query of the short-list (S1) exact-bound launch; 0 = the per-XCD hand-out
  int hot_static = 1;    // hot kernel: waves take a query's claims round-robin (no cursor atomic: a device-scope atomic per claim
                         // on a line all XCDs share costs ~50 ns, serialised): 2.06 -> 1.68 ms at 10 M documents
  int ub_static = 0;     // the same for the exact-bound kernel: measured WORSE there (single-level S4 4.43 -> 5.17 ms: workgroups
                         // that run ahead pull the next query's table into the L2), so it keeps the cursor
  int s4_probe = 0;      // DIAGNOSTIC (results invalid when != 0): phases of the hot kernel to skip, for timing them
  int ub_nbx = 96;       // filter workgroups per XCD: 3 per CU (48 KB of LDS each).  64 makes the stage itself 3 % faster (0.70 vs 0.73 ms at
                         // 1 M, 4.43 vs 4.57 ms at 10 M documents) but the sustained 3-stream rate at 10 M drops 10.8 k -> 10.3 k queries/s
  int ub_steal = 16384;  // filter: an idle XCD joins a running query that has at least this many unclaimed documents (0x7fffffff = never)
  int ub_nt = 2;         // filter loads: 0 plain, 1 non-temporal records / code lists, 2 bounds-checked buffer loads of the table
  int s6_xcd = 1;        // one XCD per query in S6
  int s6_lds = 1;        // QC-reuse S6: query fragments in LDS + C-in rows prefetched one tile ahead (exact_qcl_kernel); 0 = exact_qct_kernel
  int s6_tiles = 1;      // QC-reuse S6: one launch of the one-tile kernel per 32-token query tile (0: the multi-tile kernels)
  int s1_split = 0;      // OPT-IN (np_hip_index_tune / NP_S1_SPLIT): split-bf16 S1 (qc_gemm_b3_kernel) when precision >= 1 and K >
                         // centroid_batch_size -- S1-S5 are then no longer bit-equal to the f32 chain (near-ties < ~1e-5 can reorder)
  int gemm_cpw = 1;      // centroid fragments per wave in S1
  int exact_rowmax = 0;  // force the row-max form of the QC-reuse S6 kernel
};

// documents per range of the posting lists' range table (d_ivf_split) = per block of the zeroth filter level's sweep (np_kernels.h)
#define NP_IVF_SPLIT_RANGE 32768

struct DeviceIndex {
  int device = 0;
  int64_t N_total = 0, n_emb_total = 0;
  double avg_doclen = 0.0;
  int64_t doc_begin = 0, n_docs = 0;
  int64_t K = 0, KP = 0;  // KP = K rounded up to 64
  int32_t dim = 0, nbits = 0, pd = 0;     // STORAGE geometry: what every kernel is instantiated on (see storage_dim below)
  int32_t ldim = 0, lnbits = 0, lpd = 0;  // geometry of the index files and of the caller's queries / embeddings
  float pad_ss = 0.f;                     // (dim - ldim) * wlut[0]^2, see ExactP::pad_ss
  int64_t T = 0;
  int64_t max_doc_len = 0;
  float* d_centroids = nullptr;
  float* d_wlut = nullptr;
  void* d_codes = nullptr;        // [T] u16 when code_wide == 0 (K <= 65536), else u32
  int code_wide = 0;
  void* d_ucodes = nullptr;       // [n_ucodes] dense per-document sorted distinct-code lists, same element type (derived)
  int64_t n_ucodes = 0;           // entries of d_ucodes without the tail padding
  int ublock_stride = 0;          // entries per document block of d_ucodes (header + codes; lists that do not fit: overflow region)
  int ublock_hdr = 0;             // entries of a block's 16-byte header {#distinct, doc length, overflow index, 0}
  int32_t* d_ulen = nullptr;      // [n_docs] number of distinct codes per document (derived)
  uint4* d_useg = nullptr;        // [n_docs] 8 x u16: distinct codes below each eighth of the centroid range (derived)
  uint4* d_doc_meta = nullptr;    // [n_docs] the 16-B candidate record of every document {doc, n distinct codes, offset of its
                                  // distinct-code list in d_ucodes: low 32 bits, bits 32..39 | doc length << 8} (derived)
  float ulen_mean = 0.f;          // mean number of distinct codes per document (derived; scales the exact level's floor)
  bool sliced_ok = false;         // every document's distinct-code list is sorted and < 65536 long
  float cmax = 0.f;               // upper bound of the centroid row norms (derived; scales the S4 u8 score table)
  bool filter_ok = false;         // every centroid value is finite: the S4 upper-bound filter may run
  bool s6_fast_ok = false;        // centroids and bucket weights finite and < 1e6 in magnitude: with an unflagged query no S6
                                  // product can overflow, so the QC-reuse kernel drops its non-finite guard
  float* d_inv_norm = nullptr;    // [T] 1 / max(||centroid[code] + residual||, 1e-12) per token (derived)
  uint16_t* d_tok_pos = nullptr;  // [T] original position (inside its document) of the token stored here (derived; see below)
  bool tok_sorted = false;        // codes / residuals / inv_norm hold every document's tokens ORDERED BY CODE
  uint8_t* d_residuals = nullptr;
  int64_t* d_doc_offsets = nullptr;
  uint32_t* d_ivf = nullptr;
  int64_t* d_ivf_offsets = nullptr;
  int64_t ivf_size = 0;
  bool ivf_sorted = false;        // every posting list ascends (what the crate writes): S3 may bisect a list for a document range
  uint32_t* d_ivf_split = nullptr;   // [K][n_ranges + 1] (derived, ascending lists only): entries of list c with id < 32768 r -- the
                                  // zeroth filter level's blocks read their range's part of a posting list without a bisection
  int n_ranges = 0;               // ceil(n_docs / 32768)
  // ivf_top_prefix[i] = entries of the i longest posting lists together (host side, i <= K): a query that probes c cells cannot
  // have more candidates than ivf_top_prefix[c] (search.rs:427-452 unions the probed cells' lists), which is what the workspace
  // planner sizes the candidate pool and the number of pool rounds for -- n_docs per query only when the lists do not say less
  std::vector<int64_t> ivf_top_prefix;
  size_t device_bytes = 0;
  np_open_opts opts{};
  // per-context scratch budget the planner uses.  A caller-given workspace_bytes is kept as it is; the default (what the device
  // had free at open, np_index.hip default_workspace) is re-clamped against what is free NOW whenever a candidate pool has to
  // grow, and halved when a reservation still fails: a second index opened on the device (the "swap handles" reload pattern),
  // or an encoder allocating later, shrinks the pool (more rounds) instead of failing the search with OutOfMemory.
  mutable std::atomic<int64_t> ws_budget{0};
  int64_t ws_budget_open = 0;   // the budget at open: the live one grows back towards it when the device has headroom again
  bool ws_auto = false;
  Tuning tune;
  // zeroth filter level, run / skip policy (s3_gain = 1): batches left to skip, and the parameters the count belongs to
  mutable std::atomic<int> gain_skip{0};
  mutable std::atomic<int> gain_run{0};    // 1: the last report of these parameters said the level pays (no guard between batches)
  mutable std::atomic<uint64_t> gain_key{0};
  CodeArr codes() const { return CodeArr{d_codes, code_wide}; }
  CodeArr ucodes() const { return CodeArr{d_ucodes, code_wide}; }
  size_t code_bytes() const { return code_wide ? 4 : 2; }
  // context pool
  mutable std::mutex mu;
  mutable std::condition_variable cv;
  mutable std::vector<Context*> contexts;
  mutable uint64_t use_clock = 0;
};

// Any index the crate can write is searchable up to dim 128: rows are stored zero-padded to the next instantiated width
// (queries are padded with zeros too, so the padded dims add exact zeros to every dot product; norms and outputs stop at
// ldim), and 1-bit residuals are stored as 2-bit ones (bucket b -> segment b << 1, weights {w0, w1, 0, 0}: the same values).
// 8-bit residuals keep their layout and take the all-f32 S6 kernel at every precision.  codec.rs:161-166 accepts nbits
// in {1, 2, 4, 8} and any dim with dim * nbits % 8 == 0.
static inline int storage_dim(int dim) { return dim <= 0 || dim > 128 ? dim : (dim + 31) / 32 * 32; }
static inline int storage_nbits(int dim, int nbits) { return (nbits == 1 && dim <= 128) ? 2 : nbits; }

// np_index.hip
int build_device_index(const HostIndex& h, const np_open_opts* opts, DeviceIndex** out);
void destroy_device_index(DeviceIndex* ix);
int normalise_opts(const np_open_opts* in, np_open_opts* out);
void read_tuning_env(Tuning* t);
bool set_tuning(Tuning* t, const std::string& name, int value);   // shared clamp table (environment + np_hip_index_tune)
void shard_range(int64_t n_total, int rank, int count, int64_t* b, int64_t* e);

// np_search.hip
void destroy_context(Context* c);
// document-sharded exchange on records with a per-rank status trailer (np_dist.hip); the C ABI's np_hip_select_cut /
// np_hip_merge_packed are the status-free cases
int select_cut_strided(const DeviceIndex* ix, const uint64_t* d_all_keys, int64_t rank_stride, int64_t status_off, int G,
                       int B, int n_sel, uint64_t* d_cut, hipStream_t st);
bool select_cut_fits(int G, int n_sel);
int merge_packed_status(const DeviceIndex* ix, const void* d_records, int64_t record_bytes, int64_t off_keys,
                        int64_t off_scores, int64_t off_counts, int64_t off_status, uint64_t* h_status, int G, int B,
                        int top_k, int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts, hipStream_t st);
int set_status_word(const DeviceIndex* ix, uint64_t* d_word, uint64_t value, hipStream_t st);

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace np

struct np_index : np::DeviceIndex {};
