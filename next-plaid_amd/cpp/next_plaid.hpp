// next_plaid.hpp -- C++ host-side mirror of the next-plaid crate's search API over the C ABI.
//
// The reference host code is Rust (next-plaid/src/index.rs, search.rs, error.rs); this image has no
// Rust toolchain, so the host side above include/nextplaid_hip.h is written in C++ with the SAME
// names, argument meaning and error behaviour.  The Rust shim a maintainer would add is shown in
// INTEGRATION.md; it is a line-for-line analogue of this header.
//
//   next_plaid::MmapIndex::load(path)                      index.rs:1026
//   index.search(query, n_tokens, params, subset)          index.rs:1258  -> QueryResult, query_id = 0
//   index.search_batch(queries, params, parallel, subset)  index.rs:1279  -> query_id = batch position
//   SearchParameters (defaults search.rs:58-69), QueryResult (search.rs:71-80), Error (error.rs:9-66)
//
// Accelerator policy (the crate's precedent for its CUDA feature, lib.rs:71-84 and cuda.rs:52-182):
//   NEXT_PLAID_FORCE_GPU=1|true   a device failure is an Error, never a fallback            (is_force_gpu)
//   NEXT_PLAID_FORCE_CPU=1|true   the HIP library is not touched at all (unless FORCE_GPU)  (is_force_cpu)
//   is_hip_broken / mark_hip_broken / clear_hip_broken: once DeviceUnavailable is seen the flag makes every later call
//   skip the device until it is cleared (CUDA_BROKEN, get_global_context's fast path).  OutOfMemory hands THAT call (or
//   that index) to the CPU hook without raising the process-wide flag: one oversized index must not take the device away
//   from every other index of the process.
// The CPU implementation itself is the crate's existing Rust path; here it is a hook (set_cpu_fallback) that the
// DeviceUnavailable hand-off calls -- this header ships NO CPU search of its own, and with no hook installed a
// device failure stays an Error (nothing is papered over).
//
// Header-only; link with -lnextplaid_hip.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nextplaid_hip.h"

namespace next_plaid {

struct Error : std::runtime_error {  // error.rs:9-66
  enum Kind { IndexLoad = 1, Search = 2, Shape = 3, Codec = 4, Io = 5, DeviceUnavailable = 6, OutOfMemory = 7, Config = 8 };
  Kind kind;
  Error(int code, const char* msg) : std::runtime_error(msg && *msg ? msg : "next-plaid error"), kind((Kind)code) {}
};

inline void check(int rc) {
  if (rc != NP_OK) throw Error(rc, np_hip_last_error());
}

// ---- accelerator policy: lib.rs:71-84 -----------------------------------------------------------------------
inline bool env_flag(const char* name) {
  const char* v = std::getenv(name);
  if (!v) return false;
  if (std::strcmp(v, "1") == 0) return true;
  std::string s(v);
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  return s == "true";
}
inline bool is_force_gpu() { return env_flag("NEXT_PLAID_FORCE_GPU"); }
inline bool is_force_cpu() { return !is_force_gpu() && env_flag("NEXT_PLAID_FORCE_CPU"); }

// ---- broken flag: cuda.rs:52-182 ----------------------------------------------------------------------------------
inline std::atomic<bool>& hip_broken_flag() {
  static std::atomic<bool> f{false};
  return f;
}
inline bool is_hip_broken() { return hip_broken_flag().load(std::memory_order_relaxed); }
inline void mark_hip_broken() { hip_broken_flag().store(true, std::memory_order_relaxed); }
inline void clear_hip_broken() { hip_broken_flag().store(false, std::memory_order_relaxed); }
inline bool is_device_failure(int rc) { return rc == NP_ERR_DEVICE_UNAVAILABLE || rc == NP_ERR_OUT_OF_MEMORY; }

struct SearchParameters {  // search.rs:26-69
  size_t batch_size = 2000;
  size_t n_full_scores = 4096;
  size_t top_k = 10;
  size_t n_ivf_probe = 8;
  size_t centroid_batch_size = 100000;
  std::optional<float> centroid_score_threshold = 0.4f;
  int precision = 2;  // exact-MaxSim arithmetic (np_search_params.precision): 2 = split-bf16 QC-reuse, f32-class (default)
  np_search_params c() const {
    np_search_params p{};
    p.top_k = (int32_t)top_k;
    p.n_full_scores = (int32_t)n_full_scores;
    p.n_ivf_probe = (int32_t)n_ivf_probe;
    p.centroid_batch_size = (int32_t)centroid_batch_size;
    p.centroid_score_threshold = centroid_score_threshold.value_or(0.f);
    p.has_threshold = centroid_score_threshold.has_value() ? 1 : 0;
    p.precision = precision;
    return p;
  }
};

struct QueryResult {  // search.rs:71-80
  size_t query_id = 0;
  std::vector<int64_t> passage_ids;
  std::vector<float> scores;
};
using SearchResult = QueryResult;  // search.rs:678

// A row-major [n_tokens, dim] f32 query (ndarray::Array2<f32> in the crate).
struct Query {
  const float* data;
  size_t n_tokens;
};

// The CPU path a device failure hands off to (in the crate: search::search_many_mmap on the mmap'ed index).
using CpuSearchFn = std::function<std::vector<QueryResult>(const std::string& index_path, const Query* queries, size_t n,
                                                           size_t dim, const SearchParameters& params, bool parallel,
                                                           const std::vector<int64_t>* subset)>;
inline CpuSearchFn& cpu_fallback() {
  static CpuSearchFn f;
  return f;
}
inline void set_cpu_fallback(CpuSearchFn f) { cpu_fallback() = std::move(f); }

class MmapIndex {
 public:
  // MmapIndex::load (index.rs:1026).  `opts` selects the device / document shard.
  // Policy: FORCE_CPU or a raised broken flag never touch the device (the handle stays empty and searches go to the
  // CPU hook); a device failure raises the flag and falls back unless FORCE_GPU; every other error is the caller's.
  static MmapIndex load(const std::string& index_path, const np_open_opts* opts = nullptr) {
    if ((is_force_cpu() || (is_hip_broken() && !is_force_gpu())) && cpu_fallback()) return MmapIndex(nullptr, index_path);
    check_abi();
    np_index* h = nullptr;
    const int rc = np_hip_index_open(index_path.c_str(), opts, &h);
    if (is_device_failure(rc)) {
      if (rc == NP_ERR_DEVICE_UNAVAILABLE) mark_hip_broken();   // OutOfMemory concerns this index only
      if (!is_force_gpu() && cpu_fallback()) {
        std::fprintf(stderr, "[next-plaid] HIP device unavailable: %s. Falling back to CPU. Set NEXT_PLAID_FORCE_CPU=1 to "
                             "skip the GPU and silence this warning.\n", np_hip_last_error());
        return MmapIndex(nullptr, index_path);
      }
    }
    check(rc);
    return MmapIndex(h, index_path);
  }
  // The library this binary RUNS against must speak the header it was COMPILED against: np_info / np_stats are allocated here and
  // have grown between ABI versions (np_hip_abi_version exists since v6; an older library fails at link / load time already).
  static void check_abi() {
    if (np_hip_abi_version() != NP_ABI_VERSION || np_hip_struct_size(0) != (int64_t)sizeof(np_info) ||
        np_hip_struct_size(1) != (int64_t)sizeof(np_stats))
      throw std::runtime_error("libnextplaid_hip speaks ABI v" + std::to_string(np_hip_abi_version()) + ", this host was built against v" +
                               std::to_string(NP_ABI_VERSION));
  }
  // MmapIndex::reload (index.rs:1767-1775): after delete / update rewrote the directory.  The crate releases its maps before
  // it loads again; here the device copy is dropped first for the same reason (two copies of a 200 GB index do not fit).
  // Exclusive access like `&mut self`; a service swaps handles.  Same device policy as load().
  void reload(const np_open_opts* opts = nullptr) {
    MmapIndex fresh = (close(), load(path, opts));
    *this = std::move(fresh);
  }
  bool on_device() const { return h_ != nullptr; }
  MmapIndex(MmapIndex&& o) noexcept : path(std::move(o.path)), h_(o.h_), info_(o.info_) { o.h_ = nullptr; }
  MmapIndex& operator=(MmapIndex&& o) noexcept {
    if (this != &o) {
      close();
      h_ = o.h_;
      o.h_ = nullptr;
      path = std::move(o.path);
      info_ = o.info_;
    }
    return *this;
  }
  MmapIndex(const MmapIndex&) = delete;
  MmapIndex& operator=(const MmapIndex&) = delete;
  ~MmapIndex() { close(); }

  // index.rs:1258-1265
  QueryResult search(const float* query, size_t n_tokens, const SearchParameters& params,
                     const std::vector<int64_t>* subset = nullptr) const {
    Query q{query, n_tokens};
    auto r = search_batch(&q, 1, params, /*parallel=*/false, subset);
    r[0].query_id = 0;
    return std::move(r[0]);
  }

  // index.rs:1279-1287 -> search.rs:643-675.  `parallel` keeps the reference's error policy: with
  // parallel = true a failing search yields empty results instead of an error (search.rs:656-660).
  std::vector<QueryResult> search_batch(const Query* queries, size_t n, const SearchParameters& params, bool parallel,
                                        const std::vector<int64_t>* subset = nullptr) const {
    if (!h_) return cpu_search(queries, n, params, parallel, subset);
    const size_t dim = embedding_dim();
    std::vector<int32_t> off(n + 1, 0);
    for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + (int32_t)queries[i].n_tokens;
    std::vector<float> flat((size_t)off[n] * dim);
    for (size_t i = 0; i < n; ++i)
      std::copy(queries[i].data, queries[i].data + queries[i].n_tokens * dim, flat.begin() + (size_t)off[i] * dim);
    const size_t k = params.top_k;
    std::vector<int64_t> ids(std::max<size_t>(n * k, 1));
    std::vector<float> sc(std::max<size_t>(n * k, 1));
    std::vector<int32_t> cnt(std::max<size_t>(n, 1));
    np_search_params p = params.c();
    int rc = np_hip_search_batch(h_, flat.data(), off.data(), (int32_t)n, (int32_t)dim, &p,
                                 subset ? subset->data() : nullptr, subset ? (int64_t)subset->size() : -1, ids.data(),
                                 sc.data(), cnt.data(), &last_stats);
    std::vector<QueryResult> out(n);
    for (size_t i = 0; i < n; ++i) out[i].query_id = i;
    if (rc != NP_OK) {
      if (is_device_failure(rc)) {   // mid-flight device loss: flag it, hand this call to the CPU unless FORCE_GPU
        if (rc == NP_ERR_DEVICE_UNAVAILABLE) mark_hip_broken();   // an OutOfMemory of one call leaves the device usable
        if (!is_force_gpu() && cpu_fallback()) return cpu_search(queries, n, params, parallel, subset);
      }
      if (parallel && rc == NP_ERR_SEARCH) return out;
      check(rc);
    }
    for (size_t i = 0; i < n; ++i) {
      // ABI v5: a negative count (NP_COUNT_ABANDONED) marks a batch a sharded peer abandoned -- never a length
      if (cnt[i] < 0) throw Error(NP_ERR_SEARCH, "Search failed: the batch was abandoned (a peer shard failed)");
      out[i].passage_ids.assign(ids.begin() + i * k, ids.begin() + i * k + cnt[i]);
      out[i].scores.assign(sc.begin() + i * k, sc.begin() + i * k + cnt[i]);
    }
    return out;
  }

  // index.rs:1197-1245 decompress_documents: (embeddings [sum len, dim], lengths)
  std::pair<std::vector<float>, std::vector<int64_t>> decompress_documents(const std::vector<int64_t>& doc_ids) const {
    require_device("decompress_documents");
    std::vector<int64_t> lens(std::max<size_t>(doc_ids.size(), 1));
    check(np_hip_decompress_documents(h_, doc_ids.data(), (int64_t)doc_ids.size(), nullptr, 0, lens.data()));
    lens.resize(doc_ids.size());
    int64_t total = 0;
    for (int64_t l : lens) total += l;
    std::vector<float> emb((size_t)std::max<int64_t>(total, 1) * embedding_dim());
    check(np_hip_decompress_documents(h_, doc_ids.data(), (int64_t)doc_ids.size(), emb.data(), total, lens.data()));
    emb.resize((size_t)total * embedding_dim());
    return {std::move(emb), std::move(lens)};
  }

  // index.rs:289-371 encode_index_chunk for a flat [n, dim] batch: (codes, packed residuals [n, dim*nbits/8])
  std::pair<std::vector<int64_t>, std::vector<uint8_t>> encode_tokens(const float* embeddings, size_t n,
                                                                      const std::vector<float>& bucket_cutoffs) const {
    require_device("encode_tokens");
    const size_t pd = embedding_dim() * (size_t)info_.nbits / 8;
    std::vector<int64_t> codes(std::max<size_t>(n, 1));
    std::vector<uint8_t> packed(std::max<size_t>(n * pd, 1));
    if (bucket_cutoffs.size() + 1 != ((size_t)1 << info_.nbits)) throw Error(NP_ERR_CODEC, "Codec error: bucket_cutoffs size");
    check(np_hip_encode_tokens(h_, embeddings, (int64_t)n, (int32_t)embedding_dim(), bucket_cutoffs.data(), codes.data(),
                               packed.data()));
    codes.resize(n);
    packed.resize(n * pd);
    return {std::move(codes), std::move(packed)};
  }

  // index.rs:1290-1312
  size_t num_documents() const { return (size_t)info_.num_documents; }
  size_t num_embeddings() const { return (size_t)info_.num_embeddings; }
  size_t num_partitions() const { return (size_t)info_.num_partitions; }
  double avg_doclen() const { return info_.avg_doclen; }
  size_t embedding_dim() const { return (size_t)info_.embedding_dim; }
  const np_info& info() const { return info_; }
  np_index* handle() const { return h_; }

  std::string path;
  mutable np_stats last_stats{};
  size_t cpu_dim = 0;   // overrides the embedding dim handed to the CPU hook (0 = the index's own, from metadata)

 private:
  MmapIndex(np_index* h, std::string p) : path(std::move(p)), h_(h) {
    if (h_) check(np_hip_index_info(h_, &info_));
    // CPU hand-off mode: the geometry accessors (index.rs:1290-1312) stay valid -- host-only parse of the same directory
    else check(np_hip_index_probe_dir(path.c_str(), &info_));
  }
  void require_device(const char* what) const {
    if (!h_)
      throw Error(NP_ERR_DEVICE_UNAVAILABLE, (std::string(what) + ": this index runs on the CPU hand-off (no device handle)").c_str());
  }
  std::vector<QueryResult> cpu_search(const Query* queries, size_t n, const SearchParameters& params, bool parallel,
                                      const std::vector<int64_t>* subset) const {
    if (!cpu_fallback()) throw Error(NP_ERR_DEVICE_UNAVAILABLE, "HIP device unavailable and no CPU fallback installed");
    return cpu_fallback()(path, queries, n, cpu_dim ? cpu_dim : embedding_dim(), params, parallel, subset);
  }
  void close() {
    if (h_) np_hip_index_close(h_);
    h_ = nullptr;
  }
  np_index* h_ = nullptr;
  np_info info_{};
};

// index.rs:373-528 write_index_from_encoded_chunks: encoded chunks (flat here) -> an index directory in the crate's
// on-disk format.  Host only.  Posting lists are built from the codes (index.rs:479-504).
struct IndexFiles {
  size_t num_centroids = 0, dim = 0;
  int nbits = 4;
  const float* centroids = nullptr;            // [K, dim]
  std::vector<float> bucket_weights;           // [2^nbits]
  std::vector<float> bucket_cutoffs;           // [2^nbits - 1] or empty
  std::vector<float> avg_residual;             // [dim] or empty (zeros)
  float cluster_threshold = 0.f;
  std::vector<int64_t> doc_lengths;            // [N]
  const int64_t* codes = nullptr;              // [sum doc_lengths]
  const uint8_t* residuals = nullptr;          // [sum doc_lengths, dim * nbits / 8]
  size_t chunk_docs = 50000;                   // IndexConfig.batch_size (index.rs:92)
};
inline void write_index(const std::string& path, const IndexFiles& f) {
  if (f.bucket_weights.size() != ((size_t)1 << f.nbits)) throw Error(NP_ERR_CODEC, "Codec error: bucket_weights size");
  if (!f.bucket_cutoffs.empty() && f.bucket_cutoffs.size() + 1 != ((size_t)1 << f.nbits))
    throw Error(NP_ERR_CODEC, "Codec error: bucket_cutoffs size");
  if (!f.avg_residual.empty() && f.avg_residual.size() != f.dim) throw Error(NP_ERR_SHAPE, "Shape error: avg_residual size");
  np_index_arrays a{};
  a.num_documents_total = a.num_docs = (int64_t)f.doc_lengths.size();
  a.num_centroids = (int64_t)f.num_centroids;
  a.dim = (int32_t)f.dim;
  a.nbits = f.nbits;
  a.centroids = f.centroids;
  a.bucket_weights = f.bucket_weights.data();
  a.doc_lengths = f.doc_lengths.data();
  a.codes = f.codes;
  a.residuals = f.residuals;
  np_write_opts o{};
  o.chunk_docs = (int64_t)f.chunk_docs;
  o.bucket_cutoffs = f.bucket_cutoffs.empty() ? nullptr : f.bucket_cutoffs.data();
  o.avg_residual = f.avg_residual.empty() ? nullptr : f.avg_residual.data();
  o.cluster_threshold = f.cluster_threshold;
  check(np_hip_index_write_dir(path.c_str(), &a, &o));
}

// next-plaid-api handlers/rerank.rs:57-170: MaxSim of one query against caller-supplied document embeddings
// (document i = rows doc_tok_offsets[i] .. doc_tok_offsets[i+1] of `docs`).  Returns (order, scores).
inline std::pair<std::vector<int64_t>, std::vector<float>> rerank_maxsim(const float* query, size_t n_query_tokens,
                                                                         size_t dim, const float* docs,
                                                                         const std::vector<int64_t>& doc_tok_offsets,
                                                                         int device = 0) {
  const size_t n = doc_tok_offsets.empty() ? 0 : doc_tok_offsets.size() - 1;
  std::vector<float> scores(std::max<size_t>(n, 1));
  std::vector<int64_t> order(std::max<size_t>(n, 1));
  check(np_hip_rerank_maxsim(device, query, (int32_t)n_query_tokens, (int32_t)dim, docs, doc_tok_offsets.data(), (int64_t)n,
                             scores.data(), order.data()));
  scores.resize(n);
  order.resize(n);
  return {std::move(order), std::move(scores)};
}

}  // namespace next_plaid
