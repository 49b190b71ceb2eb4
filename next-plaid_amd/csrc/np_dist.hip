// np_dist.hip -- document-sharded search below the C ABI: one process per GPU, RCCL all-gathers over xGMI.
//
// The reference has no multi-GPU path.  north_star: "the index shards by document across the 8 GPUs
// of one node with a final RCCL all-gather of per-shard top-k over xGMI".  A compiled host (the Rust crate, a C++
// service) gets the whole protocol from ONE call, np_hip_search_batch_sharded, with no torch in the process:
//
//   [subset only] local eligible-centroid bitmap -> all-gather (K/8 bytes per rank) -> OR      search.rs:350-364
//   phase A   S1-S5 on the local shard -> the shard's n_sel best rank keys                      search.rs:460-469
//   gather 1  ncclAllGather of the keys, [B][n_sel] u64 per rank
//   cut       every rank takes the global n_sel-th key per query (the reference cuts GLOBALLY)
//   phase B   exact MaxSim only for local candidates with key >= cut -> local top-k            search.rs:481-515
//   gather 2  ncclAllGather of one packed record per rank: ids | keys | scores | counts
//   merge     top-k by (exact score desc, approx rank) == the reference's final stable sort
//
// Every buffer is preallocated in the np_comm (grow-only), every step is enqueued on the caller's stream; the two
// payload collectives are <= 0.5 MB per rank, i.e. latency-bound on xGMI.  librccl is dlopen'ed on first use
// (librccl.so.1 resolves to the copy a host process already loaded, e.g. torch's), so the library has no link-time
// dependency on it and single-GPU users never load it.  next_plaid_amd/dist.py keeps the torch.distributed
// harness of the same protocol for the gloo CPU tests.
#include "np_internal.h"

#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <mutex>

namespace np {

// ---- librccl, resolved at run time -------------------------------------------------------------------------------
struct NcclUniqueId {
  char internal[128];   // NCCL_UNIQUE_ID_BYTES
};
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclUint8 = 1 };

struct NcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;   // optional
  std::string err;
};

static NcclApi* nccl_api() {
  static NcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("NEXTPLAID_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
      const char* e = dlerror();   // one call: dlerror() clears the message it returns
      api.err = e ? e : "dlopen failed";
    }
    if (!api.lib) return;
    api.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (int (*)(NcclComm))dlsym(api.lib, "ncclCommDestroy");
    api.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(api.lib, "ncclAllGather");
    api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
    api.CommCount = (int (*)(NcclComm, int*))dlsym(api.lib, "ncclCommCount");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
      api.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      api.lib = nullptr;
    }
  });
  return &api;
}

static int nccl_check(int rc, const char* what) {
  if (rc == kNcclSuccess) return NP_OK;
  NcclApi* a = nccl_api();
  set_error("%s failed: %s", what, a->GetErrorString ? a->GetErrorString(rc) : "RCCL error");
  return NP_ERR_DEVICE_UNAVAILABLE;
}

}  // namespace np

using namespace np;

struct np_comm {
  NcclComm comm = nullptr;
  np_all_gather_host_fn host_fn = nullptr;   // hosted transport (np_hip_comm_create_hosted): the host moves the bytes
  void* host_ctx = nullptr;
  int host_flags = 0;
  int rank = 0, nranks = 1, device = 0;
  DevBuf keys_local, keys_all, cut, pack_local, pack_all, elig_local, elig_all, elig_global;
  char* h_stage = nullptr;       // pinned staging of the hosted transport: [send | recv x nranks]
  size_t h_stage_cap = 0;
  uint64_t* h_status = nullptr;  // pinned, device-visible: the merge kernel leaves a failed rank's status word here
  std::mutex mu;   // one protocol pass at a time per communicator (RCCL orders a communicator's collectives)
};

static int comm_common(const np_index* ix, int32_t rank, int32_t nranks, np_comm** out, const char* who) {
  if (!ix || !out || nranks < 1 || rank < 0 || rank >= nranks) {
    set_error("%s: invalid argument", who);
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (ix->opts.shard_count != nranks || ix->opts.shard_rank != rank) {
    set_error("%s: the handle holds shard %d/%d but the communicator is rank %d/%d", who, ix->opts.shard_rank,
              ix->opts.shard_count, rank, nranks);
    return NP_ERR_INVALID_ARGUMENT;
  }
  np_comm* c = new np_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = ix->device;
  DeviceGuard g(ix->device);
  hipError_t e = hipHostMalloc((void**)&c->h_status, 64, hipHostMallocDefault);
  if (e != hipSuccess) {
    set_error("%s: hipHostMalloc failed: %s", who, hipGetErrorString(e));
    delete c;
    return NP_ERR_OUT_OF_MEMORY;
  }
  *c->h_status = 0;
  *out = c;
  return NP_OK;
}

extern "C" {

int np_hip_comm_unique_id(void* id128) {
  clear_error();
  if (!id128) {
    set_error("comm_unique_id: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  NcclApi* a = nccl_api();
  if (!a->lib) {
    set_error("librccl is not available: %s", a->err.c_str());
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  NcclUniqueId id;
  NP_TRY(nccl_check(a->GetUniqueId(&id), "ncclGetUniqueId"));
  memcpy(id128, id.internal, 128);
  return NP_OK;
}

int np_hip_comm_create(const np_index* ix, const void* id128, int32_t rank, int32_t nranks, np_comm** out) {
  clear_error();
  if (!out) {
    set_error("comm_create: NULL out pointer");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (nranks > 1 && !id128) {
    set_error("comm_create: invalid argument (more than one rank needs the RCCL id)");
    return NP_ERR_INVALID_ARGUMENT;
  }
  np_comm* c = nullptr;
  NP_TRY(comm_common(ix, rank, nranks, &c, "comm_create"));
  if (id128) {   // nranks == 1 with an id still builds a real (single-rank) RCCL communicator
    NcclApi* a = nccl_api();
    if (!a->lib) {
      set_error("librccl is not available: %s", a->err.c_str());
      np_hip_comm_destroy(c);
      return NP_ERR_DEVICE_UNAVAILABLE;
    }
    DeviceGuard g(ix->device);
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    int rc = nccl_check(a->CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");
    if (rc != NP_OK) {
      c->comm = nullptr;
      np_hip_comm_destroy(c);
      return rc;
    }
  }
  *out = c;
  return NP_OK;
}

int np_hip_comm_create_hosted(const np_index* ix, int32_t rank, int32_t nranks, np_all_gather_host_fn all_gather,
                              void* ctx, int32_t flags, np_comm** out) {
  clear_error();
  if (!out) {
    set_error("comm_create_hosted: NULL out pointer");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (!all_gather) {
    set_error("comm_create_hosted: NULL all-gather callback");
    return NP_ERR_INVALID_ARGUMENT;
  }
  np_comm* c = nullptr;
  NP_TRY(comm_common(ix, rank, nranks, &c, "comm_create_hosted"));
  c->host_fn = all_gather;
  c->host_ctx = ctx;
  c->host_flags = flags;
  *out = c;
  return NP_OK;
}

void np_hip_comm_destroy(np_comm* c) {
  if (!c) return;
  {
    DeviceGuard g(c->device);
    if (c->comm) (void)nccl_api()->CommDestroy(c->comm);
    DevBuf* all[] = {&c->keys_local, &c->keys_all, &c->cut, &c->pack_local, &c->pack_all, &c->elig_local, &c->elig_all,
                     &c->elig_global};
    for (DevBuf* b : all) b->release();
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_status) (void)hipHostFree(c->h_status);
  }
  delete c;
}

int np_hip_comm_status(np_comm* c, int32_t* failed_rank, int32_t* code) {
  if (!c || !c->h_status) return NP_ERR_INVALID_ARGUMENT;
  const uint64_t w = __atomic_exchange_n(c->h_status, 0ull, __ATOMIC_ACQ_REL);
  if (failed_rank) *failed_rank = w ? (int32_t)(w >> 32) - 1 : -1;
  if (code) *code = (int32_t)(w & 0xffffffffu);
  return NP_OK;
}

int np_hip_comm_info(np_comm* c, int32_t* transport, int32_t* nranks, int32_t* rccl_ranks) {
  if (!c) return NP_ERR_INVALID_ARGUMENT;
  if (transport) *transport = c->host_fn ? NP_COMM_HOSTED : (c->comm ? NP_COMM_RCCL : NP_COMM_LOCAL);
  if (nranks) *nranks = c->nranks;
  if (rccl_ranks) {
    int n = 0;
    NcclApi* a = c->comm ? nccl_api() : nullptr;
    if (a && a->CommCount && a->CommCount(c->comm, &n) != kNcclSuccess) n = 0;
    *rccl_ranks = n;
  }
  return NP_OK;
}

}  // extern "C"

// send -> recv[nranks][bytes].  RCCL: ncclAllGather on the caller's stream.  Hosted: the bytes go through pinned host
// staging and the host's own transport (the stream is synchronised: the callback is host code).  A communicator with
// neither is the one-rank case (a device copy).  *h_recv (hosted only) = the gathered bytes in host memory.
static int all_gather(np_comm* c, const void* send, void* recv, size_t bytes, hipStream_t st, const char** h_recv = nullptr) {
  if (h_recv) *h_recv = nullptr;
  if (c->host_fn) {
    const size_t need = bytes * ((size_t)c->nranks + 1);
    if (need > c->h_stage_cap) {
      if (c->h_stage) (void)hipHostFree(c->h_stage);
      c->h_stage = nullptr;
      c->h_stage_cap = 0;
      NP_HIP(hipHostMalloc((void**)&c->h_stage, need + 4096, hipHostMallocDefault));
      c->h_stage_cap = need + 4096;
    }
    NP_HIP(hipMemcpyAsync(c->h_stage, send, bytes, hipMemcpyDeviceToHost, st));
    NP_HIP(hipStreamSynchronize(st));
    const int rc = c->host_fn(c->host_ctx, c->h_stage, c->h_stage + bytes, (int64_t)bytes);
    if (rc != 0) {
      set_error("hosted all-gather callback failed with %d", rc);
      return NP_ERR_DEVICE_UNAVAILABLE;
    }
    NP_HIP(hipMemcpyAsync(recv, c->h_stage + bytes, bytes * (size_t)c->nranks, hipMemcpyHostToDevice, st));
    if (h_recv) *h_recv = c->h_stage + bytes;
    return NP_OK;
  }
  if (!c->comm) {
    NP_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st));
    return NP_OK;
  }
  return nccl_check(nccl_api()->AllGather(send, recv, bytes, kNcclUint8, c->comm, st), "ncclAllGather");
}

static uint64_t status_word(int rank, int rc) { return rc == NP_OK ? 0ull : ((uint64_t)(rank + 1) << 32) | (uint32_t)rc; }

// first failed rank among the gathered records (host copy of the hosted transport), or 0
static uint64_t gathered_failure(const char* h_all, size_t rec, size_t off_status, int G) {
  for (int g = 0; g < G; ++g) {
    uint64_t w;
    memcpy(&w, h_all + (size_t)g * rec + off_status, 8);
    if (w) return w;
  }
  return 0;
}

extern "C" int np_hip_search_batch_sharded(const np_index* ix, np_comm* c, const float* d_queries,
                                           const int32_t* d_q_tok_offsets, const int32_t* h_q_tok_offsets, int32_t B,
                                           int32_t dim, const np_search_params* params, const int64_t* d_subset,
                                           int64_t subset_len, int64_t* d_out_ids, float* d_out_scores,
                                           int32_t* d_out_counts, void* stream) {
  clear_error();
  if (!ix || !c || !params || !stream) {
    set_error("search_batch_sharded: NULL index / communicator / params / stream (the collectives need the caller's stream)");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return NP_OK;
  if (B < 0 || !d_out_counts || (params->top_k > 0 && (!d_out_ids || !d_out_scores))) {
    set_error("search_batch_sharded: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  DeviceGuard g(ix->device);
  hipStream_t st = (hipStream_t)stream;
  const int G = c->nranks;
  const int n_sel = np_hip_n_sel(params), ns1 = std::max(n_sel, 1), k = params->top_k, k1 = std::max(k, 1);
  // Failures every rank sees alike (they depend on the arguments only) return before the first collective.
  if (!select_cut_fits(G, n_sel)) {
    set_error("Search failed: %d shards x %d re-rank candidates exceed the 16384-key window of the global cut", G, n_sel);
    return NP_ERR_SEARCH;
  }

  // The protocol never leaves a peer blocked in a collective: every exchange record ends in a status word
  // (0, or np_status | (rank + 1) << 32).  A rank whose LOCAL work fails (out of memory for its workspace, a batch that
  // does not fit its slice, a launch error) still takes part in both all-gathers with empty data and its status; the
  // cut kernel then lets nothing survive, the merge empties every query and leaves the word in pinned host memory
  // (np_hip_comm_status, valid once the stream is synchronised); the failing rank returns its own error at once.  With
  // the hosted transport the gathered bytes pass through the host anyway, so every rank reads the statuses after
  // gather 1 and returns NP_ERR_SEARCH there (flag NP_COMM_DEFERRED_STATUS keeps the device-side propagation).
  // What stays fatal is a failure to reserve the communicator's own few-hundred-KB buffers, before the first collective.
  const bool batched = params->centroid_batch_size > 0 && ix->K > params->centroid_batch_size;
  const bool need_elig = subset_len > 0 && !batched;
  const size_t o_keys = (size_t)B * k1 * 8, o_sc = o_keys * 2, o_cnt = o_sc + (size_t)B * k1 * 4;
  const size_t o_st2 = (o_cnt + (size_t)B * 4 + 15) / 16 * 16, rec2 = o_st2 + 16;
  const size_t o_st1 = (size_t)B * ns1 * 8, rec1 = o_st1 + 16;
  const int64_t words = np_hip_elig_words(ix);
  {
    if (need_elig) {
      NP_TRY(c->elig_local.reserve((size_t)words * 4));
      NP_TRY(c->elig_all.reserve((size_t)G * words * 4));
      NP_TRY(c->elig_global.reserve((size_t)words * 4));
    }
    NP_TRY(c->keys_local.reserve(rec1));
    NP_TRY(c->keys_all.reserve((size_t)G * rec1));
    NP_TRY(c->cut.reserve((size_t)B * 8));
    NP_TRY(c->pack_local.reserve(rec2));
    NP_TRY(c->pack_all.reserve((size_t)G * rec2));
  }
  const bool host_check = c->host_fn && !(c->host_flags & NP_COMM_DEFERRED_STATUS);
  int rc = NP_OK;          // this rank's first local failure
  std::string rc_msg;      // ... and its message (later calls overwrite the thread-local one)
  auto local = [&](int r) {
    if (r != NP_OK && rc == NP_OK) {
      rc = r;
      rc_msg = np_hip_last_error();
    }
    return r == NP_OK;
  };
  auto finish = [&](int r) {
    if (r != NP_OK && !rc_msg.empty()) set_error("%s", rc_msg.c_str());
    return r;
  };

  // ---- eligible centroids of the subset, OR-ed over the shards (dense path only: search.rs:350-364 vs :542-545)
  const uint32_t* elig = nullptr;
  if (need_elig) {
    if (!local(np_hip_subset_eligible(ix, d_subset, subset_len, c->elig_local.as<uint32_t>(), st)))
      (void)hipMemsetAsync(c->elig_local.p, 0, (size_t)words * 4, st);
    NP_TRY(all_gather(c, c->elig_local.p, c->elig_all.p, (size_t)words * 4, st));
    local(np_hip_or_bitmaps(ix, c->elig_all.as<uint32_t>(), G, words, c->elig_global.as<uint32_t>(), st));
    elig = c->elig_global.as<uint32_t>();
  }

  // ---- phase A + gather 1 + cut
  void* state = nullptr;
  if (rc == NP_OK)
    local(np_hip_search_phase_a(ix, d_queries, d_q_tok_offsets, h_q_tok_offsets, B, dim, params, d_subset, subset_len,
                                elig, c->keys_local.as<uint64_t>(), st, &state));
  struct End {
    const np_index* ix;
    void*& s;
    ~End() {
      if (s) np_hip_search_end(ix, s);
    }
  } end{ix, state};
  char* kl = c->keys_local.as<char>();
  if (rc != NP_OK) NP_HIP(hipMemsetAsync(kl, 0, o_st1, st));   // no keys from this rank
  NP_TRY(set_status_word(ix, (uint64_t*)(kl + o_st1), status_word(c->rank, rc), st));
  const char* h_all = nullptr;
  NP_TRY(all_gather(c, kl, c->keys_all.p, rec1, st, &h_all));
  if (host_check && h_all) {
    const uint64_t w = gathered_failure(h_all, rec1, o_st1, G);
    if (w) {   // every rank reads the same bytes: all leave here, none enters gather 2
      if (rc != NP_OK) return finish(rc);
      set_error("Search failed: shard %d failed with status %d; the batch was abandoned on every rank",
                (int)(w >> 32) - 1, (int)(w & 0xffffffffu));
      return NP_ERR_SEARCH;
    }
  }
  if (n_sel > 0)
    local(select_cut_strided(ix, c->keys_all.as<uint64_t>(), (int64_t)(rec1 / 8), (int64_t)(o_st1 / 8), G, B, n_sel,
                             c->cut.as<uint64_t>(), st));

  // ---- phase B into one packed record: ids [B*k] i64 | keys [B*k] u64 | scores [B*k] f32 | counts [B] i32 | status
  char* pl = c->pack_local.as<char>();
  if (rc == NP_OK && state)
    local(np_hip_search_phase_b(ix, state, n_sel > 0 ? c->cut.as<uint64_t>() : nullptr, (int64_t*)pl, (float*)(pl + o_sc),
                                (uint64_t*)(pl + o_keys), (int32_t*)(pl + o_cnt), st));
  if (rc != NP_OK) NP_HIP(hipMemsetAsync(pl, 0, o_st2, st));   // counts 0
  NP_TRY(set_status_word(ix, (uint64_t*)(pl + o_st2), status_word(c->rank, rc), st));
  // ---- gather 2 + merge
  NP_TRY(all_gather(c, pl, c->pack_all.p, rec2, st, &h_all));
  NP_TRY(merge_packed_status(ix, c->pack_all.p, (int64_t)rec2, (int64_t)o_keys, (int64_t)o_sc, (int64_t)o_cnt,
                             (int64_t)o_st2,
                             /* a rank that reports the failure by its return code (its own, or -- hosted transport with the
                                host-side check -- a peer's, below) must not ALSO leave the word behind for the next healthy
                                batch's np_hip_comm_status */
                             (rc == NP_OK && !host_check) ? c->h_status : nullptr, G, B, k,
                             d_out_ids, d_out_scores, d_out_counts, st));
  if (rc != NP_OK) return finish(rc);
  if (host_check && h_all) {
    const uint64_t w = gathered_failure(h_all, rec2, o_st2, G);
    if (w) {
      set_error("Search failed: shard %d failed with status %d; the batch was abandoned on every rank",
                (int)(w >> 32) - 1, (int)(w & 0xffffffffu));
      return NP_ERR_SEARCH;
    }
  }
  return NP_OK;
}
