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
  int rank = 0, nranks = 1, device = 0;
  DevBuf keys_local, keys_all, cut, pack_local, pack_all, elig_local, elig_all, elig_global;
  std::mutex mu;   // one protocol pass at a time per communicator (RCCL orders a communicator's collectives)
};

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
  if (!ix || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id128)) {
    set_error("comm_create: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (ix->opts.shard_count != nranks || ix->opts.shard_rank != rank) {
    set_error("comm_create: the handle holds shard %d/%d but the communicator is rank %d/%d", ix->opts.shard_rank,
              ix->opts.shard_count, rank, nranks);
    return NP_ERR_INVALID_ARGUMENT;
  }
  np_comm* c = new np_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = ix->device;
  if (id128) {   // nranks == 1 with an id still builds a real (single-rank) RCCL communicator
    NcclApi* a = nccl_api();
    if (!a->lib) {
      set_error("librccl is not available: %s", a->err.c_str());
      delete c;
      return NP_ERR_DEVICE_UNAVAILABLE;
    }
    DeviceGuard g(ix->device);
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    int rc = nccl_check(a->CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");
    if (rc != NP_OK) {
      delete c;
      return rc;
    }
  }
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
  }
  delete c;
}

// send -> recv[nranks][bytes]; a communicator without an RCCL handle is the one-rank case (a device copy)
static int all_gather(np_comm* c, const void* send, void* recv, size_t bytes, hipStream_t st) {
  if (!c->comm) {
    NP_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st));
    return NP_OK;
  }
  return nccl_check(nccl_api()->AllGather(send, recv, bytes, kNcclUint8, c->comm, st), "ncclAllGather");
}

int np_hip_search_batch_sharded(const np_index* ix, np_comm* c, const float* d_queries, const int32_t* d_q_tok_offsets,
                                const int32_t* h_q_tok_offsets, int32_t B, int32_t dim, const np_search_params* params,
                                const int64_t* d_subset, int64_t subset_len, int64_t* d_out_ids, float* d_out_scores,
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

  // Every allocation of the communicator happens BEFORE the first collective: a rank that returned on an allocation
  // failure between two all-gathers would leave its peers blocked in RCCL.  (Phase A's own workspace is reserved at its
  // start, ahead of gather 1; an error there is reported to the caller, who must treat the communicator as poisoned --
  // np_hip_comm_destroy and re-create -- because the peers' gather cannot complete.)
  const bool batched = params->centroid_batch_size > 0 && ix->K > params->centroid_batch_size;
  const bool need_elig = subset_len > 0 && !batched;
  const size_t o_keys = (size_t)B * k1 * 8, o_sc = o_keys * 2, o_cnt = o_sc + (size_t)B * k1 * 4;
  const size_t rec = (o_cnt + (size_t)B * 4 + 15) / 16 * 16;
  {
    const int64_t words = np_hip_elig_words(ix);
    if (need_elig) {
      NP_TRY(c->elig_local.reserve((size_t)words * 4));
      NP_TRY(c->elig_all.reserve((size_t)G * words * 4));
      NP_TRY(c->elig_global.reserve((size_t)words * 4));
    }
    NP_TRY(c->keys_local.reserve((size_t)B * ns1 * 8));
    NP_TRY(c->keys_all.reserve((size_t)G * B * ns1 * 8));
    NP_TRY(c->cut.reserve((size_t)B * 8));
    NP_TRY(c->pack_local.reserve(rec));
    NP_TRY(c->pack_all.reserve((size_t)G * rec));
  }

  // ---- eligible centroids of the subset, OR-ed over the shards (dense path only: search.rs:350-364 vs :542-545)
  const uint32_t* elig = nullptr;
  if (need_elig) {
    const int64_t words = np_hip_elig_words(ix);
    NP_TRY(np_hip_subset_eligible(ix, d_subset, subset_len, c->elig_local.as<uint32_t>(), st));
    NP_TRY(all_gather(c, c->elig_local.p, c->elig_all.p, (size_t)words * 4, st));
    NP_TRY(np_hip_or_bitmaps(ix, c->elig_all.as<uint32_t>(), G, words, c->elig_global.as<uint32_t>(), st));
    elig = c->elig_global.as<uint32_t>();
  }

  // ---- phase A + gather 1 + cut
  void* state = nullptr;
  NP_TRY(np_hip_search_phase_a(ix, d_queries, d_q_tok_offsets, h_q_tok_offsets, B, dim, params, d_subset, subset_len,
                               elig, c->keys_local.as<uint64_t>(), st, &state));
  struct End {
    const np_index* ix;
    void* s;
    ~End() { np_hip_search_end(ix, s); }
  } end{ix, state};
  if (n_sel > 0) {
    NP_TRY(all_gather(c, c->keys_local.p, c->keys_all.p, (size_t)B * n_sel * 8, st));
    NP_TRY(np_hip_select_cut(ix, c->keys_all.as<uint64_t>(), G, B, n_sel, c->cut.as<uint64_t>(), st));
  }

  // ---- phase B into one packed record: ids [B*k] i64 | keys [B*k] u64 | scores [B*k] f32 | counts [B] i32
  char* pl = c->pack_local.as<char>();
  NP_TRY(np_hip_search_phase_b(ix, state, n_sel > 0 ? c->cut.as<uint64_t>() : nullptr, (int64_t*)pl, (float*)(pl + o_sc),
                               (uint64_t*)(pl + o_keys), (int32_t*)(pl + o_cnt), st));
  // ---- gather 2 + merge
  NP_TRY(all_gather(c, pl, c->pack_all.p, rec, st));
  const char* pa = c->pack_all.as<char>();
  NP_TRY(np_hip_merge_packed(ix, pa, (int64_t)rec, (int64_t)o_keys, (int64_t)o_sc, (int64_t)o_cnt, G, B, k, d_out_ids,
                             d_out_scores, d_out_counts, st));
  return NP_OK;
}

}  // extern "C"
