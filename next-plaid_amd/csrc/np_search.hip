// np_search.hip -- the search pipeline and its C ABI.
//
// Replaces search::search_one_mmap / search_many_mmap (next-plaid/src/search.rs:327-675) behind
// MmapIndex::search / search_batch (index.rs:1258-1287).  A batch of B queries is ONE pass of
// S1..S7 launches on one HIP stream (no host round trip between stages; every data-dependent
// size lives in device memory and kernels early-exit on it).
#include "np_internal.h"
#include "np_kernels.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <stdlib.h>
#include <string.h>

namespace np {

struct Workspace {
  DevBuf q, qoff, Qt, Qb, Qbl, QCT, gmax, tauq, cellbits, cells_tmp, cells, n_cells, docbits, chunk_counts, cand, cand_meta, approx, n_cand,
      cand_base, round_of, round_tab, QCU, qinv, qflag, ub, ub_hist, ub_thr, ub_cursor, q_order, xcd_slots, surv_meta, n_surv, n_list2, sel_keys, sel_doc, nsel, exact, out_ids, out_scores, out_keys, out_counts, ctr, subset,
      subset_bits, elig, misc, cut, cmaxu, chist, ub2, ub_hist2, ub_thr2, list_meta, n_l1, n_l2, qpad, planes, levels, hotbits,
      gain, gsmall, ghist, s0_meta, s0_u, gacc, gdeep;   // zeroth filter level (gain_sweep_kernel)
  void* h_pin = nullptr;
  size_t h_pin_cap = 0;
  unsigned long long* h_gain = nullptr;   // pinned: the zeroth level's last (candidates << 32 | kept), written by the device
  uint64_t h_gain_key = 0;                // ... and the parameters of the batch that will write (or wrote) it
  hipEvent_t done = nullptr;  // recorded at the end of every use of this workspace
  bool done_valid = false;
  static constexpr int NBUF = 66;
  std::array<DevBuf*, NBUF> all_bufs() {   // no heap allocation: total_bytes() runs on the search path
    return {&q, &qoff, &Qt, &Qb, &Qbl, &QCT, &gmax, &tauq, &cellbits, &cells_tmp, &cells, &n_cells, &docbits, &chunk_counts, &cand, &cand_meta, &approx, &n_cand, &cand_base, &round_of, &round_tab, &QCU, &qinv, &qflag, &ub, &ub_hist, &ub_thr, &ub_cursor, &q_order, &xcd_slots, &surv_meta, &n_surv, &n_list2, &sel_keys, &sel_doc, &nsel, &exact, &out_ids, &out_scores, &out_keys, &out_counts, &ctr, &subset, &subset_bits, &elig, &misc, &cut, &cmaxu, &chist, &ub2, &ub_hist2, &ub_thr2, &list_meta, &n_l1, &n_l2, &qpad, &planes, &levels, &hotbits, &gain, &gsmall, &ghist, &s0_meta, &s0_u, &gacc, &gdeep};
  }
  void release_all() {
    for (DevBuf* b : all_bufs()) b->release();
    if (h_pin) (void)hipHostFree(h_pin);
    h_pin = nullptr;
    h_pin_cap = 0;
    if (h_gain) (void)hipHostFree(h_gain);
    h_gain = nullptr;
    if (done) (void)hipEventDestroy(done);
    done = nullptr;
  }
  size_t total_bytes() {   // everything this workspace holds on the device
    size_t t = 0;
    for (DevBuf* b : all_bufs()) t += b->cap;
    return t;
  }
  unsigned probe_tick = 0;   // rate limit of the free-memory probe while the budget stands below its value at open
  size_t pool_bytes() const {   // the candidate pool and its companions (sized by the budget)
    return cand.cap + cand_meta.cap + approx.cap + ub.cap + surv_meta.cap + ub2.cap + list_meta.cap;
  }
  void release_pool() {
    DevBuf* pool[] = {&cand, &cand_meta, &approx, &ub, &surv_meta, &ub2, &list_meta};
    for (DevBuf* b : pool) b->release();
  }
  int pin(size_t bytes) {
    if (bytes <= h_pin_cap) return NP_OK;
    if (h_pin) (void)hipHostFree(h_pin);
    h_pin = nullptr;
    h_pin_cap = 0;
    hipError_t e = hipHostMalloc(&h_pin, bytes + 4096, hipHostMallocDefault);
    if (e != hipSuccess) {
      set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return NP_ERR_OUT_OF_MEMORY;
    }
    h_pin_cap = bytes + 4096;
    return NP_OK;
  }
};

void destroy_context(Context* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->ws) {
    c->ws->release_all();
    delete c->ws;
  }
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int acquire_context(const DeviceIndex* ix, Context** out) {
  std::unique_lock<std::mutex> lk(ix->mu);
  for (;;) {
    // Prefer growing the pool, then the least recently used idle context: back-to-back device-side calls
    // on different streams then land on different workspaces and their kernels can overlap on the GPU
    // (a context's workspace is guarded by its `done` event, so reuse is always safe, just serialising).
    Context* best = nullptr;
    for (Context* c : ix->contexts)
      if (!c->busy && (!best || c->last_use < best->last_use)) best = c;
    if (best && (int)ix->contexts.size() >= ix->opts.n_contexts) {
      best->busy = true;
      best->last_use = ++ix->use_clock;
      *out = best;
      return NP_OK;
    }
    if ((int)ix->contexts.size() < ix->opts.n_contexts) {
      Context* c = new Context();
      c->ws = new Workspace();
      hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
      for (auto& ev : c->ev)
        if (e == hipSuccess) e = hipEventCreate(&ev);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ws->done, hipEventDisableTiming);
      if (e != hipSuccess) {
        set_error("context creation failed: %s", hipGetErrorString(e));
        destroy_context(c);
        return NP_ERR_DEVICE_UNAVAILABLE;
      }
      c->busy = true;
      c->last_use = ++ix->use_clock;
      ix->contexts.push_back(c);
      *out = c;
      return NP_OK;
    }
    ix->cv.wait(lk);
  }
}

static void release_context(const DeviceIndex* ix, Context* c) {
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    c->busy = false;
  }
  ix->cv.notify_one();
}

// ---- one pipeline pass over a slice of the batch --------------------------------------------------
struct CallState {
  Context* ctx = nullptr;
  hipStream_t stream = nullptr;
  int B = 0, LQP = 0, n_sel = 0, NSELP = 1;
  np_search_params prm{};
  bool empty_subset = false;
  bool timed = false;
  bool hot_timed = false;   // ev[8] .. ev[9] bracket the first filter level of round 0 (np_stats.ms_hot_level)
  bool trace = false;   // debug_trace: every candidate keeps its exact approximate score
  const uint32_t* elig_global = nullptr;   // sharded + subset: eligible-centroid bitmap OR-ed over all shards (search.rs:350-364)
};

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

static int n_sel_of(const np_search_params* p) {
  int64_t nd = std::max<int64_t>((int64_t)p->n_full_scores / 4, p->top_k);  // search.rs:468
  return (int)std::min<int64_t>(nd, p->n_full_scores);                     // search.rs:461-469 take/take
}

// STORAGE geometry (np_internal.h storage_dim / storage_nbits): every index with dim <= 128 lands here
static bool dim_supported(int dim, int nbits) {
  return (dim == 32 || dim == 64 || dim == 96 || dim == 128) && (nbits == 2 || nbits == 4 || nbits == 8);
}

static int validate(const DeviceIndex* ix, int32_t B, int32_t dim, const np_search_params* p) {
  if (!ix || !p) {
    set_error("Search failed: NULL index or params");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B < 0) {
    set_error("Search failed: negative batch size");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (dim != ix->ldim) {  // ndarray .dot() would panic on this (search.rs:345)
    set_error("Shape error: query dim %d does not match index dim %d", dim, ix->ldim);
    return NP_ERR_SHAPE;
  }
  if (!dim_supported(ix->dim, ix->nbits)) {
    set_error("Shape error: the HIP search path supports dim <= 128; index has dim=%d nbits=%d", ix->ldim, ix->lnbits);
    return NP_ERR_SHAPE;
  }
  if (p->n_ivf_probe < 1 || p->top_k < 0 || p->n_full_scores < 0) {
    set_error("Search failed: invalid parameters (n_ivf_probe=%d top_k=%d n_full_scores=%d)", p->n_ivf_probe, p->top_k,
              p->n_full_scores);
    return NP_ERR_SEARCH;
  }
  // S5 / S7 order a query's re-rank window in LDS (8 bytes per document, 128 KB of the CU's 160): n_full_scores up to 65536
  if (n_sel_of(p) > 16384) {
    set_error("Search failed: max(n_full_scores/4, top_k) = %d exceeds the HIP path's 16384-document re-rank window",
              n_sel_of(p));
    return NP_ERR_SEARCH;
  }
  if (p->precision < 0 || p->precision > 3) {
    set_error("Search failed: unknown precision %d", p->precision);
    return NP_ERR_INVALID_ARGUMENT;
  }
  return NP_OK;
}

// ---- workspace plan: one expression set for slicing AND for the reserve() calls ---------------------------------
// Per-query scratch that scales with the batch (score table, probe bitmaps, doc bitmap, selection) and the
// candidate pool (NP_POOL_ENTRY bytes per entry: doc id + 16-B record + approximate score + two u16 bounds + 16-B list
// record of the two-level filter + 16-B survivor record), which is sized by the
// budget, not by n_docs: B x n_docs entries only when that fits workspace_bytes, otherwise what is left of the
// budget after the per-query scratch (never less than 2 x n_docs entries, one query's worst case twice).
#define NP_POOL_ENTRY 60
struct WsPlan {
  int S = 1;            // queries per slice
  int64_t pool = 1;     // candidate-pool entries
  int max_rounds = 1;   // rounds the host enqueues for one slice (worst case; extra rounds exit immediately)
};

static int64_t per_query_bytes(const DeviceIndex* ix, int LQP, int n_sel, int top_k) {
  const int64_t KP = ix->KP, G = KP / 32, NW = (ix->n_docs + 31) / 32;
  const int64_t nchunks = (NW + NP_CHUNK_WORDS - 1) / NP_CHUNK_WORDS;
  return KP * LQP * 6                      // QCT (f32) + QCU (u8, rows padded to a power of two)
         + KP + 1024                       // per-centroid maxima of the u8 table + their histogram (hot level)
         + KP * 6 + G * 4 + NP_UB_BINS * 8   // zeroth level: gains of the probed cells, its own deeper cell list, two histograms,
         + (ix->d_ivf_split ? (int64_t)ix->n_ranges * NP_GAIN_RANGE : 0)   // ... the documents' level bytes
         + NP_UB_BINS * 8
         + G * LQP * 4 + G * 4             // gmax, cellbits
         + KP * 8                          // cells_tmp, cells
         + (int64_t)LQP * 4                // tauq
         + std::max<int64_t>(NW, 1) * 4    // docbits
         + std::max<int64_t>(nchunks, 1) * 4
         + (int64_t)ix->dim * LQP * 8      // Qt, Qb, Qbl
         + (int64_t)std::max(n_sel, 1) * 16 + (int64_t)std::max(top_k, 1) * 20 + 64;
}

// `probed_cells`: the most cells one query of the batch can take candidates from (n_ivf_probe x its tokens), 0 = unknown (a
// subset scales n_ivf_probe on the device, search.rs:350-382): the pool and its rounds are then planned for n_docs per query.
static WsPlan plan_workspace(const DeviceIndex* ix, int B, int LQP, const np_search_params* prm, int64_t probed_cells = 0) {
  WsPlan w;
  const int64_t budget = ix->ws_budget.load(std::memory_order_relaxed);
  const int64_t pq = std::max<int64_t>(per_query_bytes(ix, LQP, n_sel_of(prm), prm->top_k), 1);
  const int64_t nd = std::max<int64_t>(ix->n_docs, 1);
  const int cap = (int)std::min<int64_t>(std::min<int64_t>(ix->opts.max_batch, NP_S4_MAXB), std::max(B, 1));
  int64_t S = std::min<int64_t>(cap, std::max<int64_t>(1, (budget * 3 / 4) / pq));   // keep >= 1/4 of the budget for the pool
  // if every query's worst case fits next to the scratch, take it (no rounds at all)
  while (S > 1 && S * pq + 2 * nd * NP_POOL_ENTRY > budget) --S;
  w.S = (int)S;
  int64_t worst_q = nd;      // candidates of one query at most: the probed cells' lists, were they the longest of the index
  if (probed_cells > 0 && !ix->ivf_top_prefix.empty())
    worst_q = std::max<int64_t>(1, std::min(nd, ix->ivf_top_prefix[(size_t)std::min<int64_t>(probed_cells, ix->K)]));
  const int64_t worst = S * worst_q;
  int64_t pool = (budget - S * pq) / NP_POOL_ENTRY;
  pool = std::min(worst, std::max(pool, std::min<int64_t>(2, S) * worst_q));
  w.pool = std::max<int64_t>(pool, 1);
  // first-fit packing in query order: every closed round holds more than pool - worst_q entries
  w.max_rounds = (worst <= w.pool) ? 1 : (int)std::min<int64_t>(S, worst / (w.pool - worst_q + 1) + 1);
  return w;
}

template <int DIM>
static void launch_gemm(hipStream_t st, const DeviceIndex* ix, const float* Qt, int B, int LQP, float* QCT,
                        uint32_t* gmax, uint8_t* QCU = nullptr, int RB = 0, const float* qinv = nullptr,
                        const int32_t* qoff = nullptr) {
  // one 32-centroid fragment per wave: 128 centroids per block, ~2 blocks per CU co-resident, so one wave's
  // epilogue (stores, key maxima) hides under another wave's MFMAs.  KP is a multiple of 64.
  if (ix->tune.gemm_cpw == 2) {
    const unsigned blocks = (unsigned)((ix->KP / 64 + 3) / 4);
    qc_gemm_kernel<DIM, 2><<<blocks, 256, 0, st>>>(ix->d_centroids, ix->K, ix->KP, Qt, B, LQP, QCT, gmax, QCU, RB, qinv, qoff);
  } else {
    const unsigned blocks = (unsigned)((ix->KP / 32 + 3) / 4);
    qc_gemm_kernel<DIM, 1><<<blocks, 256, 0, st>>>(ix->d_centroids, ix->K, ix->KP, Qt, B, LQP, QCT, gmax, QCU, RB, qinv, qoff);
  }
}

template <int DIM, int NBITS, int NQT>
static int launch_exact(hipStream_t st, const DeviceIndex* ix, const ExactP& p, int B, int precision) {
  const unsigned gx = (unsigned)((p.n_sel + 4 * NP_EXACT_DPW - 1) / (4 * NP_EXACT_DPW));
  if (gx == 0 || B == 0) return NP_OK;
  if (precision == 0) {
    const size_t lds = ((size_t)DIM * p.LQP + (1 << NBITS)) * sizeof(float);
    if (lds > 64 * 1024)
      NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&exact_f32_kernel<DIM, NBITS, NQT>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    exact_f32_kernel<DIM, NBITS, NQT><<<dim3(gx, B), 256, lds, st>>>(p);
  } else if (precision == 1 || precision == 2) {
    // Transposed form (4 float4 QC loads per tile, no per-row shuffles), ONE LAUNCH PER 32-TOKEN QUERY TILE: the one-tile
    // instantiation keeps three waves per SIMD; a two-tile one needs 242 VGPRs and ran 48-token queries 4x slower than
    // 32-token ones (2.46 vs 0.60 ms at 10 M documents).  s6_tiles = 0 restores the multi-tile kernels.
    if (!ix->tune.exact_rowmax && (NQT == 1 || ix->tune.s6_tiles)) {
      ExactP px = p;
      dim3 grid(gx, B);
      if (B >= 8 && ix->tune.s6_xcd) {   // one XCD per query (see exact_qct_kernel)
        px.xcd_B = B;
        px.gx = (int)gx;
        grid = dim3(8u * (unsigned)((B + 7) / 8) * gx, 1);
      }
      for (int qt = 0; qt < p.LQP / 32; ++qt) {
        px.qt0 = qt;
        px.acc = qt > 0;
        if (ix->tune.s6_lds == 2) {   // query fragments in LDS, C-in rows one tile ahead; registers cut for 4 waves per SIMD
          if (precision == 1) exact_qcl_kernel<DIM, NBITS, 1, 4><<<grid, 256, 0, st>>>(px);
          else exact_qcl_kernel<DIM, NBITS, 3, 4><<<grid, 256, 0, st>>>(px);
        } else if (ix->tune.s6_lds == 1) {   // the same at 3 waves per SIMD (no spills)
          if (precision == 1) exact_qcl_kernel<DIM, NBITS, 1, 3><<<grid, 256, 0, st>>>(px);
          else exact_qcl_kernel<DIM, NBITS, 3, 3><<<grid, 256, 0, st>>>(px);
        } else {
          if (precision == 1) exact_qct_kernel<DIM, NBITS, 1, 1><<<grid, 256, 0, st>>>(px);
          else exact_qct_kernel<DIM, NBITS, 1, 3><<<grid, 256, 0, st>>>(px);
        }
      }
    } else if (NQT <= 2 && !ix->tune.exact_rowmax) {
      ExactP px = p;
      dim3 grid(gx, B);
      if (B >= 8 && ix->tune.s6_xcd) {
        px.xcd_B = B;
        px.gx = (int)gx;
        grid = dim3(8u * (unsigned)((B + 7) / 8) * gx, 1);
      }
      constexpr int NQ = NQT <= 2 ? NQT : 1;
      if (precision == 1) exact_qct_kernel<DIM, NBITS, NQ, 1><<<grid, 256, 0, st>>>(px);
      else exact_qct_kernel<DIM, NBITS, NQ, 3><<<grid, 256, 0, st>>>(px);
    } else {
      if (precision == 1) exact_qc_kernel<DIM, NBITS, NQT, 1><<<dim3(gx, B), 256, 0, st>>>(p);
      else exact_qc_kernel<DIM, NBITS, NQT, 3><<<dim3(gx, B), 256, 0, st>>>(p);
    }
  } else {
    exact_bf16_kernel<DIM, NBITS, NQT><<<dim3(gx, B), 256, 0, st>>>(p);
  }
  return NP_OK;
}

template <int DIM, int NBITS>
static int launch_exact_qt(hipStream_t st, const DeviceIndex* ix, const ExactP& p, int B, int precision) {
  if (p.LQP <= 32) return launch_exact<DIM, NBITS, 1>(st, ix, p, B, precision);
  if (p.LQP <= 64) return launch_exact<DIM, NBITS, 2>(st, ix, p, B, precision);
  return launch_exact<DIM, NBITS, NP_MAX_QT>(st, ix, p, B, precision);
}

template <int DIM>
static int launch_exact_nb(hipStream_t st, const DeviceIndex* ix, const ExactP& p, int B, int precision, int nbits) {
  if (nbits == 2) return launch_exact_qt<DIM, 2>(st, ix, p, B, precision);
  if (nbits == 8) {   // one dim per byte, 256 bucket weights: the all-f32 kernel at every precision (>= what was asked for)
    const unsigned gx = (unsigned)((p.n_sel + 4 * NP_EXACT_DPW - 1) / (4 * NP_EXACT_DPW));
    if (gx == 0 || B == 0) return NP_OK;
    const size_t lds = ((size_t)DIM * p.LQP + 256) * sizeof(float);
#define NP_F32_NB8(NQT)                                                                                              \
  do {                                                                                                               \
    if (lds > 64 * 1024)                                                                                             \
      NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&exact_f32_kernel<DIM, 8, NQT>),                      \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
    exact_f32_kernel<DIM, 8, NQT><<<dim3(gx, B), 256, lds, st>>>(p);                                                 \
  } while (0)
    if (p.LQP <= 32) NP_F32_NB8(1);
    else if (p.LQP <= 64) NP_F32_NB8(2);
    else NP_F32_NB8(NP_MAX_QT);
#undef NP_F32_NB8
    return NP_OK;
  }
  return launch_exact_qt<DIM, 4>(st, ix, p, B, precision);
}

// S4 for the queries of one round: exact f32 approximate scores of `n[b]` records at meta[cand_base[b]...]
static void launch_approx(hipStream_t st, const DeviceIndex* ix, Workspace& w, const int32_t* d_qoff, int B, int LQP,
                          const uint4* meta, const int32_t* n, const RoundPlan& rp, int round, int max_rounds,
                          Counters* ctr) {
  const Tuning& t = ix->tune;
  const int64_t KP = ix->KP;
  // s4_mode: 0 = all XCDs walk one query (approx_kernel), 1..4 = one XCD per query in 8/4/2/1 phases,
  // 5..8 = the same with every group streaming through its documents (approx_stream_kernel)
  const unsigned nbx = (unsigned)t.s4_nbx;   // workgroups per XCD
  const uint32_t slice_w = (uint32_t)((ix->K + 7) / 8);
  const int s4_p = std::min(t.s4_mode > 4 ? t.s4_mode - 4 : t.s4_mode, 4);   // phases = 8 >> (s4_p - 1)
  const bool stream = t.s4_mode >= 5 && t.s4_mode <= 8 && ((uint64_t)slice_w << (t.s4_mode - 5)) <= 65536ull;
  if (stream && ix->sliced_ok && B >= t.s4_minb) {
    // streamed form (approx_stream_kernel): u16 code-in-slice needs a phase's centroid range <= 65536
#define NP_LAUNCH_APPROX_S(LPR)                                                                                       \
  approx_stream_kernel<LPR><<<8 * nbx, 256, 0, st>>>(w.QCT.as<float>(), KP, LQP, d_qoff, meta, n, rp, round,          \
                                                     max_rounds, ix->ucodes(), ix->n_ucodes, ix->d_useg, w.approx.as<float>(), \
                                                     t.s4_mode - 5, slice_w, ctr)
    if (LQP <= 32) NP_LAUNCH_APPROX_S(8);
    else if (LQP <= 64) NP_LAUNCH_APPROX_S(16);
    else if (LQP <= 128) NP_LAUNCH_APPROX_S(32);
    else NP_LAUNCH_APPROX_S(64);
#undef NP_LAUNCH_APPROX_S
  } else if (t.s4_mode > 0 && ix->sliced_ok && B >= t.s4_minb) {
#define NP_LAUNCH_APPROX_X(LPR, SWZ)                                                                                  \
  approx_xcd_kernel<LPR, SWZ><<<8 * nbx, 256, 0, st>>>(w.QCT.as<float>(), KP, LQP, d_qoff, meta, n, rp, round,        \
                                                       max_rounds, ix->ucodes(), ix->n_ucodes, ix->d_useg,         \
                                                       w.approx.as<float>(), s4_p - 1, ctr)
    if (LQP <= 32) {
      if (t.s4_swz) NP_LAUNCH_APPROX_X(8, true);
      else NP_LAUNCH_APPROX_X(8, false);
    } else if (LQP <= 64) NP_LAUNCH_APPROX_X(16, false);
    else if (LQP <= 128) NP_LAUNCH_APPROX_X(32, false);
    else NP_LAUNCH_APPROX_X(64, false);
#undef NP_LAUNCH_APPROX_X
  } else {
    const unsigned grid = 768;
#define NP_LAUNCH_APPROX(LPR)                                                                                          \
  approx_kernel<LPR><<<grid, 256, 0, st>>>(w.QCT.as<float>(), KP, LQP, d_qoff, meta, n, rp, round, max_rounds,         \
                                           ix->ucodes(), w.approx.as<float>(), ctr)
    if (LQP <= 32) NP_LAUNCH_APPROX(8);
    else if (LQP <= 64) NP_LAUNCH_APPROX(16);
    else if (LQP <= 128) NP_LAUNCH_APPROX(32);
    else NP_LAUNCH_APPROX(64);
#undef NP_LAUNCH_APPROX
  }
}

// Batched path (search.rs:259-272): approximate scores of the listed documents in the reference's mat-vec arithmetic
static void launch_matvec(hipStream_t st, const DeviceIndex* ix, Workspace& w, const float* d_q, const int32_t* d_qoff, int B,
                          const uint4* meta, const int32_t* n, const RoundPlan& rp, int round) {
  const dim3 grid(512 / NP_MV_DOCS, (unsigned)B);      // a wave per NP_MV_DOCS documents: one pass over ~n_sel listed documents
#define NP_MATVEC(D)                                                                                                   \
  do {                                                                                                                 \
    if (ix->ldim & 7)                                                                                                  \
      approx_matvec_kernel<D, true><<<grid, 256, 0, st>>>(d_q, d_qoff, ix->d_centroids, meta, n, rp, round, ix->ucodes(), \
                                                          w.approx.as<float>(), ix->ldim);                             \
    else                                                                                                               \
      approx_matvec_kernel<D, false><<<grid, 256, 0, st>>>(d_q, d_qoff, ix->d_centroids, meta, n, rp, round, ix->ucodes(), \
                                                           w.approx.as<float>(), ix->ldim);                            \
  } while (0)
  switch (ix->dim) {
    case 32: NP_MATVEC(32); break;
    case 64: NP_MATVEC(64); break;
    case 96: NP_MATVEC(96); break;
    default: NP_MATVEC(128); break;
  }
#undef NP_MATVEC
}

// S1..S5 for queries [0,B) whose rows live in d_q (absolute offsets d_qoff/h_qoff).
static int phase_a_once(const DeviceIndex* ix, CallState* cs, const float* d_q, const int32_t* d_qoff,
                        const int32_t* h_qoff, const int64_t* d_subset, int64_t subset_len, bool allow_grow);

// A reservation that fails under the DEFAULT budget (another index or an encoder took the memory since open) is retried
// with the pool released and the budget halved -- more candidate-pool rounds instead of OutOfMemory.  Every reservation of
// a pass happens before its first launch touches the buffer concerned, so a failed pass leaves nothing half-done.
static int phase_a(const DeviceIndex* ix, CallState* cs, const float* d_q, const int32_t* d_qoff,
                   const int32_t* h_qoff, const int64_t* d_subset, int64_t subset_len) {
  for (int attempt = 0;; ++attempt) {
    // (a retry never lets the budget grow back: the pass that just failed WAS the planned size)
    const int rc = phase_a_once(ix, cs, d_q, d_qoff, h_qoff, d_subset, subset_len, attempt == 0);
    if (rc != NP_ERR_OUT_OF_MEMORY || !ix->ws_auto || attempt >= 4) return rc;
    const int64_t b = ix->ws_budget.load(std::memory_order_relaxed);
    if (b <= ((int64_t)256 << 20)) return rc;
    (void)hipGetLastError();
    (void)hipStreamSynchronize(cs->stream);   // the pool may still be read by work queued before the failure
    cs->ctx->ws->release_pool();
    ix->ws_budget.store(std::max<int64_t>(b / 2, (int64_t)256 << 20), std::memory_order_relaxed);
  }
}

static int phase_a_once(const DeviceIndex* ix, CallState* cs, const float* d_q, const int32_t* d_qoff,
                        const int32_t* h_qoff, const int64_t* d_subset, int64_t subset_len, bool allow_grow) {
  Workspace& w = *cs->ctx->ws;
  hipStream_t st = cs->stream;
  const int B = cs->B;
  const np_search_params& prm = cs->prm;
  int maxLq = 1;
  for (int b = 0; b < B; ++b) maxLq = std::max(maxLq, h_qoff[b + 1] - h_qoff[b]);
  for (int b = 0; b < B; ++b)
    if (h_qoff[b + 1] < h_qoff[b]) {
      set_error("Shape error: q_tok_offsets must be non-decreasing");
      return NP_ERR_SHAPE;
    }
  const int LQP = (maxLq + 31) / 32 * 32;
  if (LQP > 32 * NP_MAX_QT) {
    set_error("Shape error: queries longer than %d tokens are not supported by the HIP path (got %d)", 32 * NP_MAX_QT,
              maxLq);
    return NP_ERR_SHAPE;
  }
  if ((uint64_t)ix->KP * (uint64_t)LQP * 4ull >= (1ull << 32)) {   // S4 / S6 address one query's table with 32-bit byte offsets
    set_error("Shape error: %lld centroids x %d query tokens exceed the 4 GiB per-query score table of the HIP path",
              (long long)ix->K, LQP);
    return NP_ERR_SHAPE;
  }
  if (ix->ldim != ix->dim) {   // caller rows -> storage rows; everything below sees ix->dim
    const int64_t r0 = h_qoff[0], nr = (int64_t)h_qoff[B] - r0;
    NP_TRY(w.qpad.reserve((size_t)std::max<int64_t>(nr, 1) * ix->dim * 4));
    if (nr > 0)
      pad_rows_kernel<<<(unsigned)((nr * ix->dim + 255) / 256), 256, 0, st>>>(d_q + r0 * ix->ldim, nr, ix->ldim, ix->dim,
                                                                              w.qpad.as<float>());
    d_q = w.qpad.as<float>() - r0 * ix->dim;   // offsets stay absolute
  }
  cs->LQP = LQP;
  cs->n_sel = n_sel_of(&prm);
  cs->NSELP = next_pow2(std::max(cs->n_sel, 1));
  cs->empty_subset = (subset_len == 0);
  const int64_t KP = ix->KP, G = KP / 32, NW = (ix->n_docs + 31) / 32;
  const int nchunks = (int)((NW + NP_CHUNK_WORDS - 1) / NP_CHUNK_WORDS);
  const int nsel1 = std::max(cs->n_sel, 1), topk1 = std::max(prm.top_k, 1);
  const int64_t probed_cells = subset_len < 0 ? (int64_t)std::max(prm.n_ivf_probe, 1) * maxLq : 0;
  WsPlan plan = plan_workspace(ix, B, LQP, &prm, probed_cells);
  if (ix->ws_auto) {
    // The default budget was what the device had free at open.  Before a pool GROWS, and whenever the budget stands below
    // its value at open, look at what is free now: the budget covers this context's scratch AND pool, so what this context
    // could hold in total is the free memory plus everything it already holds, minus a GiB for the other contexts' small
    // buffers and the allocator's granularity.  The budget shrinks when the batch would not fit (another tenant took the
    // memory since open: more rounds, not OutOfMemory) and returns to the open value only when a whole budget is FREE on the
    // device again, whatever this context holds (never on the retry of a pass that just failed to reserve its plan).  A looser
    // rule -- "a quarter more than the current budget is reachable" -- made the three contexts of a 12.5 M-document shard,
    // which share ~36 GiB with nothing to spare, take turns shrinking and regrowing the shared budget and reallocating their
    // pools: 466 instead of ~15 000 queries/s.
    const int64_t want = std::min<int64_t>(plan.pool, (int64_t)std::max(B, 1) * std::max<int64_t>(ix->n_docs, 1));
    const int64_t budget = ix->ws_budget.load(std::memory_order_relaxed);
    // (a context whose own pool fills the device never sees a whole budget free: it would pay hipMemGetInfo on every call for
    // nothing, so the regrow probe runs on every 32nd call of the context; a pool that must GROW always looks)
    const bool below = allow_grow && budget < ix->ws_budget_open && (w.probe_tick++ & 31u) == 0u;
    if ((int64_t)w.cand.cap < want * 4 || below) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const int64_t held = (int64_t)w.total_bytes();
        const int64_t avail = (int64_t)free_b + held - ((int64_t)1 << 30);
        const int64_t need = plan.S * per_query_bytes(ix, LQP, n_sel_of(&prm), prm.top_k) + want * NP_POOL_ENTRY;
        int64_t nb = budget;
        if (avail < budget && need > avail) nb = std::max<int64_t>(avail, (int64_t)256 << 20);
        else if (below && (int64_t)free_b >= ix->ws_budget_open + ((int64_t)1 << 30)) nb = ix->ws_budget_open;
        if (nb != budget) {
          // concurrent contexts share the budget: only the context whose view is still current installs its value (a lost
          // race re-plans from whatever the winner stored)
          int64_t seen = budget;
          (void)ix->ws_budget.compare_exchange_strong(seen, nb, std::memory_order_relaxed);
          plan = plan_workspace(ix, B, LQP, &prm, probed_cells);
        }
      }
    }
  }
  const int64_t pool = std::min<int64_t>(plan.pool, (int64_t)std::max(B, 1) * std::max<int64_t>(ix->n_docs, 1));
  const int max_rounds = std::max(1, std::min(plan.max_rounds, std::max(B, 1)));

  NP_TRY(w.Qt.reserve((size_t)B * ix->dim * LQP * 4));
  NP_TRY(w.Qb.reserve((size_t)B * ix->dim * LQP * 2));
  NP_TRY(w.Qbl.reserve((size_t)B * ix->dim * LQP * 2));
  NP_TRY(w.QCT.reserve((size_t)B * KP * LQP * 4));
  NP_TRY(w.gmax.reserve((size_t)B * G * LQP * 4));
  NP_TRY(w.cellbits.reserve((size_t)B * G * 4));
  NP_TRY(w.tauq.reserve((size_t)B * LQP * 4));
  NP_TRY(w.cells_tmp.reserve((size_t)B * KP * 4));
  NP_TRY(w.cells.reserve((size_t)B * KP * 4));
  NP_TRY(w.n_cells.reserve((size_t)B * 4));
  NP_TRY(w.docbits.reserve((size_t)B * std::max<int64_t>(NW, 1) * 4));
  NP_TRY(w.chunk_counts.reserve((size_t)B * std::max(nchunks, 1) * 4));
  NP_TRY(w.cand.reserve((size_t)pool * 4));
  NP_TRY(w.cand_meta.reserve((size_t)pool * 16));
  NP_TRY(w.approx.reserve((size_t)pool * 4));
  NP_TRY(w.n_cand.reserve((size_t)B * 4));
  NP_TRY(w.cand_base.reserve((size_t)B * 8));
  NP_TRY(w.round_of.reserve((size_t)B * 4));
  NP_TRY(w.round_tab.reserve((size_t)(2 * max_rounds + 1) * 4));
  NP_TRY(w.q_order.reserve((size_t)B * 4));
  NP_TRY(w.n_list2.reserve((size_t)B * 4));
  // S4 upper-bound filter (np_kernels.h): off for debug traces (every candidate keeps its exact score) and for
  // indices with a non-finite centroid value
  const bool use_filter = ix->tune.s4_filter && ix->filter_ok && !cs->trace && cs->n_sel > 0 && ix->T > 0;
  const int RB = LQP <= 32 ? 32 : (LQP <= 64 ? 64 : (LQP <= 128 ? 128 : 256));   // u8 table row bytes
  NP_TRY(w.qinv.reserve((size_t)B * 4));
  NP_TRY(w.qflag.reserve((size_t)B * 4));
  // two-level filter (np_kernels.h, "S4, first filter level"): the hot bitmap of a query lives in LDS (K / 8 bytes)
  // bit-plane form of the first level (approx_hotp_kernel): rows of 32 / 64 query tokens; list blocks of up to 512 bytes
  // (staged by a whole wave, LPD = 4) exist only with it -- approx_hot_kernel stages at most 256-byte blocks
  const int old_cap = ix->code_wide ? 64 : 128;
  const bool use_planes = ix->tune.s4_planes && RB <= 64;
  const bool two_level = use_filter && ix->tune.s4_hot > 0 && KP / 8 <= 64 * 1024 && KP * RB < ((int64_t)1 << 31) &&
                         ix->ublock_stride > 0 && (use_planes || ix->ublock_stride <= old_cap);
  // Share of the centroids whose rows the exact level still gathers (the rest: floored at Lambda2).  A token more is a floor
  // more and a longer list has a higher maximum per token, so the best share RISES with the query length and FALLS with the
  // documents' distinct-code count (tools/sim/s4_warm_sim.py: 50 % at 32 tokens / 68 codes, ~70 % at 48 tokens, ~30 % at 240
  // codes per document; measured at 48 tokens: 10.10 k -> 10.47 k queries/s with 70 %).  s4_warm > 0 pins one value.
  const int s4_warm = ix->tune.s4_warm > 0
                          ? ix->tune.s4_warm
                          : (int)std::min(1000.f, std::max(300.f, 500.f + 12.5f * (float)std::max(0, maxLq - 32) -
                                                                      1.16f * std::max(0.f, ix->ulen_mean - 68.f)));
  const size_t slot_words = (size_t)(8 * (B + 1) + 1);   // hand-out slots + ticket of one filter launch
  if (use_filter) {
    NP_TRY(w.QCU.reserve((size_t)B * KP * RB));
    NP_TRY(w.ub.reserve((size_t)pool * 2));
    NP_TRY(w.ub_hist.reserve((size_t)B * NP_UB_BINS * 4));
    NP_TRY(w.surv_meta.reserve((size_t)pool * 16));
    NP_TRY(w.n_surv.reserve((size_t)B * 4));
    NP_TRY(w.ub_thr.reserve((size_t)B * 4));
    NP_TRY(w.ub_cursor.reserve((size_t)3 * B * 4));
    NP_TRY(w.xcd_slots.reserve(((size_t)max_rounds * 3 + 1) * slot_words * 4));   // + the zeroth level's S0 launch
  }
  if (two_level) {
    NP_TRY(w.cmaxu.reserve((size_t)B * KP));
    NP_TRY(w.chist.reserve((size_t)B * 256 * 4));
    NP_TRY(w.ub2.reserve((size_t)pool * 2));
    NP_TRY(w.ub_hist2.reserve((size_t)B * NP_UB_BINS * 4));
    NP_TRY(w.ub_thr2.reserve((size_t)3 * B * 4));   // [B] tau bins, [B] Lambda, [B] Lambda2 (floor of the exact level)
    NP_TRY(w.list_meta.reserve((size_t)pool * 16));
    NP_TRY(w.n_l1.reserve((size_t)B * 4));
    NP_TRY(w.n_l2.reserve((size_t)B * 4));
    if (use_planes) {
      NP_TRY(w.planes.reserve((size_t)B * KP * RB));
      NP_TRY(w.levels.reserve((size_t)B * 16 * 4));
      NP_TRY(w.hotbits.reserve((size_t)2 * B * (KP / 32) * 4));   // hot bitmap, then the exact level's kept-centroid bitmap
    }
  }
  // Zeroth filter level (np_kernels.h, gain_sweep_kernel): per-document sums of the probed cells' gains prune the candidates
  // before any list block is read.  Only where no centroid_score_threshold is set (the cells a threshold removes would lift
  // the bound's floor above the cut: tools/sim/s3_gain_sim.py), on ascending posting lists (range table built at open), with
  // the bit-plane first level behind it (it takes the candidate ids in any order) and without a subset.
  bool gain_path = two_level && use_planes && ix->d_ivf_split != nullptr && ix->tune.s3_gain &&
                   subset_len < 0 && ix->n_docs > 0 && cs->n_sel > 0 && B > 0 &&
                   (int64_t)std::max(prm.n_ivf_probe, 32) * maxLq <= 16384;   // probed cells per query: the scaled gains of all of
                                                                               // them must fit a 15-bit accumulator (gain_prep_kernel)
  const bool gain_possible = gain_path;   // the level's buffers are reserved whenever it MAY run: a first run in the middle of a
                                          // service's life must not stall every stream on a dozen hipMalloc calls
  if (gain_path && ix->tune.s3_gain == 1) {
    // Run / skip policy.  The level costs about the same whatever it prunes -- one sweep of the probed lists (to depth 32; with a
    // threshold also the cells it removes), a level byte per document and query written and read twice -- and what it buys is the
    // filter's time per candidate it removes (~0.1 ns of GPU time per 192-byte list block).  The device leaves (candidates, kept,
    // posting entries swept) of each batch in pinned words; a context reads the words of ITS previous batch here -- never waited
    // for: a batch still in flight simply has not reported -- and when the removed candidates would not have paid for the level,
    // the handle skips it for 31 batches (255 when it was not even close) and then tries again.  With a threshold the level starts
    // skipped (the metric corpus: it does not pay) and is tried for the first time after 511 batches -- a trial costs a short-lived
    // process more than the level's ~1.3 ms (the first launch of its kernels loads their code: ~40 ms measured inside a 300-batch
    // bench), a service never notices.  Results do not depend on the decision.
    const uint64_t key = ((uint64_t)(uint32_t)prm.n_ivf_probe << 40) ^ ((uint64_t)(uint32_t)cs->n_sel << 16) ^ (uint64_t)(uint32_t)LQP ^
                         ((uint64_t)(prm.has_threshold ? 1u : 0u) << 63);
    if (ix->gain_key.exchange(key, std::memory_order_relaxed) != key) {
      ix->gain_run.store(0, std::memory_order_relaxed);
      ix->gain_skip.store(prm.has_threshold ? 511 : 0, std::memory_order_relaxed);
    }
    if (w.h_gain) {
      const unsigned long long v = __atomic_exchange_n(&w.h_gain[0], 0ull, __ATOMIC_ACQUIRE);
      const double raw = (double)(v >> 32), kept = (double)(v & 0xFFFFFFFFull), swept = (double)w.h_gain[1];
      const double block_b = (double)ix->ublock_stride * (double)ix->code_bytes();
      const double benefit_ms = (raw - kept) * 1e-7 * std::max(1.0, block_b / 192.0);   // 8 ms per 130 M candidates at K = 2^16
      const double cost_ms = 0.8 * ((double)ix->n_docs / 1e7) * ((double)B / 64.0) + swept * 3e-9;   // passes + ~330 M entries per ms
      // (a report of a batch with other parameters says nothing about these)
      if (raw > 0 && w.h_gain_key == key) {
        if (benefit_ms < cost_ms) {
          ix->gain_run.store(0, std::memory_order_relaxed);
          ix->gain_skip.store(benefit_ms > 0.7 * cost_ms ? 31 : 255, std::memory_order_relaxed);
        } else {
          ix->gain_run.store(1, std::memory_order_relaxed);
        }
      }
    } else if (hipHostMalloc((void**)&w.h_gain, 64, hipHostMallocDefault) == hipSuccess) {
      w.h_gain[0] = w.h_gain[1] = 0;
    } else {
      w.h_gain = nullptr;
      (void)hipGetLastError();
    }
    if (ix->gain_skip.load(std::memory_order_relaxed) > 0) {
      ix->gain_skip.fetch_sub(1, std::memory_order_relaxed);
      gain_path = false;
    } else {
      w.h_gain_key = key;
      // a trial run: the other contexts wait for its report instead of each paying for one
      if (!ix->gain_run.load(std::memory_order_relaxed)) ix->gain_skip.store(3, std::memory_order_relaxed);
    }
  }
  // the level probes on its own to depth 32 where the search stops earlier: the bound's floor falls with the depth (np_kernels.h)
  // ... and with a threshold it sweeps the cells the threshold removes too (bound-only): its own probe, without the threshold
  const int gain_depth = std::max(32, prm.n_ivf_probe);
  const bool deep_wanted = (prm.n_ivf_probe < gain_depth || prm.has_threshold) && ix->K > gain_depth;
  const bool gain_deep = gain_path && deep_wanted;
  if (gain_path && prm.has_threshold && !gain_deep) gain_path = false;
  const int s0_target = ix->tune.s3_gain_mult * cs->n_sel;
  // S0 takes whole histogram bins: the marginal bin may hold a few whole posting lists (documents in ONE probed cell share a bound)
  // S0 takes the bins above the marginal one whole and fills the rest of its slice from the marginal bin (documents in ONE probed
  // cell share a bound: a bin may hold whole posting lists)
  const int s0cap = s0_target + cs->n_sel;
  // u32 words of w.gsmall: [0, 4B) base / shift / floor bin / 0, then B each: n_raw, thr0, cut0, n_s0, n_emit, n_direct, round_of0, order0, cursor0,
  // n_hi, n_hi_emit, n_marg, lcut; 4 words round_tab0; 6 words = 3 x u64 batch report; then (8-byte aligned) cand_base0 i64 [B]
  const size_t gs_words = (size_t)17 * B + 4 + 8, gs_bytes = (gs_words + (gs_words & 1)) * 4 + (size_t)B * 8;
  if (gain_possible) {
    NP_TRY(w.gain.reserve((size_t)B * KP * 2));
    NP_TRY(w.gsmall.reserve(gs_bytes));
    NP_TRY(w.ghist.reserve((size_t)B * (256 + NP_UB_BINS) * 4));   // levels of all candidates; exact lower bounds of S0
    NP_TRY(w.s0_meta.reserve((size_t)B * s0cap * 16));
    NP_TRY(w.s0_u.reserve((size_t)B * s0cap * 2));
    NP_TRY(w.gacc.reserve((size_t)B * ix->n_ranges * NP_GAIN_RANGE));   // one level byte per document and query
    if (deep_wanted) NP_TRY(w.gdeep.reserve(((size_t)B * G + (size_t)B * LQP + (size_t)B + (size_t)B * G + (size_t)B * KP) * 4));
  }
  NP_TRY(w.sel_keys.reserve((size_t)B * nsel1 * 8));
  NP_TRY(w.sel_doc.reserve((size_t)B * nsel1 * 4));
  NP_TRY(w.nsel.reserve((size_t)B * 4));
  NP_TRY(w.exact.reserve((size_t)B * nsel1 * 4));
  NP_TRY(w.out_ids.reserve((size_t)B * topk1 * 8));
  NP_TRY(w.out_scores.reserve((size_t)B * topk1 * 4));
  NP_TRY(w.out_keys.reserve((size_t)B * topk1 * 8));
  NP_TRY(w.out_counts.reserve((size_t)B * 4));
  NP_TRY(w.ctr.reserve(sizeof(Counters)));
  NP_TRY(w.misc.reserve(64));

  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[0], st));
  {
    // every small region of the call in ONE launch (a dozen stream memsets were ~50 us per batch)
    ClearList cl;
    cl.n = 0;
    auto add = [&](void* p, size_t bytes, uint32_t fill) {
      if (bytes == 0) return;
      cl.p[cl.n] = static_cast<uint32_t*>(p);
      cl.words[cl.n] = (uint32_t)(bytes / 4);
      cl.fill[cl.n] = fill;
      ++cl.n;
    };
    static_assert(sizeof(Counters) % 4 == 0, "Counters is cleared by words");
    add(w.ctr.p, sizeof(Counters), 0);
    add(w.n_cells.p, (size_t)B * 4, 0);
    add(w.n_cand.p, (size_t)B * 4, 0);
    add(w.nsel.p, (size_t)B * 4, 0);
    add(w.cellbits.p, (size_t)B * G * 4, 0);
    add(w.tauq.p, (size_t)B * LQP * 4, 0);
    if (cs->n_sel > 0) add(w.sel_keys.p, (size_t)B * cs->n_sel * 8, 0);
    if (use_filter && B > 0) {
      add(w.ub_hist.p, (size_t)B * NP_UB_BINS * 4, 0);
      add(w.n_surv.p, (size_t)B * 4, 0);
      add(w.ub_cursor.p, (size_t)3 * B * 4, 0);
      add(w.xcd_slots.p, ((size_t)max_rounds * 3 + 1) * slot_words * 4, 0xFFFFFFFFu);   // slots and tickets of every launch: -1
    }
    if (gain_deep) add(w.gdeep.p, ((size_t)B * G + (size_t)B * LQP + (size_t)B + (size_t)B * G) * 4, 0);   // marks, per-token thresholds, cell counts, kept-cell bitmap
    if (gain_path) {
      add(w.gsmall.p, gs_bytes, 0);
      add(w.ghist.p, (size_t)B * (256 + NP_UB_BINS) * 4, 0);
    }
    if (two_level && B > 0) {
      add(w.chist.p, (size_t)B * 256 * 4, 0);
      add(w.ub_hist2.p, (size_t)B * NP_UB_BINS * 4, 0);
      add(w.n_l1.p, (size_t)B * 4, 0);
      add(w.n_l2.p, (size_t)B * 4, 0);
    }
    if (cl.n > 0) clear_regions_kernel<<<128, 256, 0, st>>>(cl);
  }
  if (NW > 0 && !ix->tune.s3_slices) NP_HIP(hipMemsetAsync(w.docbits.p, 0, (size_t)B * NW * 4, st));   // mark_slices_kernel writes every word
  if (use_filter && B > 0 && RB != LQP) NP_HIP(hipMemsetAsync(w.QCU.p, 0, (size_t)B * KP * RB, st));   // row bytes LQP .. RB-1 stay 0
  if (B == 0) return NP_OK;

  // ---- S1
  prep_queries_kernel<<<B, 256, 0, st>>>(d_q, d_qoff, ix->dim, LQP, w.Qt.as<float>(), w.Qb.as<__bf16>(),
                                         w.Qbl.as<__bf16>(), ix->cmax, w.qinv.as<float>(), w.qflag.as<uint32_t>());
  // split-bf16 S1 (qc_gemm_b3_kernel): opt-in, only where the crate itself leaves the dense path (K > centroid_batch_size)
  // and the caller asked for a reduced-precision mode; precision 0 keeps the exact-f32 chain everywhere
  // (with it the approximate scores stay the GEMM's: the batched path's mat-vec re-scoring -- there to reproduce the
  // reference's non-FMA summation order bit for bit, 1.6 ms of packed-f32 VALU work per batch at K = 2^19 -- has nothing
  // left to reproduce)
  const bool s1_split = ix->tune.s1_split && prm.precision >= 1 && prm.centroid_batch_size > 0 &&
                        ix->K > prm.centroid_batch_size && !cs->trace;
  if (s1_split) {
    uint8_t* qcu = use_filter ? w.QCU.as<uint8_t>() : nullptr;
    const unsigned blocks = (unsigned)((ix->KP / 32 + 3) / 4);
#define NP_GEMM_B3(D)                                                                                                     \
  qc_gemm_b3_kernel<D><<<blocks, 256, 0, st>>>(ix->d_centroids, ix->K, ix->KP, w.Qb.as<__bf16>(), w.Qbl.as<__bf16>(), B, LQP, \
                                               w.QCT.as<float>(), w.gmax.as<uint32_t>(), qcu, RB, w.qinv.as<float>(), d_qoff)
    switch (ix->dim) {
      case 32: NP_GEMM_B3(32); break;
      case 64: NP_GEMM_B3(64); break;
      case 96: NP_GEMM_B3(96); break;
      default: NP_GEMM_B3(128); break;
    }
#undef NP_GEMM_B3
  } else {
    uint8_t* qcu = use_filter ? w.QCU.as<uint8_t>() : nullptr;
    const float* qinv = w.qinv.as<float>();
    switch (ix->dim) {
      case 32: launch_gemm<32>(st, ix, w.Qt.as<float>(), B, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>(), qcu, RB, qinv, d_qoff); break;
      case 64: launch_gemm<64>(st, ix, w.Qt.as<float>(), B, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>(), qcu, RB, qinv, d_qoff); break;
      case 96: launch_gemm<96>(st, ix, w.Qt.as<float>(), B, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>(), qcu, RB, qinv, d_qoff); break;
      default: launch_gemm<128>(st, ix, w.Qt.as<float>(), B, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>(), qcu, RB, qinv, d_qoff); break;
    }
  }
  if (two_level) {   // per-centroid maxima of the u8 table, their histogram, the hot level's Lambda
    hot_prep_kernel<<<dim3((unsigned)std::min<int64_t>((KP + 255) / 256, 64), B), 256, 0, st>>>(
        w.QCU.as<uint8_t>(), ix->K, KP, RB, w.cmaxu.as<uint8_t>(), w.chist.as<uint32_t>());
    if (!use_planes) hot_lam_kernel<<<B, 256, 0, st>>>(w.chist.as<uint32_t>(), ix->K, ix->tune.s4_hot, w.ub_thr2.as<uint32_t>() + B);
  }
  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[1], st));

  // ---- subset pre-filter (search.rs:350-382); the batched path only filters candidates (:542-545)
  const bool have_subset = subset_len > 0;
  const bool batched = prm.centroid_batch_size > 0 && ix->K > prm.centroid_batch_size;  // search.rs:337
  const bool use_elig = have_subset && !batched;
  const uint32_t* elig_bits = nullptr;
  if (have_subset) {
    NP_TRY(w.subset_bits.reserve((size_t)std::max<int64_t>(NW, 1) * 4));
    NP_TRY(w.elig.reserve((size_t)G * 4));
    NP_HIP(hipMemsetAsync(w.subset_bits.p, 0, (size_t)std::max<int64_t>(NW, 1) * 4, st));
    NP_HIP(hipMemsetAsync(w.elig.p, 0, (size_t)G * 4, st));
    // a document shard sees only its own documents' codes: the sharded host ORs the shards' bitmaps
    // (np_hip_subset_eligible + one small all-gather) and hands the global one in
    const bool local_elig = use_elig && !cs->elig_global;
    subset_kernel<<<(unsigned)((subset_len + 3) / 4), 256, 0, st>>>(
        d_subset, subset_len, ix->doc_begin, ix->n_docs, ix->d_doc_offsets, ix->codes(), w.subset_bits.as<uint32_t>(),
        local_elig ? w.elig.as<uint32_t>() : nullptr);
    if (use_elig) {
      elig_bits = cs->elig_global ? cs->elig_global : w.elig.as<uint32_t>();
      subset_nprobe_kernel<<<1, 256, 0, st>>>(elig_bits, G, prm.n_ivf_probe, ix->N_total, subset_len,
                                              w.misc.as<int32_t>(), w.misc.as<int32_t>() + 1);
      // the probe prunes by group maxima: restrict them to the eligible centroids
      masked_gmax_kernel<<<dim3((unsigned)((G + 3) / 4), B), 256, 0, st>>>(w.QCT.as<float>(), KP, ix->K, LQP, elig_bits,
                                                                           w.gmax.as<uint32_t>());
    }
  }

  // ---- S2
  if (!cs->empty_subset) {
    ProbeP pp;
    pp.QCT = w.QCT.as<float>();
    pp.gmax = w.gmax.as<uint32_t>();
    pp.qoff = d_qoff;
    pp.K = ix->K;
    pp.KP = KP;
    pp.LQP = LQP;
    pp.nprobe = prm.n_ivf_probe;
    pp.nprobe_dev = use_elig ? w.misc.as<int32_t>() + 1 : nullptr;
    pp.elig = use_elig ? elig_bits : nullptr;
    pp.n_elig = use_elig ? w.misc.as<int32_t>() : nullptr;
    pp.has_thr = prm.has_threshold;
    pp.thr = prm.centroid_score_threshold;
    pp.slab = batched ? (int64_t)prm.centroid_batch_size : 0;
    pp.cellbits = w.cellbits.as<uint32_t>();
    pp.tauq = w.tauq.as<uint32_t>();
    pp.cells_tmp = w.cells_tmp.as<uint32_t>();
    pp.cells = w.cells.as<uint32_t>();
    pp.n_cells = w.n_cells.as<int32_t>();
    pp.ctr = w.ctr.as<Counters>();
    // K <= 65536: the block's group maxima (KP/32 x 4 tokens x 4 B <= 32 KB) are staged in LDS once
    const size_t gm_lds = (size_t)(KP / 32) * 4 * 4;
    pp.lds_gm = gm_lds <= 32 * 1024 ? 1 : 0;
    if (pp.lds_gm) probe_mark_kernel<4><<<dim3((unsigned)(LQP / 4), B), 256, gm_lds, st>>>(pp);
    else probe_mark_kernel<8><<<dim3((unsigned)(LQP / 8), B), 256, 0, st>>>(pp);
    probe_finish_kernel<<<dim3(NP_PROBE_NF, B), 256, 0, st>>>(pp);
    if (gain_deep) {   // the zeroth level's own, deeper probe: bound-only cells beyond the search's
      ProbeP p2 = pp;
      uint32_t* gd = w.gdeep.as<uint32_t>();
      p2.nprobe = gain_depth;
      p2.has_thr = 0;          // every probed cell: the ones a threshold removes are swept as bound-only cells
      p2.cellbits = gd;
      p2.tauq = gd + (size_t)B * G;
      p2.n_cells = reinterpret_cast<int32_t*>(gd + (size_t)B * G + (size_t)B * LQP);
      p2.cells = gd + (size_t)2 * B * G + (size_t)B * LQP + (size_t)B;   // (the kept-cell bitmap sits in between)
      p2.ctr = nullptr;
      if (p2.lds_gm) probe_mark_kernel<4><<<dim3((unsigned)(LQP / 4), B), 256, gm_lds, st>>>(p2);
      else probe_mark_kernel<8><<<dim3((unsigned)(LQP / 8), B), 256, 0, st>>>(p2);
      probe_finish_kernel<<<dim3(NP_PROBE_NF, B), 256, 0, st>>>(p2);
      if (prm.has_threshold)   // which of them make candidates: the cells the threshold kept
        cells_to_bits_kernel<<<B, 256, 0, st>>>(w.cells.as<uint32_t>(), w.n_cells.as<int32_t>(), KP,
                                                gd + (size_t)B * G + (size_t)B * LQP + (size_t)B);
    }
  }
  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[2], st));

  // ---- the filter's launch helper and its loop-invariant parameters (used by the zeroth level before the round plan, and by
  // every round)
  const int hshift = RB == 32 ? 2 : (RB == 64 ? 3 : (RB == 128 ? 4 : 5));   // U <= 255 * RB fits NP_UB_BINS << hshift
  const unsigned nbx = (unsigned)ix->tune.ub_nbx;
  const bool oob = ix->tune.ub_nt == 2 && KP * RB < ((int64_t)1 << 30);   // the table behind a 32-bit buffer offset
  // exact u8 bound of the records meta[begin[b] .. begin[b] + count[b]) -> U, histogram (optional)
  // direct_wpq > 0: a short list per query (S1, about n_sel documents): wpq workgroups per query, every query at once,
  // instead of one query per XCD at a time (8 hand-out steps of ~25 us each for a handful of claims)
  // floor: the S2 list of the two-level filter with u16 codes -- rows of the centroids no query token is close to are
  // skipped, U = the floored upper bound, the histogram counts the lower bound (approx_ub_kernel, FLOOR)
  const bool can_floor = two_level && use_planes && oob && RB <= 64 && s4_warm < 1000;
  auto launch_ub_at = [&](const RoundPlan& rpx, int r, int max_rounds, int32_t* sl, int32_t* tk, uint32_t* cursor, const uint4* meta,
                          const int32_t* begin, const int32_t* count, const int32_t* n_all, uint16_t* U, uint32_t* hist,
                          int count_tokens, int direct_wpq, bool floor_rows) {
    const unsigned grid = direct_wpq > 0 ? (unsigned)(B * direct_wpq) : 8 * nbx;
    if (floor_rows && can_floor) {
#define NP_LAUNCH_UBF(ROWB, CT)                                                                                               \
  approx_ub_kernel<ROWB, CT, 2, 1><<<grid, 256, 0, st>>>(                                                                       \
  w.QCU.as<uint8_t>(), KP, meta, begin, count, n_all, rpx, r, max_rounds, (const CT*)ix->d_ucodes,           \
  w.qflag.as<uint32_t>(), cs->n_sel, U, hist, hshift, cursor, sl, tk, B, ix->tune.ub_steal, w.ctr.as<Counters>(),      \
  count_tokens, direct_wpq, ix->tune.ub_static, w.hotbits.as<uint32_t>() + (size_t)B * (KP / 32),                           \
  w.ub_thr2.as<uint32_t>() + 2 * B, d_qoff)
      if (!ix->code_wide) {
        if (RB == 32) NP_LAUNCH_UBF(32, uint16_t);
        else NP_LAUNCH_UBF(64, uint16_t);
      } else {
        if (RB == 32) NP_LAUNCH_UBF(32, uint32_t);
        else NP_LAUNCH_UBF(64, uint32_t);
      }
#undef NP_LAUNCH_UBF
      return;
    }
#define NP_LAUNCH_UB(ROWB, CT, NT)                                                                                        \
  approx_ub_kernel<ROWB, CT, NT><<<grid, 256, 0, st>>>(w.QCU.as<uint8_t>(), KP, meta, begin, count, n_all,   \
                                                   rpx, r, max_rounds, (const CT*)ix->d_ucodes, w.qflag.as<uint32_t>(),   \
                                                   cs->n_sel, U, hist, hshift, cursor, sl, tk, B,                \
                                                   ix->tune.ub_steal, w.ctr.as<Counters>(), count_tokens, direct_wpq,   \
                                                   ix->tune.ub_static)
#define NP_LAUNCH_UB_RB(CT, NT)                 \
  do {                                          \
    if (RB == 32) NP_LAUNCH_UB(32, CT, NT);     \
    else if (RB == 64) NP_LAUNCH_UB(64, CT, NT);   \
    else if (RB == 128) NP_LAUNCH_UB(128, CT, NT); \
    else NP_LAUNCH_UB(256, CT, NT);             \
  } while (0)
    if (!ix->code_wide) {
      if (oob) NP_LAUNCH_UB_RB(uint16_t, 2);
      else if (ix->tune.ub_nt == 1) NP_LAUNCH_UB_RB(uint16_t, 1);
      else NP_LAUNCH_UB_RB(uint16_t, 0);
    } else {
      if (oob) NP_LAUNCH_UB_RB(uint32_t, 2);
      else if (ix->tune.ub_nt == 1) NP_LAUNCH_UB_RB(uint32_t, 1);
      else NP_LAUNCH_UB_RB(uint32_t, 0);
    }
#undef NP_LAUNCH_UB_RB
#undef NP_LAUNCH_UB
  };

  // slack of the bound (np_kernels.h); the batched path's mat-vec scores differ from the GEMM's by < 1 more unit
  // (per query the bracket is Lq + 2 with its OWN token count: padding tokens contribute exactly 0 to both sides, so the
  // slice's longest query bounds it -- 48-token queries in 64-token rows keep 50, not 66)
  // (that unit count follows gcut_kernel's bound e >= |G - R| = 1.5 (152 Lq + 2 Lq^2) 2^-24 s, in table units of s / 254:
  // below one unit up to 64 tokens, four at 256)
  const float lqf = (float)maxLq;
  const int slack = maxLq + 2 +
                    (batched ? std::max(1, (int)std::ceil(1.5f * (152.0f * lqf + 2.0f * lqf * lqf) * 5.9604645e-8f * 254.0f)) : 0);
  // ---- S3: posting-list union (bitmap), per-chunk counts, round plan
  RoundPlan rp;
  rp.n_cand = w.n_cand.as<int32_t>();
  rp.cand_base = w.cand_base.as<int64_t>();
  rp.round_of = w.round_of.as<int32_t>();
  rp.round_tab = w.round_tab.as<int32_t>();
  rp.order = w.q_order.as<int32_t>();
  const bool have_cands = !cs->empty_subset && ix->n_docs > 0;
  GainP gp{};
  if (have_cands && gain_path) {
    uint32_t* gs = w.gsmall.as<uint32_t>();
    uint32_t* g_base = gs;
    int32_t* g_nraw = reinterpret_cast<int32_t*>(gs + 4 * B);
    uint32_t* g_thr0 = gs + 5 * B;
    uint32_t* g_cut0 = gs + 6 * B;
    int32_t* g_ns0 = reinterpret_cast<int32_t*>(gs + 7 * B);
    int32_t* g_nemit = reinterpret_cast<int32_t*>(gs + 8 * B);
    int32_t* g_ndirect = reinterpret_cast<int32_t*>(gs + 9 * B);
    uint32_t* g_cursor0 = gs + 12 * B;
    RoundPlan rp0;          // the S0 launch: one round, identity order, slices of s0cap records
    rp0.n_cand = g_ns0;
    rp0.round_of = reinterpret_cast<int32_t*>(gs + 10 * B);
    rp0.order = reinterpret_cast<int32_t*>(gs + 11 * B);
    int32_t* g_nhi = reinterpret_cast<int32_t*>(gs + 13 * B);
    int32_t* g_nhi_emit = reinterpret_cast<int32_t*>(gs + 14 * B);
    int32_t* g_nmarg = reinterpret_cast<int32_t*>(gs + 15 * B);
    uint32_t* g_lcut = gs + 16 * B;
    rp0.round_tab = reinterpret_cast<int32_t*>(gs + 17 * B);
    unsigned long long* g_report = reinterpret_cast<unsigned long long*>(gs + ((17 * (size_t)B + 4 + 1) & ~(size_t)1));
    rp0.cand_base = reinterpret_cast<int64_t*>(gs + gs_words + (gs_words & 1));
    uint32_t* hist0 = w.ghist.as<uint32_t>();
    uint32_t* hist_s0 = hist0 + (size_t)B * 256;
    const uint32_t* g_tauq = w.tauq.as<uint32_t>();
    gp.cells = w.cells.as<uint32_t>();
    gp.n_cells = w.n_cells.as<int32_t>();
    if (gain_deep) {
      const uint32_t* gd = w.gdeep.as<uint32_t>();
      g_tauq = gd + (size_t)B * G;
      gp.n_cells = reinterpret_cast<const int32_t*>(gd + (size_t)B * G + (size_t)B * LQP);
      gp.cells = gd + (size_t)2 * B * G + (size_t)B * LQP + (size_t)B;
    }
    // the cells whose documents are candidates: the search's own marks, or -- with a threshold -- the cells it kept
    const uint32_t* g_real = !gain_deep ? nullptr
                             : (prm.has_threshold ? w.gdeep.as<uint32_t>() + (size_t)B * G + (size_t)B * LQP + (size_t)B : w.cellbits.as<uint32_t>());
    gp.KP = KP;
    gp.ivf_off = ix->d_ivf_offsets;
    gp.ivf = ix->d_ivf;
    gp.split = ix->d_ivf_split;
    gp.R1 = ix->n_ranges + 1;
    gp.gain = w.gain.as<uint16_t>();
    gp.gbase = g_base;
    gp.hshift = hshift;
    gp.hist0 = hist0;
    gp.n_raw = g_nraw;
    gp.thr = g_thr0;
    gp.s0_meta = w.s0_meta.as<uint4>();
    gp.s0cap = s0cap;
    gp.ucodes = ix->d_ucodes;
    gp.code_wide = ix->code_wide;
    gp.ublock_stride = ix->ublock_stride;
    gp.ovf_base = (int64_t)ix->n_docs * ix->ublock_stride;
    gp.cand = w.cand.as<uint32_t>();
    gp.n_emit = g_nemit;
    gp.rp = rp;
    gp.ctr = w.ctr.as<Counters>();
    gp.lvl = w.gacc.as<uint8_t>();
    gp.n_ranges = ix->n_ranges;
    gp.n_hi = g_nhi;
    gp.n_marg = g_nmarg;
    const size_t glds = (size_t)NP_GAIN_RANGE * 2;
    NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gain_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
    if (RB == 32)
      gain_prep_kernel<32><<<B, 256, 0, st>>>(w.QCU.as<uint8_t>(), KP, gp.cells, gp.n_cells, g_tauq, LQP, w.qinv.as<float>(),
                                              d_qoff, w.gain.as<uint16_t>(), g_base, B, s0cap, rp0, hshift, g_real);
    else
      gain_prep_kernel<64><<<B, 256, 0, st>>>(w.QCU.as<uint8_t>(), KP, gp.cells, gp.n_cells, g_tauq, LQP, w.qinv.as<float>(),
                                              d_qoff, w.gain.as<uint16_t>(), g_base, B, s0cap, rp0, hshift, g_real);
    const dim3 ggrid((unsigned)ix->n_ranges, (unsigned)B), egrid((unsigned)ix->n_ranges, (unsigned)B);
    gain_sweep_kernel<<<ggrid, 1024, glds, st>>>(gp);                                         // accumulators, histogram of U0, counts
    gain_thr_kernel<<<B, 64, 0, st>>>(hist0, s0_target, s0cap, g_nraw, w.qflag.as<uint32_t>(), g_thr0, g_nhi, g_ns0);
    gp.n_emit = g_nhi_emit;
    gain_emit_kernel<1><<<egrid, 256, 0, st>>>(gp, 0);                                        // S0: records of the best bounds
    gp.n_emit = g_nemit;
    {
      int32_t* sl0 = w.xcd_slots.as<int32_t>() + (size_t)max_rounds * 3 * slot_words;
      launch_ub_at(rp0, 0, 1, sl0, sl0 + 8 * (B + 1), g_cursor0, w.s0_meta.as<uint4>(), nullptr, g_ns0, g_ns0, w.s0_u.as<uint16_t>(), hist_s0,
                   0, ix->tune.s3_gain_direct, false);                                        // exact bounds of S0 (histogram: lower bounds)
    }
    ub_thr_kernel<<<B, 256, 0, st>>>(hist_s0, hshift, slack, cs->n_sel, g_ns0, rp0, 0, w.qflag.as<uint32_t>(), g_cut0);   // tau0 - slack
    gain_count_kernel<<<B, 64, 0, st>>>(g_cut0, hist0, g_base, g_nraw, g_lcut, g_ndirect, w.ctr.as<Counters>(),
                                        ix->tune.s3_gain == 1 ? w.h_gain : nullptr, g_report, B);   // the cut in levels, candidates kept
    plan_rounds_kernel<<<1, 256, 0, st>>>(nullptr, 0, B, pool, max_rounds, rp, w.ctr.as<Counters>(), g_ndirect);
    gp.thr = g_lcut;
  }
  if (have_cands && !gain_path) {
    if (ix->tune.s3_slices) {
      // bitmap ranges in LDS (mark_slices_kernel): ranges of <= 32 chunks, enough of them to fill the chip, at most 16
      // sweeps of the posting lists per query beyond what the range size forces
      const int smin = (nchunks + 31) / 32, swant = std::min(16, (512 + B - 1) / B);
      const int ns0 = std::max(smin, std::min(swant, nchunks));
      const int slice_chunks = (nchunks + ns0 - 1) / ns0;
      const int nslices = (nchunks + slice_chunks - 1) / slice_chunks;
      const size_t lds = (size_t)slice_chunks * NP_CHUNK_WORDS * 4;
      if (lds > 32 * 1024)
        NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mark_slices_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      mark_slices_kernel<<<dim3(nslices, B), 1024, lds, st>>>(w.cells.as<uint32_t>(), w.n_cells.as<int32_t>(), KP,
                                                              ix->d_ivf_offsets, ix->d_ivf,
                                                              have_subset ? w.subset_bits.as<uint32_t>() : nullptr, NW,
                                                              slice_chunks, nchunks, w.docbits.as<uint32_t>(),
                                                              w.chunk_counts.as<int32_t>(), w.ctr.as<Counters>(),
                                                              (ix->ivf_sorted && ix->tune.s3_bisect) ? 1 : 0);
    } else {
      mark_candidates_kernel<<<dim3(128, B), 256, 0, st>>>(w.cells.as<uint32_t>(), w.n_cells.as<int32_t>(), KP,
                                                           ix->d_ivf_offsets, ix->d_ivf,
                                                           have_subset ? w.subset_bits.as<uint32_t>() : nullptr, NW,
                                                           w.docbits.as<uint32_t>(), w.ctr.as<Counters>());
      count_chunks_kernel<<<dim3(nchunks, B), 256, 0, st>>>(w.docbits.as<uint32_t>(), NW, nchunks,
                                                            w.chunk_counts.as<int32_t>());
    }
    plan_rounds_kernel<<<1, 256, 0, st>>>(w.chunk_counts.as<int32_t>(), nchunks, B, pool, max_rounds, rp,
                                          w.ctr.as<Counters>());
  }
  if (have_cands) {
    // Lambda, the thresholds of the 8 planes and the hot bitmap in one launch, then the plane rows of the hot centroids -- AFTER
    // the round plan: the hot share of a query follows its candidate count (hot_levels_kernel), which S3 has just counted
    if (two_level && use_planes) {
      hot_levels_kernel<<<dim3((unsigned)std::min<int64_t>(std::max<int64_t>((KP >> 5) / 256, 1), 16), B), 256, 0, st>>>(
          w.chist.as<uint32_t>(), ix->K, ix->tune.s4_hot, w.cmaxu.as<uint8_t>(), KP, ix->tune.s4_pexp, w.ub_thr2.as<uint32_t>() + B,
          w.levels.as<uint32_t>(), w.hotbits.as<uint32_t>(), s4_warm, w.ub_thr2.as<uint32_t>() + 2 * B,
          w.hotbits.as<uint32_t>() + (size_t)B * (KP / 32), ix->tune.s4_hot_auto ? w.n_cand.as<int32_t>() : nullptr,
          ix->tune.s4_hot_auto);
      const dim3 pg((unsigned)std::min<int64_t>((KP + 2047) / 2048, 64), B);
      if (RB == 32)
        hot_planes_kernel<32><<<pg, 256, 0, st>>>(w.QCU.as<uint8_t>(), KP, w.cmaxu.as<uint8_t>(), w.ub_thr2.as<uint32_t>() + B,
                                                  w.levels.as<uint32_t>(), w.planes.as<uint32_t>());
      else
        hot_planes_kernel<64><<<pg, 256, 0, st>>>(w.QCU.as<uint8_t>(), KP, w.cmaxu.as<uint8_t>(), w.ub_thr2.as<uint32_t>() + B,
                                                  w.levels.as<uint32_t>(), w.planes.as<uint32_t>());
    }
  }
  SelectP sp;
  sp.approx = w.approx.as<float>();
  sp.cand = w.cand.as<uint32_t>();
  sp.cand_step = 1;
  sp.rp = rp;
  sp.n_cand = w.n_cand.as<int32_t>();
  sp.doc_begin = ix->doc_begin;
  sp.n_sel = cs->n_sel;
  sp.NSELP = cs->NSELP;
  sp.sel_keys = w.sel_keys.as<uint64_t>();
  sp.sel_doc = w.sel_doc.as<uint32_t>();
  sp.nsel_out = w.nsel.as<int32_t>();
  sp.ctr = nullptr;
  const size_t sel_lds = (size_t)cs->NSELP * 8;
  if (cs->n_sel > 0 && sel_lds > 48 * 1024)
    NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
  // ---- per round: S3 compaction -> S4 -> S5 (stage events bracket round 0, which holds the whole batch unless
  // the candidates overflow the pool; later rounds are charged to S5)
  for (int r = 0; r < (have_cands ? max_rounds : 0); ++r) {
    // two-level filter: bare ids only -- the hot level finds a document's list block from the id and writes the 16-B records
    // itself (no record gather here: a 128-B line per candidate at 1.9 % density was this kernel's whole cost)
    const bool ids_only = two_level && ix->ublock_stride > 0;
    if (gain_path)   // the candidates that pass the zeroth level's cut (every candidate where it does not apply), bare ids
      gain_emit_kernel<2><<<dim3((unsigned)ix->n_ranges, (unsigned)B), 256, 0, st>>>(gp, r);
    else
    compact_kernel<<<dim3(nchunks, B), 256, 0, st>>>(w.docbits.as<uint32_t>(), NW, nchunks, w.chunk_counts.as<int32_t>(),
                                                     (use_filter && !ids_only) ? nullptr : w.cand.as<uint32_t>(), rp, r,
                                                     ids_only ? nullptr : ix->d_doc_meta, w.cand_meta.as<uint4>());
    if (cs->timed && r == 0) NP_HIP(hipEventRecord(cs->ctx->ev[3], st));
    if (use_filter) {
      // hand-out state of the (up to three) filter launches of this round: slots = -1 (empty), ticket = -1; cursors 0
      auto xslots = [&](int lvl) { return w.xcd_slots.as<int32_t>() + ((size_t)r * 3 + lvl) * slot_words; };
      auto xcursor = [&](int lvl) { return w.ub_cursor.as<uint32_t>() + (size_t)lvl * B; };
      auto launch_ub = [&](int lvl, const uint4* meta, const int32_t* begin, const int32_t* count, uint16_t* U, uint32_t* hist,
                           int count_tokens, int direct_wpq, bool floor_rows = false) {
        launch_ub_at(rp, r, max_rounds, xslots(lvl), xslots(lvl) + 8 * (B + 1), xcursor(lvl), meta, begin, count, w.n_cand.as<int32_t>(), U,
                     hist, count_tokens, direct_wpq, floor_rows);
      };
      const unsigned ncut = (unsigned)std::min<int64_t>(ix->tune.ub_ncut, std::max<int64_t>(1, ix->n_docs / 16384));
      CutP cp{};
      cp.hshift = hshift;
      cp.all_src = w.cand_meta.as<uint4>();
      cp.n_all = w.n_cand.as<int32_t>();
      if (!two_level) {
        launch_ub(0, w.cand_meta.as<uint4>(), nullptr, w.n_cand.as<int32_t>(), w.ub.as<uint16_t>(), w.ub_hist.as<uint32_t>(), 1, 0);
        ub_thr_kernel<<<B, 256, 0, st>>>(w.ub_hist.as<uint32_t>(), hshift, slack, cs->n_sel, w.n_cand.as<int32_t>(), rp, r,
                                         w.qflag.as<uint32_t>(), w.ub_thr.as<uint32_t>());
        cp.src = w.cand_meta.as<uint4>();
        cp.n_src_a = w.n_cand.as<int32_t>();
        cp.U = w.ub.as<uint16_t>();
        cp.lo = w.ub_thr.as<uint32_t>();
        cp.zero_mode = 0;
        cp.dst = w.surv_meta.as<uint4>();
        cp.n_dst = w.n_surv.as<int32_t>();
        cp.ctr = w.ctr.as<Counters>();
        ub_cut_kernel<<<dim3(ncut, B), 256, 0, st>>>(cp, rp, r);
      } else {
        // level 1: the hot bound U' of every candidate
        {
          int32_t* sl = xslots(0);
          const size_t dyn = (size_t)(KP / 8);
#define NP_LAUNCH_HOT(ROWB, CT)                                                                                          \
  do {                                                                                                                   \
    if (dyn > 16 * 1024)                                                                                                 \
      NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&approx_hot_kernel<ROWB, CT>),                             \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                                 \
    approx_hot_kernel<ROWB, CT><<<8 * nbx, 256, dyn, st>>>(w.QCU.as<uint8_t>(), ix->K, KP, w.cmaxu.as<uint8_t>(),         \
                                                           w.ub_thr2.as<uint32_t>() + B, w.cand.as<uint32_t>(),                \
                                                           w.cand_meta.as<uint4>(), ix->ublock_stride,                         \
                                                           (int64_t)ix->n_docs * ix->ublock_stride,                            \
                                                           w.n_cand.as<int32_t>(), rp, r, max_rounds, (const CT*)ix->d_ucodes, \
                                                           w.qflag.as<uint32_t>(), d_qoff, cs->n_sel, w.ub.as<uint16_t>(),  \
                                                           w.ub_hist.as<uint32_t>(), hshift, xcursor(0), sl, sl + 8 * (B + 1), \
                                                           B, ix->tune.ub_steal, w.ctr.as<Counters>(), ix->tune.s4_probe,   \
                                                           ix->tune.hot_static);                                           \
  } while (0)
#define NP_LAUNCH_HOT_RB(CT)                    \
  do {                                          \
    if (RB == 32) NP_LAUNCH_HOT(32, CT);        \
    else if (RB == 64) NP_LAUNCH_HOT(64, CT);   \
    else if (RB == 128) NP_LAUNCH_HOT(128, CT); \
    else NP_LAUNCH_HOT(256, CT);                \
  } while (0)
#define NP_LAUNCH_HOTP_W(ROWB, CT, LPDV, PFV, DPIV, QMV, WPBV, NBX)                                                        \
  do {                                                                                                                   \
    const size_t bm = sizeof(CT) == 2 ? 0 : (size_t)(((KP >> 5) + 3) & ~(int64_t)3) * 4;   /* u16 codes: static bitmap */   \
    /* idle lanes of the last packed staging instruction write 16 B each past the rows it fills (1 KiB per instruction);     \
       the one-block-per-instruction fallback overruns by at most 256 B */                                                  \
    const size_t rowb = (size_t)ix->ublock_stride * sizeof(CT) + 16;                                                          \
    const int slack = (int)std::max<int64_t>(256, 1024 - (int64_t)(DPIV) * (int64_t)rowb);                                   \
    const size_t dynp = bm + (size_t)(WPBV) * ((64 / LPDV) * rowb + (size_t)slack);                                          \
    if (dynp > 16 * 1024)                                                                                                \
      NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&approx_hotp_kernel<ROWB, CT, LPDV, PFV, DPIV, QMV, WPBV>), \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynp));                                \
    approx_hotp_kernel<ROWB, CT, LPDV, PFV, DPIV, QMV, WPBV><<<8 * (NBX), 64 * (WPBV), dynp, st>>>(                 \
        w.planes.as<uint32_t>(), ix->K, KP, w.hotbits.as<uint32_t>(), w.ub_thr2.as<uint32_t>() + B, w.levels.as<uint32_t>(), \
        w.cand.as<uint32_t>(), w.cand_meta.as<uint4>(), ix->ublock_stride, (int64_t)ix->n_docs * ix->ublock_stride,        \
        w.n_cand.as<int32_t>(), rp, r, max_rounds, (const CT*)ix->d_ucodes, w.qflag.as<uint32_t>(), d_qoff, cs->n_sel,     \
        w.ub.as<uint16_t>(), w.ub_hist.as<uint32_t>(), hshift, sl, sl + 8 * (B + 1), B, w.ctr.as<Counters>(), slack,       \
        ix->tune.s4_probe, gain_path ? 1 : 0);                                                                                            \
  } while (0)
#define NP_LAUNCH_HOTP(ROWB, CT, LPDV, PFV, DPIV, QMV) NP_LAUNCH_HOTP_W(ROWB, CT, LPDV, PFV, DPIV, QMV, 4, pnbx)
  // lanes per document: 2 (32 documents per claim; blocks of at most 256 bytes) or 4 (16 per claim: half the LDS rows per
  // wave; the only choice for 512-byte blocks).  Documents per staging instruction from the block size (16 B per lane,
  // block + 16 B of padding per row)
#define NP_LAUNCH_HOTP_L(ROWB, CT)                                      \
  do {                                                                  \
    const int sb = ix->ublock_stride * (int)sizeof(CT);                 \
    if (plpd == 2) {                                                    \
      if (sb <= 240 && ix->tune.s4_qm) NP_LAUNCH_HOTP(ROWB, CT, 2, 2, 4, 1);   \
      else if (sb <= 240) NP_LAUNCH_HOTP(ROWB, CT, 2, 2, 4, 0);         \
      else NP_LAUNCH_HOTP(ROWB, CT, 2, 2, 2, 0);                        \
    } else if (sb <= 240) NP_LAUNCH_HOTP(ROWB, CT, 4, 1, 4, 0);         \
    else if (sb <= 496) NP_LAUNCH_HOTP(ROWB, CT, 4, 1, 2, 0);           \
    else NP_LAUNCH_HOTP(ROWB, CT, 4, 1, 1, 0);                          \
  } while (0)
          const int plpd = (ix->ublock_stride > old_cap || ix->tune.s4_lpd == 4) ? 4 : 2;
          const unsigned pnbx = (unsigned)ix->tune.s4_pnbx;   // workgroups per XCD of the plane kernel
          if (cs->timed && r == 0) NP_HIP(hipEventRecord(cs->ctx->ev[8], st));
          if (use_planes) {
            if (!ix->code_wide) {
              if (RB == 32) NP_LAUNCH_HOTP_L(32, uint16_t);
              else NP_LAUNCH_HOTP_L(64, uint16_t);
            } else if ((KP >> 3) >= 32 * 1024 && plpd == 4 && ix->ublock_stride * 4 > 240 && ix->ublock_stride * 4 <= 496) {
              // u32 code lists with a hot bitmap of 32 KB or more (K >= 2^18), the usual 4-lane form: ONE workgroup of 12 waves per
              // CU shares the bitmap (with 4-wave workgroups the 64 KB of K = 2^19 left one per CU: 4 waves)
              if (RB == 32) NP_LAUNCH_HOTP_W(32, uint32_t, 4, 1, 2, 0, 12, 32);
              else NP_LAUNCH_HOTP_W(64, uint32_t, 4, 1, 2, 0, 12, 32);
            } else {
              if (RB == 32) NP_LAUNCH_HOTP_L(32, uint32_t);
              else NP_LAUNCH_HOTP_L(64, uint32_t);
            }
          } else if (!ix->code_wide) NP_LAUNCH_HOT_RB(uint16_t);
          else NP_LAUNCH_HOT_RB(uint32_t);
#undef NP_LAUNCH_HOTP_L
#undef NP_LAUNCH_HOTP
#undef NP_LAUNCH_HOTP_W
#undef NP_LAUNCH_HOT_RB
#undef NP_LAUNCH_HOT
          if (cs->timed && r == 0) {
            NP_HIP(hipEventRecord(cs->ctx->ev[9], st));
            cs->hot_timed = true;
          }
        }
        // S1 = the n_sel documents with the largest U' (whole bins): exact bound -> tau
        ub_thr_kernel<<<B, 256, 0, st>>>(w.ub_hist.as<uint32_t>(), hshift, 0, cs->n_sel, w.n_cand.as<int32_t>(), rp, r,
                                         w.qflag.as<uint32_t>(), w.ub_thr.as<uint32_t>());
        cp.src = w.cand_meta.as<uint4>();
        cp.n_src_a = w.n_cand.as<int32_t>();
        cp.U = w.ub.as<uint16_t>();
        cp.lo = w.ub_thr.as<uint32_t>();
        cp.zero_mode = 1;
        cp.dst = w.list_meta.as<uint4>();
        cp.n_dst = w.n_l1.as<int32_t>();
        ub_cut_kernel<<<dim3(ncut, B), 256, 0, st>>>(cp, rp, r);
        launch_ub(1, w.list_meta.as<uint4>(), nullptr, w.n_l1.as<int32_t>(), w.ub2.as<uint16_t>(), w.ub_hist2.as<uint32_t>(), 0,
                  ix->tune.ub_direct);
        ub_thr_kernel<<<B, 256, 0, st>>>(w.ub_hist2.as<uint32_t>(), hshift, slack, cs->n_sel, w.n_cand.as<int32_t>(), rp, r,
                                         w.qflag.as<uint32_t>(), w.ub_thr2.as<uint32_t>());
        // S2 = the other documents with U' >= tau: exact bound too (appended behind S1)
        cp.lo = w.ub_thr2.as<uint32_t>();
        cp.hi = w.ub_thr.as<uint32_t>();
        cp.dst_begin = w.n_l1.as<int32_t>();
        cp.n_dst = w.n_l2.as<int32_t>();
        ub_cut_kernel<<<dim3(ncut, B), 256, 0, st>>>(cp, rp, r);
        launch_ub(2, w.list_meta.as<uint4>(), w.n_l1.as<int32_t>(), w.n_l2.as<int32_t>(), w.ub2.as<uint16_t>(), w.ub_hist2.as<uint32_t>(), 0, 0,
                  true);
        // every document with U' >= tau now has its exact bound in the histogram: the cut over S1 + S2 is the single-level
        // filter's cut (the n_sel-th largest exact U of ALL candidates lies in S1 + S2), tau can only rise
        ub_thr_kernel<<<B, 256, 0, st>>>(w.ub_hist2.as<uint32_t>(), hshift, slack, cs->n_sel, w.n_cand.as<int32_t>(), rp, r,
                                         w.qflag.as<uint32_t>(), w.ub_thr2.as<uint32_t>());
        // survivors: the documents of S1 + S2 whose exact bound reaches tau (every candidate where the filter does not apply)
        cp.src = w.list_meta.as<uint4>();
        cp.n_src_a = w.n_l1.as<int32_t>();
        cp.n_src_b = w.n_l2.as<int32_t>();
        cp.U = w.ub2.as<uint16_t>();
        cp.lo = w.ub_thr2.as<uint32_t>();
        cp.hi = nullptr;
        cp.zero_mode = 0;
        cp.dst = w.surv_meta.as<uint4>();
        cp.dst_begin = nullptr;
        cp.n_dst = w.n_surv.as<int32_t>();
        cp.ctr = w.ctr.as<Counters>();
        ub_cut_kernel<<<dim3(ncut, B), 256, 0, st>>>(cp, rp, r);
      }
      // exact f32 approximate scores of the survivors only
      launch_approx(st, ix, w, d_qoff, B, LQP, w.surv_meta.as<uint4>(), w.n_surv.as<int32_t>(), rp, r, max_rounds, nullptr);
      sp.cand = reinterpret_cast<const uint32_t*>(w.surv_meta.p);
      sp.cand_step = 4;
      sp.n_cand = w.n_surv.as<int32_t>();
      sp.ctr = w.ctr.as<Counters>();
      if (batched && !s1_split) {
        // reference arithmetic of the batched path: G-valued cut with a rounding margin, then the mat-vec scores
        // of what is left (the candidate records of this round are consumed: their array takes the second list)
        gcut_kernel<<<B, 1024, 0, st>>>(w.approx.as<float>(), w.surv_meta.as<uint4>(), w.n_surv.as<int32_t>(), rp, r, cs->n_sel,
                                        w.qinv.as<float>(), w.qflag.as<uint32_t>(), d_qoff, w.cand_meta.as<uint4>(),
                                        w.n_list2.as<int32_t>());
        launch_matvec(st, ix, w, d_q, d_qoff, B, w.cand_meta.as<uint4>(), w.n_list2.as<int32_t>(), rp, r);
        sp.cand = reinterpret_cast<const uint32_t*>(w.cand_meta.p);
        sp.n_cand = w.n_list2.as<int32_t>();
      }
    } else if (ix->T > 0) {
      if (batched && !s1_split) {   // debug trace / filter off: the mat-vec score of every candidate
        launch_matvec(st, ix, w, d_q, d_qoff, B, w.cand_meta.as<uint4>(), w.n_cand.as<int32_t>(), rp, r);
        count_work_kernel<<<dim3(32, (unsigned)B), 256, 0, st>>>(w.cand_meta.as<uint4>(), w.n_cand.as<int32_t>(), rp, r,
                                                                 w.ctr.as<Counters>());
      } else {
        launch_approx(st, ix, w, d_qoff, B, LQP, w.cand_meta.as<uint4>(), w.n_cand.as<int32_t>(), rp, r, max_rounds,
                      w.ctr.as<Counters>());
      }
    }
    if (cs->timed && r == 0) NP_HIP(hipEventRecord(cs->ctx->ev[4], st));
    if (cs->n_sel > 0) {
      sp.round = r;
      select_kernel<<<B, 1024, sel_lds, st>>>(sp);
    }
  }
  if (cs->timed && !have_cands) {
    NP_HIP(hipEventRecord(cs->ctx->ev[3], st));
    NP_HIP(hipEventRecord(cs->ctx->ev[4], st));
  }
  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[5], st));
  NP_HIP(hipGetLastError());
  return NP_OK;
}

// S6..S7.  d_cut may be NULL (keep every locally selected document).
static int phase_b(const DeviceIndex* ix, CallState* cs, const int32_t* d_qoff, const uint64_t* d_cut,
                   int64_t* d_out_ids, float* d_out_scores, uint64_t* d_out_keys, int32_t* d_out_counts) {
  Workspace& w = *cs->ctx->ws;
  hipStream_t st = cs->stream;
  const int B = cs->B;
  if (B == 0) return NP_OK;
  if (cs->n_sel > 0) {
    ExactP ep;
    ep.Qt = w.Qt.as<float>();
    ep.Qb = w.Qb.as<__bf16>();
    ep.Qb_lo = w.Qbl.as<__bf16>();
    ep.QCT = w.QCT.as<float>();
    ep.KP = ix->KP;
    ep.inv_norm = ix->d_inv_norm;
    ep.qoff = d_qoff;
    ep.LQP = cs->LQP;
    ep.centroids = ix->d_centroids;
    ep.wlut = ix->d_wlut;
    ep.codes = ix->codes();
    ep.residuals = ix->d_residuals;
    ep.doc_off = ix->d_doc_offsets;
    ep.sel_keys = w.sel_keys.as<uint64_t>();
    ep.sel_doc = w.sel_doc.as<uint32_t>();
    ep.nsel = w.nsel.as<int32_t>();
    ep.cut = d_cut;
    ep.n_sel = cs->n_sel;
    ep.exact = w.exact.as<float>();
    ep.ctr = w.ctr.as<Counters>();
    ep.xcd_B = 0;
    ep.gx = 0;
    ep.qflag = w.qflag.as<uint32_t>();
    ep.fast_ok = ix->s6_fast_ok ? 1 : 0;
    ep.qt0 = 0;
    ep.acc = 0;
    ep.pad_ss = ix->pad_ss;
    switch (ix->dim) {
      case 32: NP_TRY((launch_exact_nb<32>(st, ix, ep, B, cs->prm.precision, ix->nbits))); break;
      case 64: NP_TRY((launch_exact_nb<64>(st, ix, ep, B, cs->prm.precision, ix->nbits))); break;
      case 96: NP_TRY((launch_exact_nb<96>(st, ix, ep, B, cs->prm.precision, ix->nbits))); break;
      default: NP_TRY((launch_exact_nb<128>(st, ix, ep, B, cs->prm.precision, ix->nbits))); break;
    }
  }
  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[6], st));
  {
    TopkP tp;
    tp.exact = w.exact.as<float>();
    tp.sel_keys = w.sel_keys.as<uint64_t>();
    tp.sel_doc = w.sel_doc.as<uint32_t>();
    tp.nsel = w.nsel.as<int32_t>();
    tp.cut = d_cut;
    tp.n_sel = cs->n_sel;
    tp.NSELP = cs->NSELP;
    tp.top_k = cs->prm.top_k;
    tp.doc_begin = ix->doc_begin;
    tp.doc_off = ix->d_doc_offsets;
    tp.ctr = w.ctr.as<Counters>();
    tp.out_ids = d_out_ids;
    tp.out_scores = d_out_scores;
    tp.out_keys = d_out_keys;
    tp.out_counts = d_out_counts;
    const size_t lds = (size_t)cs->NSELP * 8;
    if (lds > 48 * 1024)
      NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    topk_kernel<<<B, 1024, lds, st>>>(tp);
  }
  if (cs->timed) NP_HIP(hipEventRecord(cs->ctx->ev[7], st));
  NP_HIP(hipGetLastError());
  return NP_OK;
}

static int begin_use(CallState* cs, void* user_stream) {
  Context* c = cs->ctx;
  cs->stream = user_stream ? (hipStream_t)user_stream : c->stream;
  // the workspace's previous use (on whatever stream) must have finished before this one touches it
  if (c->ws->done_valid) NP_HIP(hipStreamWaitEvent(cs->stream, c->ws->done, 0));
  return NP_OK;
}
static int end_use(CallState* cs) {
  Context* c = cs->ctx;
  NP_HIP(hipEventRecord(c->ws->done, cs->stream));
  c->ws->done_valid = true;
  return NP_OK;
}
// Every exit path of a call that touched a workspace records its `done` event (a failed call may already have
// queued work on the stream) and only then hands the context back.
struct UseGuard {
  const DeviceIndex* ix;
  CallState* cs;
  bool began = false;
  ~UseGuard() {
    if (began && cs->stream) (void)end_use(cs);
    if (cs->ctx) release_context(ix, cs->ctx);
  }
};

static int lqp_of(const int32_t* h_qoff, int B) {
  int maxLq = 1;
  for (int b = 0; b < B; ++b) maxLq = std::max(maxLq, h_qoff[b + 1] - h_qoff[b]);
  return std::min((maxLq + 31) / 32 * 32, 32 * NP_MAX_QT);
}

static int slice_size(const DeviceIndex* ix, const int32_t* h_qoff, int B, const np_search_params* prm) {
  return std::max(1, std::min(plan_workspace(ix, B, lqp_of(h_qoff, B), prm).S, std::max(B, 1)));
}

// Whole batch on device buffers, sliced.  No host synchronisation.
static int run_device(const DeviceIndex* ix, CallState* cs, const float* d_q, const int32_t* d_qoff,
                      const int32_t* h_qoff, int B, const np_search_params* prm, const int64_t* d_subset,
                      int64_t subset_len, int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts) {
  const int S = slice_size(ix, h_qoff, B, prm);
  for (int s0 = 0; s0 < B; s0 += S) {
    cs->B = std::min(S, B - s0);
    cs->prm = *prm;
    NP_TRY(phase_a(ix, cs, d_q, d_qoff + s0, h_qoff + s0, d_subset, subset_len));
    NP_TRY(phase_b(ix, cs, d_qoff + s0, nullptr, d_out_ids + (int64_t)s0 * prm->top_k,
                   d_out_scores + (int64_t)s0 * prm->top_k, nullptr, d_out_counts + s0));
  }
  return NP_OK;
}

// ---- document-sharded exchange: the strided forms np_dist.hip uses (one record per rank with a status trailer) -----
int select_cut_strided(const DeviceIndex* ix, const uint64_t* d_all_keys, int64_t rank_stride, int64_t status_off, int G,
                       int B, int n_sel, uint64_t* d_cut, hipStream_t st) {
  if (!ix || !d_all_keys || !d_cut || G < 1 || B < 0 || n_sel < 0 || rank_stride < (int64_t)B * n_sel) {
    set_error("select_cut: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return NP_OK;
  DeviceGuard g(ix->device);
  const int NP2 = next_pow2(std::max(G * n_sel, 1));
  const size_t lds = (size_t)NP2 * 8;
  if (lds > 128 * 1024) {
    set_error("select_cut: G*n_sel = %d exceeds the 16384-key merge window", G * n_sel);
    return NP_ERR_SEARCH;
  }
  if (lds > 48 * 1024)
    NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_cut_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  select_cut_kernel<<<B, 1024, lds, st>>>(d_all_keys, rank_stride, status_off, G, B, n_sel, NP2, d_cut);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

bool select_cut_fits(int G, int n_sel) { return (size_t)next_pow2(std::max(G * n_sel, 1)) * 8 <= 128 * 1024; }

int merge_packed_status(const DeviceIndex* ix, const void* d_records, int64_t record_bytes, int64_t off_keys,
                        int64_t off_scores, int64_t off_counts, int64_t off_status, uint64_t* h_status, int G, int B,
                        int top_k, int64_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts, hipStream_t st) {
  if (!ix || !d_records || !d_out_counts || G < 1 || B < 0 || top_k < 0 || (record_bytes & 7) || (off_keys & 7) ||
      (off_scores & 3) || (off_counts & 3) || (off_status >= 0 && (off_status & 7))) {
    set_error("merge_packed: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return NP_OK;
  DeviceGuard g(ix->device);
  const char* r = (const char*)d_records;
  merge_topk_kernel<<<B, 256, 0, st>>>((const int64_t*)r, (const float*)(r + off_scores), (const uint64_t*)(r + off_keys),
                                       (const int32_t*)(r + off_counts), record_bytes / 8, record_bytes / 4,
                                       record_bytes / 8, record_bytes / 4, G, B, top_k, d_out_ids, d_out_scores,
                                       d_out_counts, off_status >= 0 ? (const uint64_t*)(r + off_status) : nullptr,
                                       record_bytes / 8, h_status);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

int set_status_word(const DeviceIndex* ix, uint64_t* d_word, uint64_t value, hipStream_t st) {
  DeviceGuard g(ix->device);
  set_status_kernel<<<1, 1, 0, st>>>(d_word, value);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

}  // namespace np

using namespace np;

extern "C" {

int32_t np_hip_n_sel(const np_search_params* p) { return p ? n_sel_of(p) : 0; }

int np_hip_search_batch_device(const np_index* ix, const float* d_queries, const int32_t* d_q_tok_offsets,
                               const int32_t* h_q_tok_offsets, int32_t B, int32_t dim, const np_search_params* params,
                               const int64_t* d_subset, int64_t subset_len, int64_t* d_out_ids, float* d_out_scores,
                               int32_t* d_out_counts, void* stream) {
  clear_error();
  NP_TRY(validate(ix, B, dim, params));
  if (B == 0) return NP_OK;
  if (!d_queries || !d_q_tok_offsets || !h_q_tok_offsets || !d_out_counts || (params->top_k > 0 && (!d_out_ids || !d_out_scores))) {
    set_error("Search failed: NULL buffer");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (subset_len > 0 && !d_subset) {
    set_error("Search failed: subset_len > 0 but subset is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  CallState cs;
  NP_TRY(acquire_context(ix, &cs.ctx));
  int rc = begin_use(&cs, stream);
  if (rc == NP_OK)
    rc = run_device(ix, &cs, d_queries, d_q_tok_offsets, h_q_tok_offsets, B, params, d_subset, subset_len, d_out_ids,
                    d_out_scores, d_out_counts);
  int rc2 = end_use(&cs);
  release_context(ix, cs.ctx);
  return rc != NP_OK ? rc : rc2;
}

int np_hip_search_batch(const np_index* ix, const float* queries, const int32_t* q_tok_offsets, int32_t B, int32_t dim,
                        const np_search_params* params, const int64_t* subset, int64_t subset_len, int64_t* out_ids,
                        float* out_scores, int32_t* out_counts, np_stats* stats) {
  clear_error();
  if (stats) memset(stats, 0, sizeof *stats);
  NP_TRY(validate(ix, B, dim, params));
  if (B == 0) return NP_OK;
  if (!queries || !q_tok_offsets || !out_counts || (params->top_k > 0 && (!out_ids || !out_scores))) {
    set_error("Search failed: NULL buffer");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (subset_len > 0 && !subset) {
    set_error("Search failed: subset_len > 0 but subset is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (q_tok_offsets[0] != 0) {
    set_error("Shape error: q_tok_offsets[0] must be 0");
    return NP_ERR_SHAPE;
  }
  for (int b = 0; b < B; ++b)
    if (q_tok_offsets[b + 1] < q_tok_offsets[b]) {
      set_error("Shape error: q_tok_offsets must be non-decreasing");
      return NP_ERR_SHAPE;
    }
  DeviceGuard g(ix->device);
  CallState cs;
  NP_TRY(acquire_context(ix, &cs.ctx));
  UseGuard guard{ix, &cs};
  Workspace& w = *cs.ctx->ws;
  NP_TRY(begin_use(&cs, nullptr));
  guard.began = true;
  hipStream_t st = cs.stream;
  const int64_t ntok = q_tok_offsets[B];
  const int topk = params->top_k;
  NP_TRY(w.q.reserve((size_t)std::max<int64_t>(ntok, 1) * dim * 4));
  NP_TRY(w.qoff.reserve((size_t)(B + 1) * 4));
  if (subset_len > 0) NP_TRY(w.subset.reserve((size_t)subset_len * 8));
  // results of the whole batch land in one pinned staging area
  const size_t ob_ids = (size_t)B * std::max(topk, 1) * 8, ob_sc = (size_t)B * std::max(topk, 1) * 4,
               ob_cnt = (size_t)B * 4;
  NP_TRY(w.pin(ob_ids + ob_sc + ob_cnt + sizeof(Counters) + 64));
  DevBuf& oi = w.out_ids;   // reserved per slice in phase_a; reserve for the whole batch here
  NP_TRY(oi.reserve(ob_ids));
  NP_TRY(w.out_scores.reserve(ob_sc));
  NP_TRY(w.out_counts.reserve(ob_cnt));
  if (ntok > 0) NP_HIP(hipMemcpyAsync(w.q.p, queries, (size_t)ntok * dim * 4, hipMemcpyHostToDevice, st));
  NP_HIP(hipMemcpyAsync(w.qoff.p, q_tok_offsets, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st));
  if (subset_len > 0) NP_HIP(hipMemcpyAsync(w.subset.p, subset, (size_t)subset_len * 8, hipMemcpyHostToDevice, st));

  // slices: each slice's outputs go to its rows of the batch-wide output buffers
  const int S = slice_size(ix, q_tok_offsets, B, params);
  np_stats acc;
  memset(&acc, 0, sizeof acc);
  char* pin = (char*)w.h_pin;
  int64_t* h_ids = (int64_t*)pin;
  float* h_sc = (float*)(pin + ob_ids);
  int32_t* h_cnt = (int32_t*)(pin + ob_ids + ob_sc);
  Counters* h_ctr = (Counters*)(pin + ob_ids + ob_sc + ((ob_cnt + 63) / 64) * 64);
  for (int s0 = 0; s0 < B; s0 += S) {
    cs.B = std::min(S, B - s0);
    cs.prm = *params;
    cs.timed = stats != nullptr;
    // batch-wide output buffers must survive phase_a's reserve() calls: they only grow, and were
    // reserved above for the full batch, so phase_a's per-slice reserve is a no-op for them.
    NP_TRY(phase_a(ix, &cs, w.q.as<float>(), w.qoff.as<int32_t>() + s0, q_tok_offsets + s0,
                   subset_len > 0 ? w.subset.as<int64_t>() : nullptr, subset_len));
    NP_TRY(phase_b(ix, &cs, w.qoff.as<int32_t>() + s0, nullptr, w.out_ids.as<int64_t>() + (int64_t)s0 * topk,
                   w.out_scores.as<float>() + (int64_t)s0 * topk, nullptr, w.out_counts.as<int32_t>() + s0));
    if (stats) {
      NP_HIP(hipMemcpyAsync(h_ctr, w.ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
      NP_HIP(hipStreamSynchronize(st));
      float ms[8] = {0};
      for (int i = 0; i < 7; ++i) (void)hipEventElapsedTime(&ms[i], cs.ctx->ev[i], cs.ctx->ev[i + 1]);
      acc.ms_centroid += ms[0];
      acc.ms_probe += ms[1];
      acc.ms_candidates += ms[2];
      acc.ms_approx += ms[3];
      acc.ms_select += ms[4];
      acc.ms_exact += ms[5];
      acc.ms_topk += ms[6];
      float tot = 0;
      (void)hipEventElapsedTime(&tot, cs.ctx->ev[0], cs.ctx->ev[7]);
      acc.ms_total += tot;
      if (cs.hot_timed) {
        float hot = 0;
        (void)hipEventElapsedTime(&hot, cs.ctx->ev[8], cs.ctx->ev[9]);
        acc.ms_hot_level += hot;
      }
      acc.n_cells += (int64_t)h_ctr->n_cells;
      acc.n_ivf_ids += (int64_t)h_ctr->n_ivf_ids;
      acc.n_candidates += (int64_t)h_ctr->n_candidates;
      acc.n_cand_tokens += (int64_t)h_ctr->n_cand_tokens;
      acc.n_exact_docs += (int64_t)h_ctr->n_exact_docs;
      acc.n_exact_tokens += (int64_t)h_ctr->n_exact_tokens;
      acc.n_cand_codes += (int64_t)h_ctr->n_cand_codes;
      acc.n_survivors += (int64_t)h_ctr->n_survivors;
      acc.n_cand_dcodes += (int64_t)(h_ctr->n_cand_dcodes ? h_ctr->n_cand_dcodes : h_ctr->n_cand_codes);
      acc.n_level2 += (int64_t)h_ctr->n_level2;
      acc.n_level0 += (int64_t)h_ctr->n_level0;
      acc.n_rounds = std::max(acc.n_rounds, (int32_t)h_ctr->n_rounds);
    }
  }
  if (topk > 0) {
    NP_HIP(hipMemcpyAsync(h_ids, w.out_ids.p, (size_t)B * topk * 8, hipMemcpyDeviceToHost, st));
    NP_HIP(hipMemcpyAsync(h_sc, w.out_scores.p, (size_t)B * topk * 4, hipMemcpyDeviceToHost, st));
  }
  NP_HIP(hipMemcpyAsync(h_cnt, w.out_counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  NP_TRY(end_use(&cs));
  NP_HIP(hipStreamSynchronize(st));
  if (topk > 0) {
    memcpy(out_ids, h_ids, (size_t)B * topk * 8);
    memcpy(out_scores, h_sc, (size_t)B * topk * 4);
  }
  memcpy(out_counts, h_cnt, (size_t)B * 4);
  if (stats) {
    acc.n_queries = B;
    *stats = acc;
  }
  return NP_OK;
}

// ---- document-sharded two-phase call -------------------------------------------------------------------
int np_hip_search_phase_a(const np_index* ix, const float* d_queries, const int32_t* d_q_tok_offsets,
                          const int32_t* h_q_tok_offsets, int32_t B, int32_t dim, const np_search_params* params,
                          const int64_t* d_subset, int64_t subset_len, const uint32_t* d_elig_global,
                          uint64_t* d_sel_keys, void* stream, void** call_state) {
  clear_error();
  if (!call_state) {
    set_error("Search failed: call_state is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *call_state = nullptr;
  NP_TRY(validate(ix, B, dim, params));
  if (!d_queries || !d_q_tok_offsets || !h_q_tok_offsets || !d_sel_keys) {
    set_error("Search failed: NULL buffer");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (subset_len > 0 && !d_subset) {
    set_error("Search failed: subset_len > 0 but subset is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B > slice_size(ix, h_q_tok_offsets, B, params)) {
    set_error("Search failed: a sharded call must fit one workspace slice (B=%d); raise workspace_bytes/max_batch", B);
    return NP_ERR_SEARCH;
  }
  DeviceGuard g(ix->device);
  CallState* cs = new CallState();
  int rc = acquire_context(ix, &cs->ctx);
  if (rc != NP_OK) {
    delete cs;
    return rc;
  }
  cs->B = B;
  cs->prm = *params;
  cs->elig_global = d_elig_global;
  rc = begin_use(cs, stream);
  if (rc == NP_OK) rc = phase_a(ix, cs, d_queries, d_q_tok_offsets, h_q_tok_offsets, d_subset, subset_len);
  if (rc == NP_OK && cs->n_sel > 0 && B > 0) {
    hipError_t e = hipMemcpyAsync(d_sel_keys, cs->ctx->ws->sel_keys.p, (size_t)B * cs->n_sel * 8,
                                  hipMemcpyDeviceToDevice, cs->stream);
    if (e != hipSuccess) {
      set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e));
      rc = NP_ERR_DEVICE_UNAVAILABLE;
    }
  }
  if (rc != NP_OK) {
    (void)end_use(cs);
    release_context(ix, cs->ctx);
    delete cs;
    return rc;
  }
  // phase B needs the offsets again; keep a device copy owned by the workspace
  Workspace& w = *cs->ctx->ws;
  rc = w.qoff.reserve((size_t)(B + 1) * 4);
  if (rc == NP_OK) {
    hipError_t e = hipMemcpyAsync(w.qoff.p, d_q_tok_offsets, (size_t)(B + 1) * 4, hipMemcpyDeviceToDevice, cs->stream);
    if (e != hipSuccess) rc = NP_ERR_DEVICE_UNAVAILABLE;
  }
  if (rc != NP_OK) {
    (void)end_use(cs);
    release_context(ix, cs->ctx);
    delete cs;
    return rc;
  }
  *call_state = cs;
  return NP_OK;
}

int np_hip_search_phase_b(const np_index* ix, void* call_state, const uint64_t* d_cut, int64_t* d_out_ids,
                          float* d_out_scores, uint64_t* d_out_keys, int32_t* d_out_counts, void* stream) {
  clear_error();
  CallState* cs = (CallState*)call_state;
  if (!ix || !cs || !d_out_counts) {
    set_error("Search failed: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  if (stream && (hipStream_t)stream != cs->stream) {
    set_error("Search failed: phase B must use phase A's stream");
    return NP_ERR_INVALID_ARGUMENT;
  }
  return phase_b(ix, cs, cs->ctx->ws->qoff.as<int32_t>(), d_cut, d_out_ids, d_out_scores, d_out_keys, d_out_counts);
}

void np_hip_search_end(const np_index* ix, void* call_state) {
  CallState* cs = (CallState*)call_state;
  if (!ix || !cs) return;
  DeviceGuard g(ix->device);
  (void)end_use(cs);
  release_context(ix, cs->ctx);
  delete cs;
}

int64_t np_hip_elig_words(const np_index* ix) { return ix ? ix->KP / 32 : 0; }

int np_hip_subset_eligible(const np_index* ix, const int64_t* d_subset, int64_t subset_len, uint32_t* d_elig_bits,
                           void* stream) {
  clear_error();
  if (!ix || !d_elig_bits || (subset_len > 0 && !d_subset)) {
    set_error("subset_eligible: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  hipStream_t st = (hipStream_t)stream;
  NP_HIP(hipMemsetAsync(d_elig_bits, 0, (size_t)(ix->KP / 32) * 4, st));
  if (subset_len > 0)
    subset_kernel<<<(unsigned)((subset_len + 3) / 4), 256, 0, st>>>(d_subset, subset_len, ix->doc_begin, ix->n_docs,
                                                                     ix->d_doc_offsets, ix->codes(), nullptr, d_elig_bits);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

int np_hip_or_bitmaps(const np_index* ix, const uint32_t* d_all, int32_t G, int64_t words, uint32_t* d_out, void* stream) {
  clear_error();
  if (!ix || !d_all || !d_out || G < 1 || words < 0) {
    set_error("or_bitmaps: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (words == 0) return NP_OK;
  DeviceGuard g(ix->device);
  or_reduce_kernel<<<(unsigned)((words + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_all, G, words, d_out);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

int np_hip_select_cut(const np_index* ix, const uint64_t* d_all_keys, int32_t G, int32_t B, int32_t n_sel,
                      uint64_t* d_cut, void* stream) {
  clear_error();
  return np::select_cut_strided(ix, d_all_keys, (int64_t)B * n_sel, -1, G, B, n_sel, d_cut, (hipStream_t)stream);
}

int np_hip_merge_topk(const np_index* ix, const int64_t* d_ids, const float* d_scores, const uint64_t* d_keys,
                      const int32_t* d_counts, int32_t G, int32_t B, int32_t top_k, int64_t* d_out_ids,
                      float* d_out_scores, int32_t* d_out_counts, void* stream) {
  clear_error();
  if (!ix || !d_ids || !d_scores || !d_keys || !d_counts || !d_out_counts || G < 1 || B < 0 || top_k < 0) {
    set_error("merge_topk: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return NP_OK;
  DeviceGuard g(ix->device);
  const int64_t rs = (int64_t)B * top_k;
  merge_topk_kernel<<<B, 256, 0, (hipStream_t)stream>>>(d_ids, d_scores, d_keys, d_counts, rs, rs, rs, B, G, B, top_k,
                                                        d_out_ids, d_out_scores, d_out_counts);
  NP_HIP(hipGetLastError());
  return NP_OK;
}

// Same merge over one PACKED record per rank (what np_hip_search_batch_sharded all-gathers): rank g's record starts at
// d_records + g * record_bytes and holds ids at 0, keys at off_keys, scores at off_scores, counts at off_counts.
int np_hip_merge_packed(const np_index* ix, const void* d_records, int64_t record_bytes, int64_t off_keys,
                        int64_t off_scores, int64_t off_counts, int32_t G, int32_t B, int32_t top_k, int64_t* d_out_ids,
                        float* d_out_scores, int32_t* d_out_counts, void* stream) {
  clear_error();
  return np::merge_packed_status(ix, d_records, record_bytes, off_keys, off_scores, off_counts, -1, nullptr, G, B, top_k,
                                 d_out_ids, d_out_scores, d_out_counts, (hipStream_t)stream);
}

// ---- N2: decompress_documents (index.rs:1197-1245) ---------------------------------------------------------
int np_hip_decompress_documents(const np_index* ix, const int64_t* doc_ids, int64_t n_docs, float* out_embeddings,
                                int64_t out_capacity_rows, int64_t* out_lengths) {
  clear_error();
  if (!ix || (n_docs > 0 && (!doc_ids || !out_lengths)) || n_docs < 0) {
    set_error("decompress_documents: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  std::vector<int64_t> off((size_t)ix->n_docs + 1);
  NP_HIP(hipMemcpy(off.data(), ix->d_doc_offsets, off.size() * 8, hipMemcpyDeviceToHost));
  std::vector<int64_t> toks, bases;
  for (int64_t i = 0; i < n_docs; ++i) {
    const int64_t d = doc_ids[i] - ix->doc_begin;
    if (d < 0 || d >= ix->n_docs) {  // index.rs:1202-1204: out-of-range ids contribute length 0
      out_lengths[i] = 0;
      continue;
    }
    out_lengths[i] = off[d + 1] - off[d];
    if (out_embeddings) {
      const int64_t base = (int64_t)toks.size();
      for (int64_t t = off[d]; t < off[d + 1]; ++t) {
        toks.push_back(t);
        bases.push_back(base);
      }
    }
  }
  if (!out_embeddings || toks.empty()) return NP_OK;
  if ((int64_t)toks.size() > out_capacity_rows) {
    set_error("decompress_documents: output holds %lld rows, %zu needed", (long long)out_capacity_rows, toks.size());
    return NP_ERR_INVALID_ARGUMENT;
  }
  int64_t* d_tok = nullptr;
  float* d_out = nullptr;
  NP_HIP(hipMalloc(&d_tok, toks.size() * 16));
  hipError_t e = hipMalloc(&d_out, toks.size() * (size_t)ix->ldim * 4);
  if (e != hipSuccess) {
    (void)hipFree(d_tok);
    set_error("hipMalloc failed: %s", hipGetErrorString(e));
    return NP_ERR_OUT_OF_MEMORY;
  }
  e = hipMemcpy(d_tok, toks.data(), toks.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_tok + toks.size(), bases.data(), toks.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    decompress_kernel<<<(unsigned)((toks.size() + 3) / 4), 256>>>(d_tok, d_tok + toks.size(),
                                                                  ix->tok_sorted ? ix->d_tok_pos : nullptr,
                                                                  (int64_t)toks.size(), ix->dim, ix->ldim, ix->nbits, ix->pd,
                                                                  ix->d_centroids, ix->d_wlut, ix->codes(), ix->d_residuals,
                                                                  d_out);
    e = hipMemcpy(out_embeddings, d_out, toks.size() * (size_t)ix->ldim * 4, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_tok);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    set_error("decompress_documents failed: %s", hipGetErrorString(e));
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  return NP_OK;
}

// ---- stage-level trace of one query (parity tests) -----------------------------------------------------------
int np_hip_debug_trace(const np_index* ix, const float* query, int32_t n_tokens, int32_t dim,
                       const np_search_params* params, const int64_t* subset, int64_t subset_len, int64_t* cells,
                       int64_t cap_cells, int64_t* n_cells, int64_t* cand, float* approx, int64_t cap_cand,
                       int64_t* n_cand, int64_t* sel, float* sel_exact, int64_t cap_sel, int64_t* n_sel) {
  clear_error();
  NP_TRY(validate(ix, 1, dim, params));
  if (!query || n_tokens < 0) {
    set_error("debug_trace: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  CallState cs;
  NP_TRY(acquire_context(ix, &cs.ctx));
  UseGuard guard{ix, &cs};
  Workspace& w = *cs.ctx->ws;
  NP_TRY(begin_use(&cs, nullptr));
  guard.began = true;
  hipStream_t st = cs.stream;
  int32_t h_off[2] = {0, n_tokens};
  NP_TRY(w.q.reserve((size_t)std::max(n_tokens, 1) * dim * 4));
  NP_TRY(w.qoff.reserve(8));
  if (subset_len > 0) NP_TRY(w.subset.reserve((size_t)subset_len * 8));
  if (n_tokens > 0) NP_HIP(hipMemcpyAsync(w.q.p, query, (size_t)n_tokens * dim * 4, hipMemcpyHostToDevice, st));
  NP_HIP(hipMemcpyAsync(w.qoff.p, h_off, 8, hipMemcpyHostToDevice, st));
  if (subset_len > 0) NP_HIP(hipMemcpyAsync(w.subset.p, subset, (size_t)subset_len * 8, hipMemcpyHostToDevice, st));
  cs.B = 1;
  cs.prm = *params;
  cs.trace = true;
  NP_TRY(phase_a(ix, &cs, w.q.as<float>(), w.qoff.as<int32_t>(), h_off, subset_len > 0 ? w.subset.as<int64_t>() : nullptr,
                 subset_len));
  NP_TRY(phase_b(ix, &cs, w.qoff.as<int32_t>(), nullptr, w.out_ids.as<int64_t>(), w.out_scores.as<float>(),
                 w.out_keys.as<uint64_t>(), w.out_counts.as<int32_t>()));
  NP_TRY(end_use(&cs));
  NP_HIP(hipStreamSynchronize(st));
  int32_t nc = 0, nd = 0, ns = 0;
  NP_HIP(hipMemcpy(&nc, w.n_cells.p, 4, hipMemcpyDeviceToHost));
  NP_HIP(hipMemcpy(&nd, w.n_cand.p, 4, hipMemcpyDeviceToHost));
  NP_HIP(hipMemcpy(&ns, w.nsel.p, 4, hipMemcpyDeviceToHost));
  if (n_cells) *n_cells = nc;
  if (n_cand) *n_cand = nd;
  if (n_sel) *n_sel = ns;
  if (cells && nc > 0) {
    std::vector<uint32_t> t((size_t)nc);
    NP_HIP(hipMemcpy(t.data(), w.cells.p, (size_t)nc * 4, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    for (int64_t i = 0; i < std::min<int64_t>(nc, cap_cells); ++i) cells[i] = t[(size_t)i];
  }
  if (nd > 0 && (cand || approx)) {
    const int64_t m = std::min<int64_t>(nd, cap_cand);
    if (cand) {
      std::vector<uint32_t> t((size_t)m);
      NP_HIP(hipMemcpy(t.data(), w.cand.p, (size_t)m * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; ++i) cand[i] = (int64_t)t[(size_t)i] + ix->doc_begin;
    }
    if (approx) NP_HIP(hipMemcpy(approx, w.approx.p, (size_t)m * 4, hipMemcpyDeviceToHost));
  }
  if (ns > 0 && (sel || sel_exact)) {
    const int64_t m = std::min<int64_t>(ns, cap_sel);
    if (sel) {
      std::vector<uint32_t> t((size_t)m);
      NP_HIP(hipMemcpy(t.data(), w.sel_doc.p, (size_t)m * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; ++i) sel[i] = (int64_t)t[(size_t)i] + ix->doc_begin;
    }
    if (sel_exact) NP_HIP(hipMemcpy(sel_exact, w.exact.p, (size_t)m * 4, hipMemcpyDeviceToHost));
  }
  return NP_OK;
}

// ---- N3: index-time encode (codec.rs:297-411; index.rs:289-371 encode_index_chunk) -----------------------------
int np_hip_encode_tokens(const np_index* ix, const float* embeddings, int64_t n_tokens, int32_t dim,
                         const float* bucket_cutoffs, int64_t* out_codes, uint8_t* out_packed) {
  clear_error();
  if (!ix || n_tokens < 0 || (n_tokens > 0 && (!embeddings || !out_codes || !out_packed)) || !bucket_cutoffs) {
    set_error("encode_tokens: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (dim != ix->ldim) {
    set_error("Shape error: embedding dim %d does not match index dim %d", dim, ix->ldim);
    return NP_ERR_SHAPE;
  }
  if (!dim_supported(ix->dim, ix->nbits)) {
    set_error("Shape error: the HIP path supports dim <= 128; index has dim=%d nbits=%d", ix->ldim, ix->lnbits);
    return NP_ERR_SHAPE;
  }
  if (ix->K <= 0) {
    set_error("Codec error: the index has no centroids");
    return NP_ERR_CODEC;
  }
  if (n_tokens == 0) return NP_OK;
  DeviceGuard g(ix->device);
  CallState cs;
  NP_TRY(acquire_context(ix, &cs.ctx));
  UseGuard guard{ix, &cs};
  Workspace& w = *cs.ctx->ws;
  NP_TRY(begin_use(&cs, nullptr));
  guard.began = true;
  hipStream_t st = cs.stream;
  const int LQP = 32;
  const int64_t KP = ix->KP, G = KP >> 5;
  int64_t S = ix->ws_budget.load(std::memory_order_relaxed) / std::max<int64_t>(per_query_bytes(ix, LQP, 1, 1), 1);
  S = std::max<int64_t>(1, std::min<int64_t>(S, std::min<int64_t>(ix->opts.max_batch, NP_S4_MAXB)));
  S = std::min<int64_t>(S, (n_tokens + LQP - 1) / LQP);
  const int64_t TB = S * LQP;
  const int pd = ix->lpd, ncut = (1 << ix->lnbits) - 1;   // the OUTPUT is in file geometry (codec.rs:356-411)
  const int sdim = ix->dim;                                // the GEMM runs on storage rows
  NP_TRY(w.q.reserve((size_t)TB * dim * 4));
  if (sdim != dim) NP_TRY(w.qpad.reserve((size_t)TB * sdim * 4));
  NP_TRY(w.qoff.reserve((size_t)(S + 1) * 4));
  NP_TRY(w.Qt.reserve((size_t)S * sdim * LQP * 4));
  NP_TRY(w.Qb.reserve((size_t)S * sdim * LQP * 2));
  NP_TRY(w.Qbl.reserve((size_t)S * sdim * LQP * 2));
  NP_TRY(w.QCT.reserve((size_t)S * KP * LQP * 4));
  NP_TRY(w.gmax.reserve((size_t)S * G * LQP * 4));
  NP_TRY(w.out_ids.reserve((size_t)TB * 8));
  NP_TRY(w.cand.reserve((size_t)TB * pd));
  NP_TRY(w.misc.reserve(std::max<size_t>(64, (size_t)ncut * 4)));
  NP_HIP(hipMemcpyAsync(w.misc.p, bucket_cutoffs, (size_t)ncut * 4, hipMemcpyHostToDevice, st));
  std::vector<int32_t> h_off((size_t)S + 1);
  for (int64_t t0 = 0; t0 < n_tokens; t0 += TB) {
    const int64_t nb = std::min<int64_t>(TB, n_tokens - t0);
    const int Sb = (int)((nb + LQP - 1) / LQP);
    for (int s = 0; s <= Sb; ++s) h_off[(size_t)s] = (int32_t)std::min<int64_t>((int64_t)s * LQP, nb);
    NP_HIP(hipMemcpyAsync(w.q.p, embeddings + t0 * dim, (size_t)nb * dim * 4, hipMemcpyHostToDevice, st));
    NP_HIP(hipMemcpyAsync(w.qoff.p, h_off.data(), (size_t)(Sb + 1) * 4, hipMemcpyHostToDevice, st));
    const float* xs = w.q.as<float>();
    if (sdim != dim) {
      pad_rows_kernel<<<(unsigned)((nb * sdim + 255) / 256), 256, 0, st>>>(w.q.as<float>(), nb, dim, sdim, w.qpad.as<float>());
      xs = w.qpad.as<float>();
    }
    prep_queries_kernel<<<Sb, 256, 0, st>>>(xs, w.qoff.as<int32_t>(), sdim, LQP, w.Qt.as<float>(),
                                            w.Qb.as<__bf16>(), w.Qbl.as<__bf16>(), 0.f, nullptr, nullptr);
    switch (ix->dim) {
      case 32: launch_gemm<32>(st, ix, w.Qt.as<float>(), Sb, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>()); break;
      case 64: launch_gemm<64>(st, ix, w.Qt.as<float>(), Sb, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>()); break;
      case 96: launch_gemm<96>(st, ix, w.Qt.as<float>(), Sb, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>()); break;
      default: launch_gemm<128>(st, ix, w.Qt.as<float>(), Sb, LQP, w.QCT.as<float>(), w.gmax.as<uint32_t>()); break;
    }
    encode_argmax_kernel<<<(unsigned)((nb + 3) / 4), 256, 0, st>>>(w.QCT.as<float>(), w.gmax.as<uint32_t>(), ix->K, KP,
                                                                  LQP, nb, w.out_ids.as<int64_t>());
    encode_pack_kernel<<<(unsigned)((nb * pd + 255) / 256), 256, 0, st>>>(w.q.as<float>(), ix->d_centroids,
                                                                         w.out_ids.as<int64_t>(), w.misc.as<float>(), nb,
                                                                         dim, sdim, ix->lnbits, w.cand.as<uint8_t>());
    NP_HIP(hipGetLastError());
    NP_HIP(hipMemcpyAsync(out_codes + t0, w.out_ids.p, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    NP_HIP(hipMemcpyAsync(out_packed + t0 * pd, w.cand.p, (size_t)nb * pd, hipMemcpyDeviceToHost, st));
    NP_HIP(hipStreamSynchronize(st));   // h_off / the workspace are reused by the next slice
  }
  NP_TRY(end_use(&cs));
  return NP_OK;
}

// ---- N4: /rerank MaxSim on caller-supplied embeddings (next-plaid-api handlers/rerank.rs:57-94,139-170) ---------
int np_hip_rerank_maxsim(int32_t device, const float* query, int32_t n_query_tokens, int32_t dim,
                         const float* doc_embeddings, const int64_t* doc_tok_offsets, int64_t n_docs, float* out_scores,
                         int64_t* out_order) {
  clear_error();
  if (n_docs <= 0) {  // rerank.rs:113-115
    set_error("No documents provided");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (!query || !doc_tok_offsets || !out_scores || n_query_tokens < 0 || dim <= 0) {
    set_error("rerank_maxsim: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (dim > 1024) {
    set_error("Shape error: rerank supports dim <= 1024, got %d", dim);
    return NP_ERR_SHAPE;
  }
  for (int64_t i = 0; i < n_docs; ++i)
    if (doc_tok_offsets[i + 1] < doc_tok_offsets[i] || doc_tok_offsets[0] != 0) {
      set_error("rerank_maxsim: doc_tok_offsets must start at 0 and be non-decreasing");
      return NP_ERR_INVALID_ARGUMENT;
    }
  const int64_t T = doc_tok_offsets[n_docs];
  if (T > 0 && !doc_embeddings) {
    set_error("rerank_maxsim: invalid argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    set_error("no usable gfx950 device %d", device);
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  DeviceGuard g(device);
  float *d_q = nullptr, *d_d = nullptr, *d_s = nullptr;
  int64_t* d_off = nullptr;
  int* d_f = nullptr;
  struct Free {
    float **a, **b, **c;
    int64_t** d;
    int** e;
    ~Free() {
      (void)hipFree(*a);
      (void)hipFree(*b);
      (void)hipFree(*c);
      (void)hipFree(*d);
      (void)hipFree(*e);
    }
  } fr{&d_q, &d_d, &d_s, &d_off, &d_f};
  NP_HIP(hipMalloc(&d_q, std::max<size_t>((size_t)n_query_tokens * dim * 4, 4)));
  NP_HIP(hipMalloc(&d_d, std::max<size_t>((size_t)T * dim * 4, 4)));
  NP_HIP(hipMalloc(&d_s, (size_t)n_docs * 4));
  NP_HIP(hipMalloc(&d_off, (size_t)(n_docs + 1) * 8));
  NP_HIP(hipMalloc(&d_f, (size_t)n_docs * 4));
  if (n_query_tokens > 0) NP_HIP(hipMemcpy(d_q, query, (size_t)n_query_tokens * dim * 4, hipMemcpyHostToDevice));
  if (T > 0) NP_HIP(hipMemcpy(d_d, doc_embeddings, (size_t)T * dim * 4, hipMemcpyHostToDevice));
  NP_HIP(hipMemcpy(d_off, doc_tok_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice));
  rerank_kernel<<<(unsigned)n_docs, 256, (size_t)8 * dim * 4>>>(d_q, n_query_tokens, dim, d_d, d_off, d_s, d_f);
  NP_HIP(hipGetLastError());
  std::vector<int> flags((size_t)n_docs);
  NP_HIP(hipMemcpy(out_scores, d_s, (size_t)n_docs * 4, hipMemcpyDeviceToHost));
  NP_HIP(hipMemcpy(flags.data(), d_f, (size_t)n_docs * 4, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n_docs; ++i)
    if (flags[(size_t)i]) {  // rerank.rs:71-75,85-89
      set_error("Rerank score contains non-finite value");
      return NP_ERR_INVALID_ARGUMENT;
    }
  if (out_order) {  // rerank.rs:162-163: stable sort by score_desc_cmp (all scores are finite here: b.total_cmp(a))
    auto key = [](float x) {
      uint32_t b;
      memcpy(&b, &x, 4);
      return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    };
    std::vector<int64_t> ord((size_t)n_docs);
    for (int64_t i = 0; i < n_docs; ++i) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(),
                     [&](int64_t a, int64_t b) { return key(out_scores[a]) > key(out_scores[b]); });
    memcpy(out_order, ord.data(), (size_t)n_docs * 8);
  }
  return NP_OK;
}

}  // extern "C"
