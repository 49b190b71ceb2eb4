// np_index.hip -- makes a next-plaid index (or one document shard of it) resident in HBM.
//
// Replaces the in-memory half of MmapIndex::load (next-plaid/src/index.rs:1089-1139): ivf_offsets and
// doc_offsets prefix sums, the codec tables (codec.rs:154-225), and -- instead of mmap'ing
// merged_codes/merged_residuals -- device arrays laid out for the search kernels (np_internal.h).
// Also: the seeded synthetic corpus generator (bench / scale tests) and export back to host.
#include "np_internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <thread>
#include <math.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

namespace np {

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return NP_OK;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  size_t want = bytes + (bytes >> 3) + 256;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    return NP_ERR_OUT_OF_MEMORY;
  }
  cap = want;
  return NP_OK;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

void shard_range(int64_t n_total, int rank, int count, int64_t* b, int64_t* e) {
  if (count <= 1) {
    *b = 0;
    *e = n_total;
    return;
  }
  *b = (int64_t)((__int128)n_total * rank / count);
  *e = (int64_t)((__int128)n_total * (rank + 1) / count);
}

int normalise_opts(const np_open_opts* in, np_open_opts* o) {
  memset(o, 0, sizeof *o);
  if (in) *o = *in;
  if (o->shard_count <= 0) o->shard_count = 1;
  if (o->shard_rank < 0 || o->shard_rank >= o->shard_count) {
    set_error("Invalid configuration: shard_rank %d out of range for shard_count %d", o->shard_rank, o->shard_count);
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (o->n_contexts <= 0) o->n_contexts = 2;
  if (o->n_contexts > 16) o->n_contexts = 16;
  if (o->max_batch <= 0) o->max_batch = 64;
  if (o->max_query_tokens <= 0) o->max_query_tokens = 64;
  // workspace_bytes <= 0: chosen when the index is resident (default_workspace: up to 16 GiB per context of what is free)
  return NP_OK;
}

// One clamp table for the environment (read once at open) and np_hip_index_tune().  Returns false for an unknown name.
bool set_tuning(Tuning* t, const std::string& n, int value) {
  auto clamp = [](int v, int lo, int hi) { return std::min(std::max(v, lo), hi); };
  if (n == "s4_mode") t->s4_mode = clamp(value, 0, 8);
  else if (n == "s4_minb") t->s4_minb = std::max(value, 1);
  else if (n == "s4_nbx") t->s4_nbx = clamp(value, 8, 512);
  else if (n == "s4_swz") t->s4_swz = value != 0;
  else if (n == "s4_filter") t->s4_filter = value != 0;
  else if (n == "s4_hot") t->s4_hot = clamp(value, 0, 500);
  else if (n == "s4_planes") t->s4_planes = value != 0;
  else if (n == "s4_pexp") t->s4_pexp = clamp(value, 5, 40);
  else if (n == "s4_lpd") t->s4_lpd = value == 2 ? 2 : 4;
  else if (n == "s4_qm") t->s4_qm = value != 0;
  else if (n == "s4_pnbx") t->s4_pnbx = clamp(value, 8, 512);
  else if (n == "s3_slices") t->s3_slices = value != 0;
  else if (n == "s4_warm") t->s4_warm = clamp(value, 0, 1000);
  else if (n == "ub_ncut") t->ub_ncut = clamp(value, 1, 512);
  else if (n == "s3_bisect") t->s3_bisect = clamp(value, 0, 1);
  else if (n == "s3_gain") t->s3_gain = clamp(value, 0, 2);
  else if (n == "s3_gain_mult") t->s3_gain_mult = clamp(value, 1, 16);
  else if (n == "s3_gain_direct") t->s3_gain_direct = clamp(value, 0, 64);
  else if (n == "s4_hot_auto") t->s4_hot_auto = clamp(value, 0, 0x7fffffff);
  else if (n == "ub_nt") t->ub_nt = value < 0 || value > 2 ? 0 : value;
  else if (n == "ub_steal") t->ub_steal = value < 1 ? 1 : value;
  else if (n == "ub_nbx") t->ub_nbx = clamp(value, 8, 256);
  else if (n == "ub_direct") t->ub_direct = clamp(value, 0, 16);
#ifdef NP_DIAGNOSTICS   // phase-skipping timing probe of the hot kernel: results are INVALID when != 0, so a production build has no way to set it
  else if (n == "s4_probe") t->s4_probe = clamp(value, 0, 7);
#endif
  else if (n == "ub_static") t->ub_static = value != 0;
  else if (n == "hot_static") t->hot_static = value != 0;
  else if (n == "s6_xcd") t->s6_xcd = value != 0;
  else if (n == "s6_tiles") t->s6_tiles = value != 0;
  else if (n == "s6_lds") t->s6_lds = clamp(value, 0, 2);
  else if (n == "gemm_cpw") t->gemm_cpw = value == 2 ? 2 : 1;
  else if (n == "s1_split") t->s1_split = value != 0;
  else if (n == "exact_rowmax") t->exact_rowmax = value != 0;
  else return false;
  return true;
}

void read_tuning_env(Tuning* t) {
  static const char* const knobs[][2] = {
      {"NP_S4_MODE", "s4_mode"}, {"NP_S4_MINB", "s4_minb"}, {"NP_S4_NBX", "s4_nbx"}, {"NP_S4_SWZ", "s4_swz"},
      {"NP_S4_FILTER", "s4_filter"}, {"NP_S4_HOT", "s4_hot"}, {"NP_S4_PLANES", "s4_planes"}, {"NP_S4_PEXP", "s4_pexp"}, {"NP_S4_LPD", "s4_lpd"}, {"NP_S4_QM", "s4_qm"}, {"NP_S4_PNBX", "s4_pnbx"}, {"NP_S4_WARM", "s4_warm"}, {"NP_UB_NCUT", "ub_ncut"}, {"NP_S3_BISECT", "s3_bisect"}, {"NP_S3_GAIN", "s3_gain"}, {"NP_S3_GAIN_MULT", "s3_gain_mult"}, {"NP_S3_GAIN_DIRECT", "s3_gain_direct"}, {"NP_S4_HOT_AUTO", "s4_hot_auto"}, {"NP_S3_SLICES", "s3_slices"}, {"NP_UB_NT", "ub_nt"},
      {"NP_UB_STEAL", "ub_steal"}, {"NP_UB_NBX", "ub_nbx"}, {"NP_UB_DIRECT", "ub_direct"}, {"NP_UB_STATIC", "ub_static"}, {"NP_HOT_STATIC", "hot_static"}, {"NP_S6_XCD", "s6_xcd"}, {"NP_S6_TILES", "s6_tiles"}, {"NP_S6_LDS", "s6_lds"}, {"NP_GEMM_CPW", "gemm_cpw"}, {"NP_S1_SPLIT", "s1_split"},
      {"NP_EXACT_ROWMAX", "exact_rowmax"}};
  for (const auto& k : knobs) {
    const char* e = getenv(k[0]);
    if (e && *e) (void)set_tuning(t, k[1], atoi(e));
  }
#ifdef NP_DIAGNOSTICS
  if (const char* e = getenv("NP_S4_PROBE")) (void)set_tuning(t, "s4_probe", atoi(e));
#endif
}

// Default per-context scratch budget: what the device has free once the index is resident, shared by the contexts, between
// 2 and 16 GiB.  The candidate pool is sized by it; the host enqueues the worst-case number of pool rounds of a batch
// (B x n_docs entries / pool) and the empty ones cost ~45 us of launches each: 16 GiB is 3 rounds at 10 M documents where
// 8 GiB was 6.  A 12.5 M x 300-token shard (268 GB) leaves 12 GiB per context with three contexts.
static void default_workspace(DeviceIndex* ix) {
  ix->ws_auto = ix->opts.workspace_bytes <= 0;
  ix->ws_budget = ix->opts.workspace_bytes;
  ix->ws_budget_open = ix->opts.workspace_bytes;
  if (!ix->ws_auto) return;
  size_t free_b = 0, total_b = 0;
  int64_t ws = (int64_t)8 << 30;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const int64_t share = ((int64_t)free_b - ((int64_t)4 << 30)) / std::max(ix->opts.n_contexts, 1);
    ws = std::min<int64_t>((int64_t)16 << 30, std::max<int64_t>((int64_t)2 << 30, share));
  }
  ix->opts.workspace_bytes = ws;
  ix->ws_budget = ws;
  ix->ws_budget_open = ws;
}

static int check_device(int dev) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available");
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  if (dev < 0 || dev >= n) {
    set_error("device %d out of range (%d devices)", dev, n);
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  return NP_OK;
}

template <class T>
static int dev_alloc(T** p, size_t n, size_t* acct) {
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) {
    *p = nullptr;
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return NP_ERR_OUT_OF_MEMORY;
  }
  if (acct) *acct += bytes;
  return NP_OK;
}

void destroy_device_index(DeviceIndex* ix) {
  if (!ix) return;
  DeviceGuard g(ix->device);
  for (Context* c : ix->contexts) destroy_context(c);
  ix->contexts.clear();
  (void)hipFree(ix->d_centroids);
  (void)hipFree(ix->d_wlut);
  (void)hipFree(ix->d_codes);
  (void)hipFree(ix->d_ucodes);
  (void)hipFree(ix->d_ulen);
  (void)hipFree(ix->d_useg);
  (void)hipFree(ix->d_doc_meta);
  (void)hipFree(ix->d_inv_norm);
  (void)hipFree(ix->d_tok_pos);
  (void)hipFree(ix->d_residuals);
  (void)hipFree(ix->d_doc_offsets);
  (void)hipFree(ix->d_ivf);
  (void)hipFree(ix->d_ivf_offsets);
  (void)hipFree(ix->d_ivf_split);
}

static uint32_t bitrev(uint32_t v, int nbits) {
  uint32_t r = 0;
  for (int k = 0; k < nbits; ++k)
    if (v & (1u << k)) r |= 1u << (nbits - 1 - k);
  return r;
}

static int check_geometry(int64_t K, int dim, int nbits, int64_t n_total) {
  if (nbits <= 0 || 8 % nbits != 0) {  // codec.rs:161-166
    set_error("Codec error: nbits must be a divisor of 8, got %d", nbits);
    return NP_ERR_CODEC;
  }
  if (K <= 0 || dim <= 0) {
    set_error("Shape error: centroids must be [K > 0, dim > 0], got [%lld, %d]", (long long)K, dim);
    return NP_ERR_SHAPE;
  }
  if ((dim * nbits) % 8 != 0) {
    set_error("Shape error: dim*nbits must be a multiple of 8 (dim=%d nbits=%d)", dim, nbits);
    return NP_ERR_SHAPE;
  }
  if (K >= ((int64_t)1 << 31) || n_total >= ((int64_t)1 << 32) - 1) {
    set_error("Index load failed: K=%lld / N=%lld exceed the 32-bit device id space", (long long)K, (long long)n_total);
    return NP_ERR_INDEX_LOAD;
  }
  return NP_OK;
}

// largest squared row norm of the centroid table (bit pattern of a non-negative float orders like the float) and a
// non-finite flag: |Q.c| <= ||q|| * cmax scales the u8 table of the S4 upper-bound filter (np_kernels.h)
__global__ void __launch_bounds__(256) centroid_bound_kernel(const float* __restrict__ C, int64_t K, int dim,
                                                             uint32_t* __restrict__ out /* [0] max bits, [1] non-finite */) {
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= K) return;
  float ss = 0.f;
  for (int j = lane; j < dim; j += 64) {
    const float x = C[c * dim + j];
    ss = fmaf(x, x, ss);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if (lane == 0) {
    if ((__float_as_uint(ss) & 0x7F800000u) == 0x7F800000u) atomicOr(&out[1], 1u);
    else atomicMax(&out[0], __float_as_uint(ss));
  }
}

static int upload_codec(DeviceIndex* ix, const float* centroids, const float* bucket_weights) {
  NP_TRY(dev_alloc(&ix->d_centroids, (size_t)ix->K * ix->dim, &ix->device_bytes));
  if (ix->ldim == ix->dim) {
    NP_HIP(hipMemcpy(ix->d_centroids, centroids, (size_t)ix->K * ix->dim * sizeof(float), hipMemcpyHostToDevice));
  } else {   // rows zero-padded to the storage width (np_internal.h storage_dim)
    NP_HIP(hipMemset(ix->d_centroids, 0, (size_t)ix->K * ix->dim * sizeof(float)));
    NP_HIP(hipMemcpy2D(ix->d_centroids, (size_t)ix->dim * sizeof(float), centroids, (size_t)ix->ldim * sizeof(float),
                       (size_t)ix->ldim * sizeof(float), (size_t)ix->K, hipMemcpyHostToDevice));
  }
  {
    uint32_t* d_b = nullptr;
    uint32_t h_b[2] = {0, 0};
    NP_HIP(hipMalloc(&d_b, 8));
    NP_HIP(hipMemset(d_b, 0, 8));
    centroid_bound_kernel<<<(unsigned)((ix->K + 3) / 4), 256>>>(ix->d_centroids, ix->K, ix->dim, d_b);
    hipError_t e = hipMemcpy(h_b, d_b, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_b);
    NP_HIP(e);
    float ss;
    memcpy(&ss, &h_b[0], 4);
    ix->cmax = sqrtf(ss) * 1.0001f;   // the f32 sum of squares is within ~dim * 2^-24 of the real one
    ix->filter_ok = h_b[1] == 0;
  }
  const int nb = 1 << ix->nbits, lnb = 1 << ix->lnbits;
  std::vector<float> wl(nb);
  for (int s = 0; s < nb; ++s) {   // 1-bit files stored as 2-bit: buckets 2 and 3 never occur
    const uint32_t b = bitrev((uint32_t)s, ix->nbits);
    wl[s] = b < (uint32_t)lnb ? bucket_weights[b] : 0.f;
  }
  bool wok = true;
  for (int s = 0; s < nb; ++s) wok = wok && std::isfinite(wl[s]) && std::fabs(wl[s]) < 1e6f;
  ix->s6_fast_ok = ix->filter_ok && ix->cmax < 1e6f && wok;
  ix->pad_ss = ix->dim > ix->ldim ? (float)(ix->dim - ix->ldim) * (wl[0] * wl[0]) : 0.f;
  NP_TRY(dev_alloc(&ix->d_wlut, nb, &ix->device_bytes));
  NP_HIP(hipMemcpy(ix->d_wlut, wl.data(), nb * sizeof(float), hipMemcpyHostToDevice));
  return NP_OK;
}


// ---- derived: per-document distinct codes (one block per document, bitonic sort in LDS) ------------------
// Two passes over the same kernel: with ucodes == NULL it only counts (ulen[d]); the host then lays the lists out DENSELY
// (exclusive scan of the lengths, each list start rounded up to NP_ULIST_ALIGN entries so that list reads of 8 bytes stay
// aligned) and the second pass writes list d at uoff[d].  (Round 2 kept each list at its document's token offset: T
// entries of 4 bytes for ~T/4 useful ones, 4 B per token of HBM.)
#define NP_UNIQ_MAX 4096
#define NP_ULIST_ALIGN 4
__global__ void __launch_bounds__(256) unique_codes_kernel(const int64_t* __restrict__ doc_off, CodeArr codes,
                                                           void* __restrict__ ucodes, const int64_t* __restrict__ uoff,
                                                           int32_t* __restrict__ ulen) {
  __shared__ uint32_t s[NP_UNIQ_MAX];
  __shared__ int s_wave[4];
  const int64_t d = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t off = doc_off[d];
  const int len = (int)(doc_off[d + 1] - off);
  const int64_t uo = ucodes ? uoff[d] : 0;
  auto put = [&](int i, uint32_t c) {
    if (codes.wide) static_cast<uint32_t*>(ucodes)[uo + i] = c;
    else static_cast<uint16_t*>(ucodes)[uo + i] = (uint16_t)c;
  };
  if (len > NP_UNIQ_MAX) {  // very long document: keep the full list (still correct, just not shorter)
    if (ucodes)
      for (int i = tid; i < len; i += 256) put(i, codes[off + i]);
    else if (tid == 0) ulen[d] = len;
    return;
  }
  int n = 1;
  while (n < len) n <<= 1;
  for (int i = tid; i < n; i += 256) s[i] = (i < len) ? codes[off + i] : 0xFFFFFFFFu;
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < n; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t a = s[i], c = s[ixj];
          const bool asc = (i & k) == 0;
          if (asc ? (a > c) : (a < c)) { s[i] = c; s[ixj] = a; }
        }
      }
    }
  __syncthreads();
  int base = 0;
  for (int i0 = 0; i0 < len; i0 += 256) {
    const int i = i0 + tid;
    const bool head = i < len && (i == 0 || s[i] != s[i - 1]);
    const unsigned long long bal = __ballot(head);
    if (lane == 0) s_wave[wave] = (int)__popcll(bal);
    __syncthreads();
    int before = base;
    for (int k = 0; k < wave; ++k) before += s_wave[k];
    if (head && ucodes) put(before + (int)__popcll(bal & ((1ull << lane) - 1ull)), s[i]);
    base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (tid == 0 && !ucodes) ulen[d] = base;
}

// ---- list layout (round 3, second step): one fixed-stride BLOCK per document + an overflow region ---------------------
// d_ucodes = [ n_docs blocks of S entries | overflow lists ].  Block d = a 16-byte header {u32 #distinct codes, u32
// document length, u32 overflow index, u32 0} followed by the document's sorted distinct codes when they fit the block
// (#distinct <= S - HDR entries); a longer list lives in the overflow region (4-entry aligned, at overflow index * 4) and
// the block carries only the header.  S is chosen at open: a multiple of 64 bytes that holds 99 % of the lists (at most 256
// bytes: one 8-byte load per lane of half a wave).  With the block address computable from the document id, the hot level of the
// S4 filter needs NO per-candidate record: S3 hands it bare document ids and the 16-byte record gather of compact_kernel
// (one 128-byte line per candidate at 1.9 % density, 0.32 ms per batch at 10 M documents) disappears; the header arrives
// with the list.  Every other consumer keeps addressing lists through the 40-bit offset of doc_meta / the records.
#define NP_UBLOCK_MAX_U16 128   // entries (256 B: one 8-byte load per lane of half a wave): 8 header + 120 codes
#define NP_UBLOCK_MAX_U32 64    // entries (256 B): 4 header + 60 codes
// with the bit-plane hot level (approx_hotp_kernel, LPD = 4: a whole wave stages a block) blocks go up to 512 bytes, so that a
// corpus with 150-250 distinct codes per document (0.5-0.8 per token at 300 tokens) keeps its lists in the blocks
#define NP_UBLOCK_BIG_U16 256   // 8 header + 248 codes
#define NP_UBLOCK_BIG_U32 128   // 4 header + 124 codes
#define NP_ULEN_BINS 258        // list lengths 0..256, > 256

__global__ void ulen_hist_kernel(const int32_t* __restrict__ ulen, int64_t n, uint32_t* __restrict__ hist /* [NP_ULEN_BINS] */) {
  __shared__ uint32_t s_h[NP_ULEN_BINS];
  for (int i = threadIdx.x; i < NP_ULEN_BINS; i += blockDim.x) s_h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&s_h[min(ulen[i], NP_ULEN_BINS - 1)], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < NP_ULEN_BINS; i += blockDim.x)
    if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// overflow entries of document i (its whole list, 4-entry aligned) or 0 when the list fits its block
__global__ void ulen_overflow_kernel(const int32_t* __restrict__ ulen, int64_t n, int fit, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ulen[i] > fit ? ((int64_t)ulen[i] + NP_ULIST_ALIGN - 1) / NP_ULIST_ALIGN * NP_ULIST_ALIGN : 0;
}

// uoff[d] = entry offset of document d's list; the block header of d
__global__ void ublock_layout_kernel(int64_t n, const int64_t* __restrict__ doc_off, const int32_t* __restrict__ ulen,
                                     const int64_t* __restrict__ ovf_incl /* inclusive scan of the overflow sizes */,
                                     int S, int hdr, int fit, int64_t base_b, void* __restrict__ ucodes, int wide,
                                     int64_t* __restrict__ uoff) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const int32_t u = ulen[d];
  const bool ovf = u > fit;
  const int64_t ovf_size = ovf ? ((int64_t)u + NP_ULIST_ALIGN - 1) / NP_ULIST_ALIGN * NP_ULIST_ALIGN : 0;
  const int64_t ovf_excl = ovf_incl[d] - ovf_size;
  uoff[d] = ovf ? base_b + ovf_excl : d * (int64_t)S + hdr;
  uint4* h = reinterpret_cast<uint4*>(static_cast<char*>(ucodes) + d * (int64_t)S * (wide ? 4 : 2));
  *h = make_uint4((uint32_t)u, (uint32_t)(doc_off[d + 1] - doc_off[d]), ovf ? (uint32_t)(ovf_excl / NP_ULIST_ALIGN) : 0u, 0u);
}


// ---- derived layout: every document's tokens ordered by centroid code ------------------------------------------------
// S6 gathers one 128-B row of the query's score table per token (the MFMA C-in).  MaxSim is a max over the document's
// tokens, so their order is free: with the tokens of a document stored in code order, the 32 tokens of a tile name
// only a few distinct rows (the texture unit merges equal addresses of one instruction) and the gather traffic drops
// from one row per TOKEN to about one per DISTINCT code of the document.  codes / residuals are permuted in place at
// open (one workgroup per document, rows staged in LDS); tok_pos keeps the original position so that
// decompress_documents and np_hip_index_export return the on-disk order.  Documents longer than NP_SORT_MAX tokens
// stay as they are.  direction 1 = sort, 0 = restore the original order (export).
#define NP_SORT_MAX 512
template <typename CT>
__global__ void __launch_bounds__(256) sort_doc_tokens_kernel(const int64_t* __restrict__ doc_off, CT* __restrict__ codes,
                                                              uint8_t* __restrict__ residuals, uint16_t* __restrict__ tok_pos,
                                                              int pd, int to_sorted) {
  extern __shared__ uint64_t s_keys[];                        // [NP_SORT_MAX] (code << 32 | position)
  uint32_t* s_rows = reinterpret_cast<uint32_t*>(s_keys + NP_SORT_MAX);   // [len][pd / 4]
  const int tid = threadIdx.x;
  const int64_t off = doc_off[blockIdx.x];
  const int len = (int)(doc_off[blockIdx.x + 1] - off);
  if (len > NP_SORT_MAX) {
    if (to_sorted)
      for (int i = tid; i < len; i += 256) tok_pos[off + i] = (uint16_t)min(i, 65535);
    return;
  }
  if (len == 0) return;
  const int pw = pd / 4;                                       // dwords per residual row (pd is a multiple of 4 here)
  uint32_t* rows_g = reinterpret_cast<uint32_t*>(residuals + off * pd);
  for (int w = tid; w < len * pw; w += 256) s_rows[w] = rows_g[w];
  if (to_sorted) {
    int n = 1;
    while (n < len) n <<= 1;
    for (int i = tid; i < n; i += 256) s_keys[i] = (i < len) ? (((uint64_t)codes[off + i] << 32) | (uint64_t)i) : ~0ull;
    for (int k = 2; k <= n; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const uint64_t a = s_keys[i], c = s_keys[ixj];
            const bool asc = (i & k) == 0;
            if (asc ? (a > c) : (a < c)) { s_keys[i] = c; s_keys[ixj] = a; }
          }
        }
      }
    __syncthreads();
    for (int i = tid; i < len; i += 256) {
      codes[off + i] = (CT)(s_keys[i] >> 32);
      tok_pos[off + i] = (uint16_t)(s_keys[i] & 0xFFFFu);
    }
    for (int w = tid; w < len * pw; w += 256) {
      const int i = w / pw, c = w - i * pw;
      rows_g[w] = s_rows[(int)(s_keys[i] & 0xFFFFFFFFull) * pw + c];
    }
  } else {
    for (int i = tid; i < len; i += 256) s_keys[i] = ((uint64_t)codes[off + i] << 32) | (uint64_t)tok_pos[off + i];
    __syncthreads();
    for (int i = tid; i < len; i += 256) codes[off + (int)(s_keys[i] & 0xFFFFull)] = (CT)(s_keys[i] >> 32);
    for (int w = tid; w < len * pw; w += 256) {
      const int i = w / pw, c = w - i * pw;
      rows_g[(int)(s_keys[i] & 0xFFFFull) * pw + c] = s_rows[w];
    }
  }
}

static int permute_tokens(const DeviceIndex* ix, int to_sorted) {
  if (ix->n_docs == 0 || ix->T == 0 || (ix->pd & 3)) return NP_OK;
  const size_t lds = (size_t)NP_SORT_MAX * 8 + (size_t)NP_SORT_MAX * ix->pd;
  if (lds > 48 * 1024) {
    NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_doc_tokens_kernel<uint16_t>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_doc_tokens_kernel<uint32_t>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  for (int64_t d0 = 0; d0 < ix->n_docs; d0 += (int64_t)1 << 30) {
    const int64_t n = std::min<int64_t>((int64_t)1 << 30, ix->n_docs - d0);
    if (ix->code_wide)
      sort_doc_tokens_kernel<uint32_t><<<(unsigned)n, 256, lds>>>(ix->d_doc_offsets + d0, (uint32_t*)ix->d_codes, ix->d_residuals,
                                                                  ix->d_tok_pos, ix->pd, to_sorted);
    else
      sort_doc_tokens_kernel<uint16_t><<<(unsigned)n, 256, lds>>>(ix->d_doc_offsets + d0, (uint16_t*)ix->d_codes, ix->d_residuals,
                                                                  ix->d_tok_pos, ix->pd, to_sorted);
  }
  NP_HIP(hipGetLastError());
  NP_HIP(hipDeviceSynchronize());
  return NP_OK;
}

static int sort_tokens(DeviceIndex* ix) {
  const char* e = getenv("NP_TOK_SORT");
  // opt-in (NP_TOK_SORT=1): measured 1 % on S6 at 1M docs for 2 B/token of HBM, so the default keeps the on-disk order
  if (!(e && *e && atoi(e) != 0) || (ix->pd & 3)) return NP_OK;
  NP_TRY(dev_alloc(&ix->d_tok_pos, (size_t)ix->T, &ix->device_bytes));
  NP_TRY(permute_tokens(ix, 1));
  ix->tok_sorted = true;
  return NP_OK;
}

// ---- derived: 1 / ||centroid[code] + residual|| per token (codec.rs:443-467's normaliser) ------------------
// one wave per 64 consecutive tokens; lanes sweep the dims of one token at a time (coalesced rows)
__global__ void __launch_bounds__(256) inv_norm_kernel(int64_t T, int dim /* row stride */, int ldim /* dims that count */, int nbits, int pd,
                                                       const float* __restrict__ centroids,
                                                       const float* __restrict__ wlut,
                                                       CodeArr codes,
                                                       const uint8_t* __restrict__ residuals,
                                                       float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const int per = 8 / nbits;
  const uint32_t mask = (1u << nbits) - 1u;
  float mine = 0.f;
  for (int i = 0; i < 64; ++i) {
    const int64_t tok = t0 + i;
    if (tok >= T) break;
    const uint32_t code = codes[tok];
    float ss = 0.f;
    for (int j = lane; j < ldim; j += 64) {
      const uint32_t byte = residuals[tok * pd + j / per];
      const int e = j % per;
      const float x = centroids[(int64_t)code * dim + j] + wlut[(byte >> (8 - nbits * (e + 1))) & mask];
      ss = fmaf(x, x, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == i) mine = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  }
  if (t0 + lane < T) inv_norm[t0 + lane] = mine;
}

// ivf_split[c][r] = number of entries of posting list c with id < r * 32768 (r = 0 .. n_ranges): the zeroth filter level
// (gain_sweep_kernel, np_kernels.h) holds the accumulators of one 32768-document range per block and reads exactly its part of
// every probed list.  One thread per (list, boundary), a bisection each; ascending lists only.
#define NP_SPLIT_RANGE NP_IVF_SPLIT_RANGE
__global__ void __launch_bounds__(256) ivf_split_kernel(const uint32_t* __restrict__ ivf, const int64_t* __restrict__ ivf_off,
                                                        int64_t K, int R1, uint32_t* __restrict__ split) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= K * R1) return;
  const int64_t c = i / R1;
  const int r = (int)(i - c * R1);
  const int64_t s0 = ivf_off[c];
  const uint32_t len = (uint32_t)(ivf_off[c + 1] - s0);
  const uint64_t bound = (uint64_t)r * NP_SPLIT_RANGE;
  const uint32_t* L = ivf + s0;
  uint32_t a = 0, z = len;
  while (a < z) {
    const uint32_t mid = (a + z) >> 1;
    if ((uint64_t)L[mid] < bound) a = mid + 1;
    else z = mid;
  }
  split[i] = a;
}

// the planner's candidate bound (np_internal.h ivf_top_prefix), from the host copy of the list offsets
static void set_ivf_bound(DeviceIndex* ix, const int64_t* ioff) {
  const int64_t K = ix->K;
  std::vector<int64_t> len((size_t)K);
  for (int64_t c = 0; c < K; ++c) len[(size_t)c] = ioff[c + 1] - ioff[c];
  std::sort(len.begin(), len.end(), std::greater<int64_t>());
  ix->ivf_top_prefix.assign((size_t)K + 1, 0);
  for (int64_t c = 0; c < K; ++c) ix->ivf_top_prefix[(size_t)c + 1] = ix->ivf_top_prefix[(size_t)c] + len[(size_t)c];
}

static int build_ivf_split(DeviceIndex* ix) {
  ix->d_ivf_split = nullptr;
  ix->n_ranges = 0;
  if (!ix->tune.s3_gain || !ix->tune.s4_planes || !ix->ivf_sorted || ix->n_docs <= 0 || ix->K <= 0 || ix->ublock_stride <= 0) return NP_OK;
  const int R = (int)((ix->n_docs + NP_SPLIT_RANGE - 1) / NP_SPLIT_RANGE), R1 = R + 1;
  const size_t n = (size_t)ix->K * (size_t)R1;
  if (n * 4 > ((size_t)2 << 30)) return NP_OK;   // a table beyond 2 GiB (K x n_docs both huge) is not worth its HBM: the level stays off
  NP_TRY(dev_alloc(&ix->d_ivf_split, n, &ix->device_bytes));
  ivf_split_kernel<<<(unsigned)((n + 255) / 256), 256>>>(ix->d_ivf, ix->d_ivf_offsets, ix->K, R1, ix->d_ivf_split);
  NP_HIP(hipGetLastError());
  NP_HIP(hipDeviceSynchronize());
  ix->n_ranges = R;
  return NP_OK;
}

static int build_inv_norm(DeviceIndex* ix) {
  if (ix->d_inv_norm) {
    set_error("internal: inv_norm built twice");
    return NP_ERR_INVALID_ARGUMENT;
  }
  NP_TRY(dev_alloc(&ix->d_inv_norm, (size_t)ix->T, &ix->device_bytes));
  if (ix->T > 0) {
    const int64_t nblk = (ix->T + 255) / 256;
    inv_norm_kernel<<<(unsigned)nblk, 256>>>(ix->T, ix->dim, ix->ldim, ix->nbits, ix->pd, ix->d_centroids, ix->d_wlut, ix->codes(),
                                             ix->d_residuals, ix->d_inv_norm);
  }
  NP_HIP(hipGetLastError());
  NP_HIP(hipDeviceSynchronize());
  return NP_OK;
}

// ---- derived: where each document's sorted distinct-code list crosses the eighths of the centroid range ----
// seg[d] = 8 x u16, seg[x] = number of distinct codes of d that are < (x+1)*ceil(K/8).  The sliced S4 kernel walks
// one eighth (or pair / quad of eighths) of every candidate at a time so that the slice of the query's score
// table it touches stays resident in one XCD's L2.
__global__ void __launch_bounds__(256) useg_kernel(int64_t n_docs, const int64_t* __restrict__ doc_off,
                                                   const int64_t* __restrict__ uoff, CodeArr ucodes,
                                                   const int32_t* __restrict__ ulen, uint32_t slice_w,
                                                   uint4* __restrict__ useg, int* __restrict__ bad) {
  const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (d >= n_docs) return;
  const int64_t off = uoff[d];
  const int n = ulen[d];
  if (doc_off[d + 1] - doc_off[d] > NP_UNIQ_MAX || n > 65535) {   // list not sorted / counts do not fit
    *bad = 1;
    useg[d] = make_uint4(0, 0, 0, 0);
    return;
  }
  uint32_t e[8];
  int pos = 0;
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    const uint64_t hi = (uint64_t)(x + 1) * slice_w;
    while (pos < n && (uint64_t)ucodes[off + pos] < hi) ++pos;
    e[x] = (uint32_t)(x == 7 ? n : pos);
  }
  useg[d] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

__global__ void doc_meta_kernel(int64_t n_docs, const int64_t* __restrict__ doc_off, const int64_t* __restrict__ uoff,
                                const int32_t* __restrict__ ulen, uint4* __restrict__ meta) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_docs) return;
  const int64_t o = uoff[d];                                     // the document's list in d_ucodes
  const uint32_t dl = (uint32_t)(doc_off[d + 1] - doc_off[d]);   // w = offset bits 32..39 | doc length << 8
  meta[d] = make_uint4((uint32_t)d, (uint32_t)ulen[d], (uint32_t)(o & 0xFFFFFFFFll), (uint32_t)((o >> 32) & 0xFF) | (dl << 8));
}

// d_uoff (list offsets, [n_docs + 1]) is returned to the caller: the IVF build of the synthetic path reads the lists
// through it; whoever receives it frees it.
static int build_unique_codes(DeviceIndex* ix, int64_t** d_uoff_out) {
  *d_uoff_out = nullptr;
  const int64_t N = ix->n_docs;
  NP_TRY(dev_alloc(&ix->d_ulen, (size_t)N, &ix->device_bytes));
  NP_TRY(dev_alloc(&ix->d_useg, (size_t)N, &ix->device_bytes));
  NP_TRY(dev_alloc(&ix->d_doc_meta, (size_t)N, &ix->device_bytes));
  int64_t* d_uoff = nullptr;
  NP_TRY(dev_alloc(&d_uoff, (size_t)N + 1, nullptr));
  struct FreeOnError {
    int64_t** p;
    bool armed = true;
    ~FreeOnError() {
      if (armed) {
        (void)hipFree(*p);
        *p = nullptr;
      }
    }
  } guard{&d_uoff};
  NP_HIP(hipMemset(d_uoff, 0, 8));
  // pass 1: list lengths
  for (int64_t d0 = 0; d0 < N; d0 += (int64_t)1 << 30) {
    const int64_t n = std::min<int64_t>((int64_t)1 << 30, N - d0);
    unique_codes_kernel<<<(unsigned)n, 256>>>(ix->d_doc_offsets + d0, ix->codes(), nullptr, nullptr, ix->d_ulen + d0);
  }
  NP_HIP(hipGetLastError());
  // block stride: the smallest multiple of 16 bytes that holds the header and 99.9 % of the lists (at most the staging row)
  const int hdr = (int)(16 / ix->code_bytes());
  const int smax = ix->tune.s4_planes ? (ix->code_wide ? NP_UBLOCK_BIG_U32 : NP_UBLOCK_BIG_U16)
                                      : (ix->code_wide ? NP_UBLOCK_MAX_U32 : NP_UBLOCK_MAX_U16);
  int fit = smax - hdr;
  if (N > 0) {
    uint32_t* d_hist = nullptr;
    uint32_t h_hist[NP_ULEN_BINS];
    NP_HIP(hipMalloc(&d_hist, sizeof h_hist));
    hipError_t e = hipMemset(d_hist, 0, sizeof h_hist);
    if (e == hipSuccess) {
      ulen_hist_kernel<<<(unsigned)std::min<int64_t>((N + 255) / 256, 1024), 256>>>(ix->d_ulen, N, d_hist);
      e = hipMemcpy(h_hist, d_hist, sizeof h_hist, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_hist);
    NP_HIP(e);
    double usum = 0;
    for (int q2 = 0; q2 < NP_ULEN_BINS; ++q2) usum += (double)q2 * h_hist[q2];
    ix->ulen_mean = (float)(usum / (double)N);
    const int64_t allow = N / 100;    // lists allowed to overflow (1 %: the filter re-reads only those)
    int64_t over = 0;
    int q = NP_ULEN_BINS - 1;
    while (q > 0 && over + h_hist[q] <= allow) over += h_hist[q--];   // q = the smallest capacity leaving <= 0.1 % outside
    fit = std::min(fit, std::max(q, 1));
  }
  // the stride is a multiple of 64 bytes (blocks start on 64-B sectors: a 192-B block is three sectors, a 16-B aligned
  // 208-B one 4.1 on average), at most 256 bytes
  const int per64 = (int)(64 / ix->code_bytes());
  const int S = std::max(per64, std::min(smax, (hdr + fit + per64 - 1) / per64 * per64));
  fit = S - hdr;
  ix->ublock_stride = S;
  ix->ublock_hdr = hdr;
  const int64_t base_b = N * (int64_t)S;   // first entry of the overflow region
  int64_t ovf_total = 0;
  int64_t* d_ovf = nullptr;                // inclusive scan of the overflow sizes
  struct FreeOvf {
    int64_t** p;
    ~FreeOvf() { (void)hipFree(*p); }
  } free_ovf{&d_ovf};
  if (N > 0) {
    int64_t* d_pad = nullptr;
    void* d_temp = nullptr;
    NP_TRY(dev_alloc(&d_pad, (size_t)N, nullptr));
    hipError_t e = hipMalloc(&d_ovf, (size_t)N * 8);
    if (e == hipSuccess) {
      ulen_overflow_kernel<<<(unsigned)((N + 255) / 256), 256>>>(ix->d_ulen, N, fit, d_pad);
      size_t tb = 0;
      e = hipcub::DeviceScan::InclusiveSum(nullptr, tb, d_pad, d_ovf, (int)N);
      if (e == hipSuccess) e = hipMalloc(&d_temp, std::max<size_t>(tb, 16));
      if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(d_temp, tb, d_pad, d_ovf, (int)N);
      if (e == hipSuccess) e = hipMemcpy(&ovf_total, d_ovf + (N - 1), 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_pad);
    (void)hipFree(d_temp);
    if (e != hipSuccess) {
      set_error("distinct-code list scan failed: %s", hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? NP_ERR_OUT_OF_MEMORY : NP_ERR_DEVICE_UNAVAILABLE;
    }
  }
  const int64_t total = base_b + ovf_total;
  if (total >= ((int64_t)1 << 40)) {   // candidate records carry a 40-bit list offset
    set_error("Index load failed: %lld distinct-code list entries exceed the 40-bit list offset; use more shards",
              (long long)total);
    return NP_ERR_INDEX_LOAD;
  }
  if (ovf_total / NP_ULIST_ALIGN >= ((int64_t)1 << 32)) {   // a block header carries the overflow index in 32 bits
    set_error("Index load failed: %lld overflow list entries exceed the 32-bit overflow index of the list blocks; use more shards",
              (long long)ovf_total);
    return NP_ERR_INDEX_LOAD;
  }
  ix->n_ucodes = total;
  // pass 2: headers, then the lists (+8 entries: list readers fetch up to 8 bytes past a list's last code)
  {
    const size_t bytes = ((size_t)total + 8) * ix->code_bytes();
    hipError_t e = hipMalloc(&ix->d_ucodes, bytes);
    if (e != hipSuccess) {
      ix->d_ucodes = nullptr;
      set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return NP_ERR_OUT_OF_MEMORY;
    }
    ix->device_bytes += bytes;
    NP_HIP(hipMemset(ix->d_ucodes, 0, bytes));
  }
  if (N > 0)
    ublock_layout_kernel<<<(unsigned)((N + 255) / 256), 256>>>(N, ix->d_doc_offsets, ix->d_ulen, d_ovf, S, hdr, fit, base_b,
                                                               ix->d_ucodes, ix->code_wide, d_uoff);
  for (int64_t d0 = 0; d0 < N; d0 += (int64_t)1 << 30) {
    const int64_t n = std::min<int64_t>((int64_t)1 << 30, N - d0);
    unique_codes_kernel<<<(unsigned)n, 256>>>(ix->d_doc_offsets + d0, ix->codes(), ix->d_ucodes, d_uoff + d0, ix->d_ulen + d0);
  }
  if (N > 0)
    doc_meta_kernel<<<(unsigned)((N + 255) / 256), 256>>>(N, ix->d_doc_offsets, d_uoff, ix->d_ulen, ix->d_doc_meta);
  NP_HIP(hipGetLastError());
  ix->sliced_ok = false;
  if (N > 0) {
    int* d_bad = nullptr;
    NP_HIP(hipMalloc(&d_bad, sizeof(int)));
    NP_HIP(hipMemset(d_bad, 0, sizeof(int)));
    const uint32_t slice_w = (uint32_t)((ix->K + 7) / 8);
    useg_kernel<<<(unsigned)((N + 255) / 256), 256>>>(N, ix->d_doc_offsets, d_uoff, ix->ucodes(), ix->d_ulen, slice_w,
                                                      ix->d_useg, d_bad);
    int bad = 1;
    hipError_t e = hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    NP_HIP(e);
    ix->sliced_ok = (bad == 0);
  }
  NP_HIP(hipDeviceSynchronize());
  guard.armed = false;
  *d_uoff_out = d_uoff;
  return NP_OK;
}

// ---- from host arrays / files ----------------------------------------------------------------------
// ---- file geometry -> storage geometry (np_internal.h storage_dim / storage_nbits) --------------------------------
// one thread per storage byte.  widen: a 1-bit file byte (8 dims, first dim in bit 7) becomes two 2-bit storage bytes whose
// segment e holds bitrev2(bucket) = bucket << 1 (the bit layout of codec.rs:356-411 at nbits = 2).
__global__ void __launch_bounds__(256) repack_rows_kernel(const uint8_t* __restrict__ src, int64_t n, int lpd, int pd,
                                                          int widen, uint8_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * pd) return;
  const int64_t t = i / pd;
  const int jb = (int)(i - t * pd);
  uint32_t out = 0;
  if (widen) {
    const int sb = jb >> 1;
    if (sb < lpd) {
      const uint32_t nib = ((uint32_t)src[t * lpd + sb] >> ((jb & 1) ? 0 : 4)) & 15u;
      out = ((nib & 8u) << 4) | ((nib & 4u) << 3) | ((nib & 2u) << 2) | ((nib & 1u) << 1);
    }
  } else if (jb < lpd) {
    out = src[t * lpd + jb];
  }
  dst[i] = (uint8_t)out;
}

// the inverse, for np_hip_index_export
__global__ void __launch_bounds__(256) unpack_rows_kernel(const uint8_t* __restrict__ src, int64_t n, int lpd, int pd,
                                                          int widen, uint8_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * lpd) return;
  const int64_t t = i / lpd;
  const int jb = (int)(i - t * lpd);
  uint32_t out;
  if (widen) {
    const uint32_t hi = src[t * pd + 2 * jb], lo = src[t * pd + 2 * jb + 1];
    const uint32_t nh = ((hi >> 4) & 8u) | ((hi >> 3) & 4u) | ((hi >> 2) & 2u) | ((hi >> 1) & 1u);
    const uint32_t nl = ((lo >> 4) & 8u) | ((lo >> 3) & 4u) | ((lo >> 2) & 2u) | ((lo >> 1) & 1u);
    out = (nh << 4) | nl;
  } else {
    out = src[t * pd + jb];
  }
  dst[i] = (uint8_t)out;
}

// NP_OPEN_TRACE=1: seconds of each phase of an open to stderr (parse, token upload, posting lists, derived structures)
struct OpenTrace {
  bool on = getenv("NP_OPEN_TRACE") != nullptr;
  double t0 = now();
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
  }
  void mark(const char* what) {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const double t = now();
    fprintf(stderr, "[np open] %-28s %8.3f s\n", what, t - t0);
    t0 = t;
  }
};

// ---- host -> HBM at the link's rate ----------------------------------------------------------------------------------
// MmapIndex::load maps the chunk files and touches them lazily (index.rs:1096-1124); here the whole index has to cross
// PCIe once.  A plain hipMemcpy from the mapping is a single-threaded copy out of the page cache into the runtime's own
// staging buffer (3-6 GB/s: a 216 GB index would open in a minute).  Uploader keeps NP_UP_SLOTS pinned pieces in flight:
// NP_UP_THREADS host threads copy a piece out of the mapping in parallel (page faults and all), the DMA engine moves the
// previous piece meanwhile, and whatever the data needs (codes i64 -> u16 / u32 with the range check, residual rows to
// storage geometry) is done by a kernel on the staged bytes in HBM -- no scalar loop on the host touches a token.
#define NP_UP_SLOTS 3
#define NP_UP_PIECE ((size_t)128 << 20)
#define NP_UP_THREADS 12
struct Uploader {
  hipStream_t st = nullptr;
  char* pin[NP_UP_SLOTS] = {};
  hipEvent_t done[NP_UP_SLOTS] = {};
  int next = 0;
  int init() {
    NP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < NP_UP_SLOTS; ++i) {
      NP_HIP(hipHostMalloc((void**)&pin[i], NP_UP_PIECE, hipHostMallocDefault));
      NP_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    return NP_OK;
  }
  ~Uploader() {
    if (st) (void)hipStreamSynchronize(st);
    for (int i = 0; i < NP_UP_SLOTS; ++i) {
      if (pin[i]) (void)hipHostFree(pin[i]);
      if (done[i]) (void)hipEventDestroy(done[i]);
    }
    if (st) (void)hipStreamDestroy(st);
  }
  // parallel copy of [src, src + bytes) into a pinned slot (bytes <= NP_UP_PIECE); returns the slot
  int fill(const void* src, size_t bytes, int* slot) {
    const int s = next;
    next = (next + 1) % NP_UP_SLOTS;
    NP_HIP(hipEventSynchronize(done[s]));   // the slot's previous piece has left (a fresh event is complete)
    const size_t per = ((bytes + NP_UP_THREADS - 1) / NP_UP_THREADS + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (int t = 1; t < NP_UP_THREADS; ++t) {
      const size_t a = (size_t)t * per;
      if (a >= bytes) break;
      th.emplace_back([=] { memcpy(pin[s] + a, (const char*)src + a, std::min(per, bytes - a)); });
    }
    memcpy(pin[s], src, std::min(per, bytes));
    for (auto& t : th) t.join();
    *slot = s;
    return NP_OK;
  }
  // the staged piece -> dst (device) ; `after` runs on the stream once the bytes are there
  template <class F>
  int send(int slot, size_t bytes, void* dst, F&& after) {
    NP_HIP(hipMemcpyAsync(dst, pin[slot], bytes, hipMemcpyHostToDevice, st));
    NP_TRY(after(st));
    NP_HIP(hipEventRecord(done[slot], st));
    return NP_OK;
  }
  int finish() {
    NP_HIP(hipStreamSynchronize(st));
    return NP_OK;
  }
};

// codes as stored (i64, N.codes.npy) -> u16 / u32 with the loader's range check: bad[0] = flag (0 = every code in range),
// bad[1] = one offending value (the thread that raises the flag stores it: any i64, -1 included, is reportable)
__global__ void __launch_bounds__(256) narrow_codes_kernel(const int64_t* __restrict__ src, int64_t n, int64_t K, void* __restrict__ dst,
                                                           int wide, long long* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t v = src[i];
  if (v < 0 || v >= K) {
    if (atomicExch((unsigned long long*)bad, 1ull) == 0ull) bad[1] = v;
    if (wide) static_cast<uint32_t*>(dst)[i] = 0u;   // never leave an unwritten (garbage) code behind
    else static_cast<uint16_t*>(dst)[i] = 0;
    return;
  }
  if (wide) static_cast<uint32_t*>(dst)[i] = (uint32_t)v;
  else static_cast<uint16_t*>(dst)[i] = (uint16_t)v;
}

static int upload_repacked(DeviceIndex* ix, const uint8_t* rows, int64_t first_tok, int64_t n_tok) {
  const int64_t PIECE = (int64_t)1 << 20;   // tokens per staging piece
  uint8_t* d_stage = nullptr;
  NP_HIP(hipMalloc(&d_stage, (size_t)std::min(PIECE, n_tok) * ix->lpd));
  hipError_t e = hipSuccess;
  const int widen = ix->lnbits != ix->nbits;
  for (int64_t s = 0; s < n_tok && e == hipSuccess; s += PIECE) {
    const int64_t n = std::min(PIECE, n_tok - s);
    e = hipMemcpy(d_stage, rows + s * ix->lpd, (size_t)n * ix->lpd, hipMemcpyHostToDevice);
    if (e != hipSuccess) break;
    repack_rows_kernel<<<(unsigned)((n * ix->pd + 255) / 256), 256>>>(d_stage, n, ix->lpd, ix->pd, widen,
                                                                     ix->d_residuals + (first_tok + s) * ix->pd);
    e = hipDeviceSynchronize();
  }
  (void)hipFree(d_stage);
  NP_HIP(e);
  return NP_OK;
}

int build_device_index(const HostIndex& h, const np_open_opts* opts_in, DeviceIndex** out) {
  *out = nullptr;
  np_open_opts o;
  NP_TRY(normalise_opts(opts_in, &o));
  NP_TRY(check_device(o.device));
  NP_TRY(check_geometry(h.K, h.dim, h.nbits, h.num_documents_total));
  int64_t sb, se;
  shard_range(h.num_documents_total, o.shard_rank, o.shard_count, &sb, &se);
  const int64_t hb = h.doc_begin, he = h.doc_begin + (int64_t)h.doc_lengths.size();
  if (sb < hb || se > he) {
    set_error("Index load failed: host arrays cover docs [%lld,%lld) but shard %d/%d needs [%lld,%lld)", (long long)hb,
              (long long)he, o.shard_rank, o.shard_count, (long long)sb, (long long)se);
    return NP_ERR_INDEX_LOAD;
  }
  DeviceGuard g(o.device);
  if (!g.ok) {
    set_error("hipSetDevice(%d) failed", o.device);
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  np_index* nix = new np_index();
  DeviceIndex* ix = nix;
  struct Cleanup {
    np_index* p;
    ~Cleanup() {
      if (p) {
        destroy_device_index(p);
        delete p;
      }
    }
  } cleanup{nix};

  ix->device = o.device;
  ix->opts = o;
  read_tuning_env(&ix->tune);
  ix->N_total = h.num_documents_total;
  ix->n_emb_total = h.num_embeddings_total;
  ix->avg_doclen = h.avg_doclen;
  ix->doc_begin = sb;
  ix->n_docs = se - sb;
  if (ix->n_docs >= ((int64_t)1 << 31)) {
    set_error("Index load failed: a shard holds < 2^31 documents (got %lld); use more shards", (long long)ix->n_docs);
    return NP_ERR_INDEX_LOAD;
  }
  ix->K = h.K;
  ix->KP = (h.K + 63) / 64 * 64;
  ix->code_wide = h.K > 65536 ? 1 : 0;   // u16 codes whenever every centroid id fits (2 B per token instead of 4)
  ix->ldim = h.dim;
  ix->lnbits = h.nbits;
  ix->lpd = h.dim * h.nbits / 8;
  ix->dim = storage_dim(h.dim);
  ix->nbits = storage_nbits(h.dim, h.nbits);
  ix->pd = ix->dim * ix->nbits / 8;
  OpenTrace trace;
  NP_TRY(upload_codec(ix, h.centroids, h.bucket_weights));
  trace.mark("codec");

  // doc offsets of the shard + its token range inside the host arrays
  int64_t tb = 0;
  for (int64_t d = hb; d < sb; ++d) tb += h.doc_lengths[d - hb];
  std::vector<int64_t> off((size_t)ix->n_docs + 1);
  off[0] = 0;
  int64_t maxlen = 0;
  for (int64_t d = 0; d < ix->n_docs; ++d) {
    int64_t l = h.doc_lengths[sb - hb + d];
    if (l < 0) {
      set_error("Index load failed: negative doc length");
      return NP_ERR_INDEX_LOAD;
    }
    maxlen = std::max(maxlen, l);
    off[d + 1] = off[d] + l;
  }
  ix->T = off[ix->n_docs];
  ix->max_doc_len = maxlen;
  const int64_t te = tb + ix->T;
  NP_TRY(dev_alloc(&ix->d_doc_offsets, off.size(), &ix->device_bytes));
  NP_HIP(hipMemcpy(ix->d_doc_offsets, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));

  // codes (i64 -> u32, range-checked) and residuals, chunk by chunk
  {
    uint8_t* cbuf = nullptr;
    NP_TRY(dev_alloc(&cbuf, (size_t)std::max<int64_t>(ix->T, 1) * ix->code_bytes(), &ix->device_bytes));
    ix->d_codes = cbuf;
  }
  NP_TRY(dev_alloc(&ix->d_residuals, (size_t)ix->T * ix->pd, &ix->device_bytes));
  {
    Uploader up;
    NP_TRY(up.init());
    // device-side staging of raw pieces that need a kernel (i64 codes; residual rows in file geometry)
    char* d_stage[NP_UP_SLOTS] = {};
    long long* d_bad = nullptr;
    struct FreeStage {
      char** p;
      long long** b;
      ~FreeStage() {
        for (int i = 0; i < NP_UP_SLOTS; ++i) (void)hipFree(p[i]);
        (void)hipFree(*b);
      }
    } free_stage{d_stage, &d_bad};
    for (int i = 0; i < NP_UP_SLOTS; ++i) NP_HIP(hipMalloc(&d_stage[i], NP_UP_PIECE));
    NP_HIP(hipMalloc(&d_bad, 16));
    NP_HIP(hipMemset(d_bad, 0, 16));
    const bool repack = ix->pd != ix->lpd;
    const int widen = ix->lnbits != ix->nbits;
    int64_t pos = 0;  // token position of the current chunk's first token in the host arrays
    for (const HostChunk& c : h.chunks) {
      const int64_t a = std::max(pos, tb), b = std::min(pos + c.n_tokens, te);
      // codes: raw i64 pieces, narrowed and range-checked on the device
      const int64_t CP = (int64_t)(NP_UP_PIECE / 8);
      for (int64_t t0 = a; t0 < b; t0 += CP) {
        const int64_t n = std::min(CP, b - t0);
        int slot;
        NP_TRY(up.fill((const char*)c.codes + (t0 - pos) * 8, (size_t)n * 8, &slot));
        NP_TRY(up.send(slot, (size_t)n * 8, d_stage[slot], [&](hipStream_t st) -> int {
          narrow_codes_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(
              (const int64_t*)d_stage[slot], n, ix->K, (char*)ix->d_codes + (t0 - tb) * (int64_t)ix->code_bytes(), ix->code_wide, d_bad);
          NP_HIP(hipGetLastError());
          return (int)NP_OK;
        }));
      }
      // residuals: straight into place, or through the staging area + repack (file rows -> storage rows: zero-padded,
      // 1-bit buckets widened to 2-bit segments)
      const int64_t RP = (int64_t)(NP_UP_PIECE / std::max(ix->lpd, 1));
      for (int64_t t0 = a; t0 < b; t0 += RP) {
        const int64_t n = std::min(RP, b - t0);
        int slot;
        NP_TRY(up.fill(c.residuals + (t0 - pos) * ix->lpd, (size_t)n * ix->lpd, &slot));
        if (!repack) {
          NP_TRY(up.send(slot, (size_t)n * ix->lpd, ix->d_residuals + (t0 - tb) * ix->pd, [](hipStream_t) -> int { return NP_OK; }));
        } else {
          NP_TRY(up.send(slot, (size_t)n * ix->lpd, d_stage[slot], [&](hipStream_t st) -> int {
            repack_rows_kernel<<<(unsigned)((n * ix->pd + 255) / 256), 256, 0, st>>>((const uint8_t*)d_stage[slot], n, ix->lpd, ix->pd,
                                                                                    widen, ix->d_residuals + (t0 - tb) * ix->pd);
            NP_HIP(hipGetLastError());
            return (int)NP_OK;
          }));
        }
      }
      pos += c.n_tokens;
    }
    NP_TRY(up.finish());
    if (pos < te) {
      set_error("Index load failed: chunks hold %lld tokens, doclens need %lld", (long long)pos, (long long)te);
      return NP_ERR_INDEX_LOAD;
    }
    long long bad[2] = {0, 0};
    NP_HIP(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
    if (bad[0] != 0) {
      set_error("Index load failed: code %lld out of range [0,%lld)", bad[1], (long long)ix->K);
      return NP_ERR_INDEX_LOAD;
    }
  }

  trace.mark("codes + residuals -> HBM");
  // IVF restricted to the shard, re-based to shard-local u32 ids: two passes over the posting lists (count, then fill), the
  // centroid range split over host threads (680 M i64 entries at 10 M x 300-token documents)
  {
    std::vector<int64_t> ioff((size_t)ix->K + 1, 0), lstart((size_t)ix->K + 1, 0);
    for (int64_t c = 0; c < ix->K; ++c) {
      const int64_t l = h.ivf_lengths[c];
      if (l < 0 || lstart[c] + l > h.ivf_size) {
        set_error("Index load failed: ivf_lengths inconsistent with ivf.npy at centroid %lld", (long long)c);
        return NP_ERR_INDEX_LOAD;
      }
      lstart[c + 1] = lstart[c] + l;
    }
    const int NT = (int)std::max<int64_t>(1, std::min<int64_t>(NP_UP_THREADS, ix->K / 1024));
    const bool whole = sb == 0 && se == h.num_documents_total;
    std::vector<int64_t> bad_id((size_t)NT, -1);
    auto run = [&](auto&& body) {
      std::vector<std::thread> th;
      for (int t = 1; t < NT; ++t) th.emplace_back([&, t] { body(t); });
      body(0);
      for (auto& x : th) x.join();
    };
    auto range = [&](int t, int64_t* c0, int64_t* c1) {   // centroid ranges of about equal posting volume
      const int64_t tot = lstart[ix->K];
      *c0 = std::lower_bound(lstart.begin(), lstart.end(), tot * t / NT) - lstart.begin();
      *c1 = t + 1 == NT ? ix->K : std::lower_bound(lstart.begin(), lstart.end(), tot * (t + 1) / NT) - lstart.begin();
      *c0 = std::min<int64_t>(*c0, ix->K);
      *c1 = std::min<int64_t>(std::max(*c1, *c0), ix->K);
    };
    run([&](int t) {   // pass 1: entries of every list that fall into the shard, ids range-checked
      int64_t c0, c1;
      range(t, &c0, &c1);
      for (int64_t c = c0; c < c1; ++c) {
        int64_t cnt = 0;
        const char* src = (const char*)h.ivf + lstart[c] * 8;
        for (int64_t i = 0, l = lstart[c + 1] - lstart[c]; i < l; ++i) {
          int64_t id;
          memcpy(&id, src + i * 8, 8);
          if (id < 0 || id >= h.num_documents_total) {
            bad_id[t] = id;
            return;
          }
          cnt += whole || (id >= sb && id < se);
        }
        ioff[c + 1] = cnt;
      }
    });
    for (int t = 0; t < NT; ++t)
      if (bad_id[t] != -1) {
        set_error("Index load failed: ivf doc id %lld out of range", (long long)bad_id[t]);
        return NP_ERR_INDEX_LOAD;
      }
    for (int64_t c = 0; c < ix->K; ++c) ioff[c + 1] += ioff[c];
    std::vector<uint32_t> ivf((size_t)ioff[ix->K]);
    std::vector<int> unsorted((size_t)NT, 0);   // the crate writes ascending unique ids (index.rs:479-504); S3 relies on it only if true
    run([&](int t) {   // pass 2
      int64_t c0, c1;
      range(t, &c0, &c1);
      for (int64_t c = c0; c < c1; ++c) {
        uint32_t* out = ivf.data() + ioff[c];
        const char* src = (const char*)h.ivf + lstart[c] * 8;
        int64_t prev = -1;
        for (int64_t i = 0, l = lstart[c + 1] - lstart[c]; i < l; ++i) {
          int64_t id;
          memcpy(&id, src + i * 8, 8);
          if (id <= prev) unsorted[(size_t)t] = 1;
          prev = id;
          if (whole || (id >= sb && id < se)) *out++ = (uint32_t)(id - sb);
        }
      }
    });
    ix->ivf_sorted = true;
    for (int t = 0; t < NT; ++t)
      if (unsorted[(size_t)t]) ix->ivf_sorted = false;
    ix->ivf_size = (int64_t)ivf.size();
    NP_TRY(dev_alloc(&ix->d_ivf, ivf.size(), &ix->device_bytes));
    if (!ivf.empty()) NP_HIP(hipMemcpy(ix->d_ivf, ivf.data(), ivf.size() * 4, hipMemcpyHostToDevice));
    NP_TRY(dev_alloc(&ix->d_ivf_offsets, ioff.size(), &ix->device_bytes));
    NP_HIP(hipMemcpy(ix->d_ivf_offsets, ioff.data(), ioff.size() * 8, hipMemcpyHostToDevice));
    set_ivf_bound(ix, ioff.data());
  }
  trace.mark("posting lists");
  NP_TRY(sort_tokens(ix));
  {
    int64_t* d_uoff = nullptr;
    const int rc = build_unique_codes(ix, &d_uoff);
    (void)hipFree(d_uoff);
    NP_TRY(rc);
  }
  trace.mark("distinct-code blocks");
  NP_TRY(build_ivf_split(ix));
  NP_TRY(build_inv_norm(ix));
  trace.mark("inverse norms");
  default_workspace(ix);
  cleanup.p = nullptr;
  *out = ix;
  return NP_OK;
}

// ---- synthetic corpus generated in HBM (spec: next_plaid_amd/synth.py) -----------------------------------
__host__ __device__ static inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
enum { S_LEN = 1, S_TOPIC = 2, S_TOK = 3, S_RES = 4 };

struct SynthP {
  uint64_t b_len, b_topic, b_tok, b_res;
  int64_t doc_begin, n_docs;
  uint32_t K;
  int32_t len_min, len_span, n_topics, rand256, pd, nw;
};

__global__ void synth_lens_kernel(SynthP p, const int32_t* __restrict__ len_table, int len_table_size, int64_t* lens) {
  int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= p.n_docs) return;
  const uint64_t r = mix64(p.b_len + (uint64_t)(p.doc_begin + d));
  lens[d] = len_table ? (int64_t)len_table[r % (uint64_t)len_table_size] : p.len_min + (int64_t)(r % (uint64_t)p.len_span);
}

// one wave per document
__global__ void __launch_bounds__(256) synth_tokens_kernel(SynthP p, const int64_t* __restrict__ doc_off,
                                                           void* __restrict__ codes, int wide, uint8_t* __restrict__ res) {
  const int lane = threadIdx.x & 63;
  int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= p.n_docs) return;
  const uint64_t gdoc = (uint64_t)(p.doc_begin + d);
  const int64_t off = doc_off[d];
  const int len = (int)(doc_off[d + 1] - off);
  for (int t = lane; t < len; t += 64) {
    uint64_t r = mix64(p.b_tok + (gdoc * 65536ull + (uint64_t)t));
    uint32_t code;
    if ((r & 0xFF) < (uint64_t)p.rand256) {
      code = (uint32_t)(((r >> 8) & 0xFFFFFFFFull) % p.K);
    } else {
      uint64_t s = (r >> 40) % (uint64_t)p.n_topics;
      uint64_t rt = mix64(p.b_topic + (gdoc * (uint64_t)p.n_topics + s));
      code = (uint32_t)(((rt & 0xFFFFFFFFull) % p.K) >> ((rt >> 32) & 3));
    }
    if (wide) static_cast<uint32_t*>(codes)[off + t] = code;
    else static_cast<uint16_t*>(codes)[off + t] = (uint16_t)code;
  }
  const int nwords = len * p.nw;
  for (int w = lane; w < nwords; w += 64) {
    int t = w / p.nw, j = w - t * p.nw;
    uint64_t v = mix64(p.b_res + ((gdoc * 65536ull + (uint64_t)t) * 16ull + (uint64_t)j));
    uint8_t* dst = res + (off + t) * (int64_t)p.pd + j * 8;
    int nb = min(8, p.pd - j * 8);
    if (nb == 8 && (p.pd & 7) == 0) {
      *(uint64_t*)dst = v;  // little endian
    } else {
      for (int b = 0; b < nb; ++b) dst[b] = (uint8_t)(v >> (8 * b));
    }
  }
}

// ---- IVF from the per-document distinct-code lists (index.rs:479-499: per centroid the ascending unique doc ids) ----
// Pairs (code << 32 | shard-local doc) are emitted from ucodes/ulen for a contiguous document range, radix-sorted,
// and scattered into the posting lists.  Ranges hold at most NP_IVF_SEG pairs, so every hipcub call sees a count
// below 2^31 whatever the shard size (a 10M-document x 300-token shard has 3.0e9 tokens, ~0.7e9 pairs).
#define NP_IVF_SEG ((int64_t)1 << 29)

// one wave per document: pair position = (exclusive prefix of ulen) - seg_base
__global__ void __launch_bounds__(256) ivf_emit_pairs_kernel(int64_t d0, int64_t d1, const int64_t* __restrict__ uoff,
                                                             CodeArr ucodes,
                                                             const int32_t* __restrict__ ulen,
                                                             const int64_t* __restrict__ upfx /* inclusive */,
                                                             int64_t seg_base, uint64_t* __restrict__ keys) {
  const int64_t d = d0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (d >= d1) return;
  const int n = ulen[d];
  const int64_t pos = upfx[d] - n - seg_base;
  const int64_t off = uoff[d];
  for (int i = lane; i < n; i += 64) keys[pos + i] = ((uint64_t)ucodes[off + i] << 32) | (uint64_t)d;
}

// start[c] = first sorted pair whose code is >= c  (c = 0..K; start[K] = n)
__global__ void ivf_code_starts_kernel(const uint64_t* __restrict__ sorted, int64_t n, int64_t K,
                                       int64_t* __restrict__ start) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > K) return;
  const uint64_t key = (uint64_t)c << 32;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  start[c] = lo;
}

// ivf[dst_base[code] + (i - start[code])] = doc
__global__ void ivf_scatter_kernel(const uint64_t* __restrict__ sorted, int64_t n, const int64_t* __restrict__ start,
                                   const int64_t* __restrict__ dst_base, uint32_t* __restrict__ ivf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = sorted[i];
  const uint32_t c = (uint32_t)(k >> 32);
  ivf[dst_base[c] + (i - start[c])] = (uint32_t)(k & 0xFFFFFFFFull);
}

__global__ void ulen_to_i64_kernel(const int32_t* __restrict__ ulen, int64_t n, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ulen[i];
}

static int build_ivf_from_ucodes(DeviceIndex* ix, const int64_t* d_uoff) {
  const int64_t K = ix->K, N = ix->n_docs;
  std::vector<int64_t> ioff((size_t)K + 1, 0);
  ix->ivf_size = 0;
  int64_t* d_upfx = nullptr;
  uint64_t *d_k1 = nullptr, *d_k2 = nullptr;
  int64_t *d_start = nullptr, *d_base = nullptr;
  void* d_temp = nullptr;
  struct Free {
    void** p[6];
    ~Free() {
      for (void** q : p) (void)hipFree(*q);
    }
  } fr{{(void**)&d_upfx, (void**)&d_k1, (void**)&d_k2, (void**)&d_start, (void**)&d_base, &d_temp}};
  std::vector<int64_t> upfx((size_t)N);
  if (N > 0) {
    NP_TRY(dev_alloc(&d_upfx, (size_t)N, nullptr));
    NP_TRY(dev_alloc(&d_start, (size_t)N, nullptr));   // scan input (freed below; d_start is re-allocated per code later)
    ulen_to_i64_kernel<<<(unsigned)((N + 255) / 256), 256>>>(ix->d_ulen, N, d_start);
    size_t tb = 0;
    NP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, d_start, d_upfx, (int)N));
    NP_HIP(hipMalloc(&d_temp, std::max<size_t>(tb, 16)));
    NP_HIP(hipcub::DeviceScan::InclusiveSum(d_temp, tb, d_start, d_upfx, (int)N));
    NP_HIP(hipMemcpy(upfx.data(), d_upfx, (size_t)N * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_temp);
    d_temp = nullptr;
    (void)hipFree(d_start);
    d_start = nullptr;
  }
  const int64_t total = N > 0 ? upfx[(size_t)N - 1] : 0;
  // document ranges of at most NP_IVF_SEG pairs (a single document never exceeds 65535 distinct codes... any size fits)
  std::vector<int64_t> seg_d0;
  for (int64_t d = 0; d < N;) {
    seg_d0.push_back(d);
    const int64_t base = d > 0 ? upfx[(size_t)d - 1] : 0;
    const int64_t e = std::upper_bound(upfx.begin() + d, upfx.end(), base + NP_IVF_SEG) - upfx.begin();
    d = std::max<int64_t>(e, d + 1);
  }
  seg_d0.push_back(N);
  const int nseg = (int)seg_d0.size() - 1;
  int64_t max_pairs = 0;
  for (int s = 0; s < nseg; ++s) {
    const int64_t b = seg_d0[s] > 0 ? upfx[(size_t)seg_d0[s] - 1] : 0, e = seg_d0[s + 1] > 0 ? upfx[(size_t)seg_d0[s + 1] - 1] : 0;
    max_pairs = std::max(max_pairs, e - b);
  }
  if (max_pairs >= ((int64_t)1 << 31)) {
    set_error("Index load failed: one document range holds %lld (code, doc) pairs", (long long)max_pairs);
    return NP_ERR_INDEX_LOAD;
  }
  ix->ivf_size = total;
  NP_TRY(dev_alloc(&ix->d_ivf, (size_t)total, &ix->device_bytes));
  if (total > 0) {
    int kbits = 1;
    while (((int64_t)1 << kbits) < K) ++kbits;
    NP_TRY(dev_alloc(&d_k1, (size_t)max_pairs, nullptr));
    NP_TRY(dev_alloc(&d_k2, (size_t)max_pairs, nullptr));
    NP_TRY(dev_alloc(&d_start, (size_t)K + 1, nullptr));
    NP_TRY(dev_alloc(&d_base, (size_t)K + 1, nullptr));
    size_t tb = 0;
    NP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_k1, d_k2, (int)max_pairs, 0, 32 + kbits));
    NP_HIP(hipMalloc(&d_temp, std::max<size_t>(tb, 16)));
    std::vector<std::vector<int64_t>> seg_start((size_t)nseg, std::vector<int64_t>((size_t)K + 1));
    auto sort_segment = [&](int s, int64_t* n_out) -> int {
      const int64_t d0 = seg_d0[s], d1 = seg_d0[s + 1];
      const int64_t base = d0 > 0 ? upfx[(size_t)d0 - 1] : 0, n = upfx[(size_t)d1 - 1] - base;
      *n_out = n;
      if (n == 0) return NP_OK;
      ivf_emit_pairs_kernel<<<(unsigned)((d1 - d0 + 3) / 4), 256>>>(d0, d1, d_uoff, ix->ucodes(), ix->d_ulen,
                                                                   d_upfx, base, d_k1);
      size_t tbs = tb;
      NP_HIP(hipcub::DeviceRadixSort::SortKeys(d_temp, tbs, d_k1, d_k2, (int)n, 0, 32 + kbits));
      ivf_code_starts_kernel<<<(unsigned)((K + 256) / 256), 256>>>(d_k2, n, K, d_start);
      NP_HIP(hipGetLastError());
      return NP_OK;
    };
    // pass 1: per-range code counts (a single range keeps its sorted pairs for the scatter below)
    for (int s = 0; s < nseg; ++s) {
      int64_t n = 0;
      NP_TRY(sort_segment(s, &n));
      if (n == 0) std::fill(seg_start[(size_t)s].begin(), seg_start[(size_t)s].end(), 0);
      else NP_HIP(hipMemcpy(seg_start[(size_t)s].data(), d_start, ((size_t)K + 1) * 8, hipMemcpyDeviceToHost));
    }
    for (int64_t c = 0; c < K; ++c) {
      int64_t len = 0;
      for (int s = 0; s < nseg; ++s) len += seg_start[(size_t)s][(size_t)c + 1] - seg_start[(size_t)s][(size_t)c];
      ioff[(size_t)c + 1] = ioff[(size_t)c] + len;
    }
    // pass 2: scatter each range's lists behind the earlier ranges' (ranges are ascending in doc id)
    std::vector<int64_t> base_c(ioff.begin(), ioff.end());
    for (int s = 0; s < nseg; ++s) {
      int64_t n = 0;
      if (nseg > 1) NP_TRY(sort_segment(s, &n));
      else n = total;
      if (n > 0) {
        NP_HIP(hipMemcpy(d_base, base_c.data(), ((size_t)K + 1) * 8, hipMemcpyHostToDevice));
        ivf_scatter_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d_k2, n, d_start, d_base, ix->d_ivf);
        NP_HIP(hipGetLastError());
        NP_HIP(hipDeviceSynchronize());
      }
      for (int64_t c = 0; c < K; ++c) base_c[(size_t)c] += seg_start[(size_t)s][(size_t)c + 1] - seg_start[(size_t)s][(size_t)c];
    }
  }
  NP_TRY(dev_alloc(&ix->d_ivf_offsets, ioff.size(), &ix->device_bytes));
  NP_HIP(hipMemcpy(ix->d_ivf_offsets, ioff.data(), ioff.size() * 8, hipMemcpyHostToDevice));
  set_ivf_bound(ix, ioff.data());
  NP_HIP(hipDeviceSynchronize());
  ix->ivf_sorted = true;   // (code, document) pairs radix-sorted, ranges ascending: every list ascends
  return NP_OK;
}

static int synth_build(const np_synth_spec* s, const np_open_opts* opts_in, DeviceIndex** out) {
  *out = nullptr;
  np_open_opts o;
  NP_TRY(normalise_opts(opts_in, &o));
  NP_TRY(check_device(o.device));
  if (!s || !s->centroids || !s->bucket_weights) {
    set_error("Invalid configuration: synth spec needs centroids and bucket_weights");
    return NP_ERR_INVALID_ARGUMENT;
  }
  NP_TRY(check_geometry(s->num_centroids, s->dim, s->nbits, s->num_docs));
  if (s->doc_len_min < 0 || s->doc_len_max < s->doc_len_min || s->doc_len_max > NP_UNIQ_MAX || s->n_topics <= 0 ||
      s->dim * s->nbits / 8 > 128 || s->len_table_size < 0 || (s->len_table_size > 0 && !s->len_table)) {
    set_error("Invalid configuration: synth doc_len/n_topics/packed width out of range");
    return NP_ERR_INVALID_ARGUMENT;
  }
  int32_t table_max = 0;
  for (int i = 0; i < s->len_table_size; ++i) {
    if (s->len_table[i] < 0 || s->len_table[i] > NP_UNIQ_MAX) {
      set_error("Invalid configuration: synth len_table entry %d out of range", s->len_table[i]);
      return NP_ERR_INVALID_ARGUMENT;
    }
    table_max = std::max(table_max, s->len_table[i]);
  }
  DeviceGuard g(o.device);
  if (!g.ok) {
    set_error("hipSetDevice(%d) failed", o.device);
    return NP_ERR_DEVICE_UNAVAILABLE;
  }
  np_index* nix = new np_index();
  DeviceIndex* ix = nix;
  void* d_temp = nullptr;
  struct Cleanup {
    np_index* p;
    void** c;
    ~Cleanup() {
      (void)hipFree(*c);
      if (p) {
        destroy_device_index(p);
        delete p;
      }
    }
  } cleanup{nix, &d_temp};

  int64_t sb, se;
  shard_range(s->num_docs, o.shard_rank, o.shard_count, &sb, &se);
  ix->device = o.device;
  ix->opts = o;
  read_tuning_env(&ix->tune);
  ix->N_total = s->num_docs;
  ix->doc_begin = sb;
  ix->n_docs = se - sb;
  if (ix->n_docs >= ((int64_t)1 << 31)) {
    set_error("synth: a shard holds < 2^31 documents (got %lld); use more shards", (long long)ix->n_docs);
    return NP_ERR_INVALID_ARGUMENT;
  }
  ix->K = s->num_centroids;
  ix->KP = (ix->K + 63) / 64 * 64;
  ix->code_wide = ix->K > 65536 ? 1 : 0;
  ix->dim = ix->ldim = s->dim;        // the generator writes storage geometry directly (spec checked above)
  ix->nbits = ix->lnbits = s->nbits;
  ix->pd = ix->lpd = s->dim * s->nbits / 8;
  ix->max_doc_len = s->len_table_size > 0 ? table_max : s->doc_len_max;
  NP_TRY(upload_codec(ix, s->centroids, s->bucket_weights));

  SynthP p;
  p.b_len = mix64(s->seed + S_LEN);
  p.b_topic = mix64(s->seed + S_TOPIC);
  p.b_tok = mix64(s->seed + S_TOK);
  p.b_res = mix64(s->seed + S_RES);
  p.doc_begin = sb;
  p.n_docs = ix->n_docs;
  p.K = (uint32_t)ix->K;
  p.len_min = s->doc_len_min;
  p.len_span = s->doc_len_max - s->doc_len_min + 1;
  p.n_topics = s->n_topics;
  p.rand256 = s->rand256;
  p.pd = ix->pd;
  p.nw = (ix->pd + 7) / 8;

  // doc lengths -> offsets (inclusive scan shifted by one)
  NP_TRY(dev_alloc(&ix->d_doc_offsets, (size_t)ix->n_docs + 1, &ix->device_bytes));
  NP_HIP(hipMemset(ix->d_doc_offsets, 0, sizeof(int64_t)));
  if (ix->n_docs > 0) {
    int64_t* d_lens = nullptr;
    int32_t* d_tab = nullptr;
    NP_TRY(dev_alloc(&d_lens, (size_t)ix->n_docs, nullptr));
    if (s->len_table_size > 0) {
      hipError_t e0 = hipMalloc(&d_tab, (size_t)s->len_table_size * 4);
      if (e0 == hipSuccess) e0 = hipMemcpy(d_tab, s->len_table, (size_t)s->len_table_size * 4, hipMemcpyHostToDevice);
      if (e0 != hipSuccess) {
        (void)hipFree(d_lens);
        (void)hipFree(d_tab);
        set_error("synth: length table upload failed: %s", hipGetErrorString(e0));
        return NP_ERR_DEVICE_UNAVAILABLE;
      }
    }
    synth_lens_kernel<<<(unsigned)((ix->n_docs + 255) / 256), 256>>>(p, d_tab, s->len_table_size, d_lens);
    size_t tb = 0;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(nullptr, tb, d_lens, ix->d_doc_offsets + 1, (int)ix->n_docs);
    if (e == hipSuccess) e = hipMalloc(&d_temp, std::max<size_t>(tb, 16));
    if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(d_temp, tb, d_lens, ix->d_doc_offsets + 1, (int)ix->n_docs);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_lens);
    (void)hipFree(d_tab);
    (void)hipFree(d_temp);
    d_temp = nullptr;
    if (e != hipSuccess) {
      set_error("synth: doc offset scan failed: %s", hipGetErrorString(e));
      return NP_ERR_DEVICE_UNAVAILABLE;
    }
  }
  NP_HIP(hipMemcpy(&ix->T, ix->d_doc_offsets + ix->n_docs, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (ix->T >= ((int64_t)1 << 40)) {   // candidate records carry a 40-bit token offset
    set_error("synth: shard holds %lld tokens; a shard addresses < 2^40 tokens (use more shards)", (long long)ix->T);
    return NP_ERR_INVALID_ARGUMENT;
  }
  ix->n_emb_total = 0;  // filled below for unsharded corpora; sharded: avg-based estimate
  {
    uint8_t* cbuf = nullptr;
    NP_TRY(dev_alloc(&cbuf, (size_t)std::max<int64_t>(ix->T, 1) * ix->code_bytes(), &ix->device_bytes));
    ix->d_codes = cbuf;
  }
  NP_TRY(dev_alloc(&ix->d_residuals, (size_t)ix->T * ix->pd, &ix->device_bytes));
  if (ix->n_docs > 0)
    synth_tokens_kernel<<<(unsigned)((ix->n_docs + 3) / 4), 256>>>(p, ix->d_doc_offsets, ix->d_codes, ix->code_wide,
                                                                   ix->d_residuals);
  NP_HIP(hipGetLastError());
  NP_HIP(hipDeviceSynchronize());

  // whole-corpus token count: exact when unsharded or fixed-length, else extrapolated
  const bool fixed_len = s->len_table_size == 0 && s->doc_len_min == s->doc_len_max;
  if (o.shard_count == 1 || fixed_len)
    ix->n_emb_total = fixed_len ? s->num_docs * (int64_t)s->doc_len_min : ix->T;
  else
    ix->n_emb_total = ix->n_docs > 0 ? (int64_t)((double)ix->T / (double)ix->n_docs * (double)s->num_docs) : 0;
  ix->avg_doclen = s->num_docs > 0 ? (double)ix->n_emb_total / (double)s->num_docs : 0.0;
  NP_TRY(sort_tokens(ix));
  {
    int64_t* d_uoff = nullptr;
    int rc = build_unique_codes(ix, &d_uoff);
    if (rc == NP_OK) rc = build_ivf_from_ucodes(ix, d_uoff);   // index.rs:479-499
    (void)hipFree(d_uoff);
    NP_TRY(rc);
  }
  NP_TRY(build_ivf_split(ix));
  NP_TRY(build_inv_norm(ix));
  default_workspace(ix);
  cleanup.p = nullptr;
  *out = ix;
  return NP_OK;
}

}  // namespace np

using namespace np;

// =================================== C ABI ====================================================
extern "C" {

int np_hip_abi_version(void) { return NP_ABI_VERSION; }

int64_t np_hip_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(np_info);
    case 1: return (int64_t)sizeof(np_stats);
    case 2: return (int64_t)sizeof(np_search_params);
    case 3: return (int64_t)sizeof(np_open_opts);
    default: return -1;
  }
}

int np_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int usable = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

const char* np_hip_last_error(void) { return np::last_error(); }

int np_hip_index_open(const char* index_dir, const np_open_opts* opts, np_index** out) {
  clear_error();
  if (!out) {
    set_error("np_hip_index_open: out is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  HostIndex h;
  OpenTrace trace;
  NP_TRY(load_index_dir(index_dir, &h));
  trace.mark("parse + map the directory");
  DeviceIndex* ix = nullptr;
  NP_TRY(build_device_index(h, opts, &ix));
  *out = static_cast<np_index*>(ix);
  return NP_OK;
}

int np_hip_index_from_arrays(const np_index_arrays* a, const np_open_opts* opts, np_index** out) {
  clear_error();
  if (!out || !a) {
    set_error("np_hip_index_from_arrays: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  if (!a->centroids || !a->ivf_lengths || (a->num_docs > 0 && !a->doc_lengths)) {
    set_error("np_hip_index_from_arrays: centroids / ivf_lengths / doc_lengths are required");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (!a->bucket_weights) {  // codec.rs:428-431
    set_error("Codec error: bucket_weights required for decompression");
    return NP_ERR_CODEC;
  }
  HostIndex h;
  h.num_documents_total = a->num_documents_total;
  h.doc_begin = a->doc_begin;
  h.K = a->num_centroids;
  h.dim = a->dim;
  h.nbits = a->nbits;
  h.centroids = a->centroids;
  h.bucket_weights = a->bucket_weights;
  h.ivf = a->ivf;
  h.ivf_lengths = a->ivf_lengths;
  h.ivf_size = 0;
  for (int64_t c = 0; c < a->num_centroids; ++c) h.ivf_size += a->ivf_lengths[c];
  h.doc_lengths.assign(a->doc_lengths, a->doc_lengths + a->num_docs);
  HostChunk c;
  c.codes = a->codes;
  c.residuals = a->residuals;
  c.n_tokens = 0;
  for (int64_t d = 0; d < a->num_docs; ++d) c.n_tokens += a->doc_lengths[d];
  if (c.n_tokens > 0 && (!a->codes || !a->residuals)) {
    set_error("np_hip_index_from_arrays: codes / residuals are required");
    return NP_ERR_INVALID_ARGUMENT;
  }
  h.chunks.push_back(c);
  h.num_embeddings_total = c.n_tokens;
  h.avg_doclen = a->num_docs > 0 ? (double)c.n_tokens / (double)a->num_docs : 0.0;
  DeviceIndex* ix = nullptr;
  NP_TRY(build_device_index(h, opts, &ix));
  *out = static_cast<np_index*>(ix);
  return NP_OK;
}

int np_hip_index_synth(const np_synth_spec* spec, const np_open_opts* opts, np_index** out) {
  clear_error();
  if (!out) {
    set_error("np_hip_index_synth: out is NULL");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceIndex* ix = nullptr;
  NP_TRY(synth_build(spec, opts, &ix));
  *out = static_cast<np_index*>(ix);
  return NP_OK;
}

int64_t np_hip_index_ivf_size(const np_index* index) { return index ? index->ivf_size : 0; }

int np_hip_index_tune(np_index* ix, const char* name, int32_t value) {
  clear_error();
  if (!ix || !name) {
    set_error("np_hip_index_tune: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  if (!set_tuning(&ix->tune, name, value)) {
    set_error("np_hip_index_tune: unknown knob '%s'", name);
    return NP_ERR_INVALID_ARGUMENT;
  }
  return NP_OK;
}

int np_hip_index_export(const np_index* ix, int64_t* doc_lengths, int64_t* codes, uint8_t* residuals, int64_t* ivf,
                        int32_t* ivf_lengths) {
  clear_error();
  if (!ix) {
    set_error("np_hip_index_export: NULL index");
    return NP_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g(ix->device);
  if (doc_lengths) {
    std::vector<int64_t> off((size_t)ix->n_docs + 1);
    NP_HIP(hipMemcpy(off.data(), ix->d_doc_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t d = 0; d < ix->n_docs; ++d) doc_lengths[d] = off[d + 1] - off[d];
  }
  // codes / residuals are kept in per-document code order (np_internal.h): put the on-disk order back for the copy
  // (not concurrent with searches on this handle: export is a test / bench utility)
  struct Resort {
    const np_index* ix;
    bool active;
    ~Resort() {
      if (active) (void)permute_tokens(ix, 1);
    }
  } resort{ix, false};
  if ((codes || residuals) && ix->tok_sorted && ix->T > 0) {
    NP_TRY(permute_tokens(ix, 0));
    resort.active = true;
  }
  if (codes && ix->T > 0) {
    const int64_t PIECE = (int64_t)8 << 20;
    std::vector<uint32_t> tmp((size_t)std::min(PIECE, ix->T));
    for (int64_t s = 0; s < ix->T; s += PIECE) {
      int64_t n = std::min(PIECE, ix->T - s);
      if (ix->code_wide) {
        NP_HIP(hipMemcpy(tmp.data(), (const uint32_t*)ix->d_codes + s, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) codes[s + i] = (int64_t)tmp[i];
      } else {
        uint16_t* t16 = reinterpret_cast<uint16_t*>(tmp.data());
        NP_HIP(hipMemcpy(t16, (const uint16_t*)ix->d_codes + s, (size_t)n * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) codes[s + i] = (int64_t)t16[i];
      }
    }
  }
  if (residuals && ix->T > 0 && ix->pd == ix->lpd) {
    NP_HIP(hipMemcpy(residuals, ix->d_residuals, (size_t)ix->T * ix->pd, hipMemcpyDeviceToHost));
  } else if (residuals && ix->T > 0) {   // storage rows -> file rows
    uint8_t* d_rows = nullptr;
    NP_HIP(hipMalloc(&d_rows, (size_t)ix->T * ix->lpd));
    unpack_rows_kernel<<<(unsigned)((ix->T * ix->lpd + 255) / 256), 256>>>(ix->d_residuals, ix->T, ix->lpd, ix->pd,
                                                                          ix->lnbits != ix->nbits, d_rows);
    hipError_t e = hipMemcpy(residuals, d_rows, (size_t)ix->T * ix->lpd, hipMemcpyDeviceToHost);
    (void)hipFree(d_rows);
    NP_HIP(e);
  }
  if (ivf && ix->ivf_size > 0) {
    std::vector<uint32_t> tmp((size_t)ix->ivf_size);
    NP_HIP(hipMemcpy(tmp.data(), ix->d_ivf, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < ix->ivf_size; ++i) ivf[i] = (int64_t)tmp[i] + ix->doc_begin;
  }
  if (ivf_lengths) {
    std::vector<int64_t> off((size_t)ix->K + 1);
    NP_HIP(hipMemcpy(off.data(), ix->d_ivf_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < ix->K; ++c) ivf_lengths[c] = (int32_t)(off[c + 1] - off[c]);
  }
  return NP_OK;
}

void np_hip_index_close(np_index* index) {
  if (!index) return;
  destroy_device_index(index);
  delete index;
}

int np_hip_index_info(const np_index* ix, np_info* out) {
  clear_error();
  if (!ix || !out) {
    set_error("np_hip_index_info: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  memset(out, 0, sizeof *out);
  out->num_documents = ix->N_total;
  out->num_embeddings = ix->n_emb_total;
  out->num_partitions = ix->K;
  out->embedding_dim = ix->ldim;
  out->nbits = ix->lnbits;
  out->avg_doclen = ix->avg_doclen;
  out->shard_doc_begin = ix->doc_begin;
  out->shard_doc_end = ix->doc_begin + ix->n_docs;
  out->shard_embeddings = ix->T;
  out->device_bytes = (int64_t)ix->device_bytes;
  out->device = ix->device;
  out->abi_version = NP_ABI_VERSION;
  out->workspace_bytes = ix->ws_budget.load(std::memory_order_relaxed);
  return NP_OK;
}

// Host-only: parse and validate an index directory exactly like np_hip_index_open does (MmapIndex::load,
// index.rs:1026-1139; file formats mmap.rs:659-749) without touching a device.  Lets a service (or a CPU test)
// check an index before it claims a GPU.  out->device = -1, no shard fields.
int np_hip_index_probe_dir(const char* index_dir, np_info* out) {
  clear_error();
  if (!out) {
    set_error("np_hip_index_probe_dir: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  HostIndex h;
  NP_TRY(load_index_dir(index_dir, &h));
  NP_TRY(check_geometry(h.K, h.dim, h.nbits, h.num_documents_total));
  // every code must name a centroid and every posting a document (the same checks build_device_index makes)
  for (const HostChunk& c : h.chunks)
    for (int64_t t = 0; t < c.n_tokens; ++t)
      if (c.codes[t] < 0 || c.codes[t] >= h.K) {
        set_error("Index load failed: code %lld is outside [0, %lld)", (long long)c.codes[t], (long long)h.K);
        return NP_ERR_INDEX_LOAD;
      }
  int64_t ivf_sum = 0;
  for (int64_t i = 0; i < h.K; ++i) ivf_sum += h.ivf_lengths[i];
  for (int64_t i = 0; i < ivf_sum; ++i)
    if (h.ivf[i] < 0 || h.ivf[i] >= h.num_documents_total) {
      set_error("Index load failed: ivf entry %lld is outside [0, %lld)", (long long)h.ivf[i],
                (long long)h.num_documents_total);
      return NP_ERR_INDEX_LOAD;
    }
  memset(out, 0, sizeof *out);
  int64_t tokens = 0;
  for (const HostChunk& c : h.chunks) tokens += c.n_tokens;
  out->num_documents = h.num_documents_total;
  out->num_embeddings = h.num_embeddings_total;
  out->num_partitions = h.K;
  out->embedding_dim = h.dim;
  out->nbits = h.nbits;
  out->avg_doclen = h.avg_doclen;
  out->shard_doc_begin = 0;
  out->shard_doc_end = h.num_documents_total;
  out->shard_embeddings = tokens;
  out->device_bytes = 0;
  out->device = -1;
  out->abi_version = NP_ABI_VERSION;
  return NP_OK;
}

}  // extern "C"
