// np_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the PLAID search path.
//
// Stage map (reference: next-plaid/src/search.rs:327-516, codec.rs:423-470, maxsim.rs:270-294):
//   prep_queries_kernel   pads/transposes the query batch                      (host glue)
//   qc_gemm_kernel        S1  Q.C^T on exact-f32 MFMA 32x32x2, k-ordered FMA chain == search.rs:345;
//                             writes QCT[b][c][q] (one 128-B line per centroid) + per-32-centroid maxima
//   probe_kernel          S2  per-token top-nprobe (radix select on group maxima, then on the
//                             surviving groups) + threshold (search.rs:388-425)
//   mark/count/compact    S3  posting-list union as a per-query doc bitmap -> ascending unique ids
//                             (index.rs:1142-1156: concat + sort_unstable + dedup)
//   approx_kernel         S4  sum_q max_tok QC[q, code[tok]]   (search.rs:305-324)
//   select_kernel         S5  stable top n_sel by approximate score (search.rs:460-469)
//   exact_f32/bf16_kernel S6  residual unpack + centroid add + L2 normalise straight into MFMA A
//                             fragments, Q.D^T on MFMA, row-max / sum (codec.rs:423-470, maxsim.rs:270-294)
//   topk_kernel           S7  stable top-k by exact score (search.rs:496-515)
//   select_cut / merge    document-sharded exchange (new)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace np {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NP_NEG_INF (-__builtin_huge_valf())
#define NP_MAX_QT 8      // query tiles of 32 tokens (LQP <= 256); exact kernels are instantiated for 1, 2 and 8

// Orderable key of search.rs:110-117's comparator: finite values keep f32::total_cmp order in
// [0x00800000, 0xFF7FFFFF]; every non-finite value maps to 0 (all Equal, below any finite).
__device__ __forceinline__ uint32_t okey(float x) {
  uint32_t b = __float_as_uint(x);
  uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((b & 0x7F800000u) == 0x7F800000u) ? 0u : k;
}
__device__ __forceinline__ float unkey(uint32_t k) {  // inverse for k != 0
  uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(b);
}
__device__ __forceinline__ bool finitef(float x) { return (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u; }
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int mfma_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

// Maximum of a non-negative int over the wave, wave-uniform result.  DPP row shifts / row broadcasts on the VALU (the
// GFX9 reduction idiom) instead of six ds_bpermute round trips through the LDS pipe per __shfl_xor butterfly.
__device__ __forceinline__ int wave_max_nonneg(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of each row = row maximum
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3 -> lane 63 = maximum
  return __builtin_amdgcn_readlane(v, 63);
}
// value of lane J (0 or 1) of this lane's group of G lanes (G a power of two): a quad permute for G <= 4
template <int G, int J>
__device__ __forceinline__ int group_lane(int v, int lane) {
  if constexpr (G == 2) return __builtin_amdgcn_mov_dpp(v, J == 0 ? 0xA0 : 0xF5, 0xf, 0xf, true);        // [0,0,2,2] / [1,1,3,3]
  else if constexpr (G == 4) return __builtin_amdgcn_mov_dpp(v, J == 0 ? 0x00 : 0x55, 0xf, 0xf, true);   // [0,0,0,0] / [1,1,1,1]
  else return __shfl(v, (lane & ~(G - 1)) + J);
}

struct Counters {  // per-call work counters (np_stats)
  unsigned long long n_cells, n_ivf_ids, n_candidates, n_cand_tokens, n_exact_docs, n_exact_tokens, n_cand_codes;
  unsigned long long n_rounds;     // candidate-pool rounds the slice needed (max over slices)
  unsigned long long n_survivors;  // candidates that passed the S4 upper-bound filter (= n_candidates when it is off)
  unsigned long long n_cand_dcodes;  // distinct (document, code) pairs of the candidates (the hot level scans these)
  unsigned long long n_level2;       // two-level filter: documents that took the exact u8 bound (S1 + S2)
  unsigned long long n_level0;       // zeroth level (gain_sweep_kernel): candidates it handed to the filter (0 = level not run)
};

// One launch instead of a dozen hipMemsetAsync calls per batch (each ~3.6 us on the stream: 50 us per batch at 1 M
// documents): fills up to NP_CLEAR_MAX small word regions (counters, per-query bitmaps, histogram, hand-out slots).
#define NP_CLEAR_MAX 24
struct ClearList {
  uint32_t* p[NP_CLEAR_MAX];
  uint32_t words[NP_CLEAR_MAX];
  uint32_t fill[NP_CLEAR_MAX];
  int n;
};
__global__ void __launch_bounds__(256) clear_regions_kernel(ClearList cl) {
  for (int i = 0; i < cl.n; ++i) {
    uint32_t* p = cl.p[i];
    const uint32_t v = cl.fill[i];
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < cl.words[i]; w += gridDim.x * 256) p[w] = v;
  }
}

// caller rows [n][ldim] -> storage rows [n][dim], zero padded (np_internal.h storage_dim): the padded dims add exact zeros
// to every dot product
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ src, int64_t n, int ldim, int dim,
                                                       float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * dim) return;
  const int64_t r = i / dim;
  const int k = (int)(i - r * dim);
  dst[i] = k < ldim ? src[r * ldim + k] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// prep: Qt[b][k][q] f32 (k-major, zero padded to LQP) and Qb[b][q][k] bf16
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prep_queries_kernel(const float* __restrict__ q, const int32_t* __restrict__ qoff,
                                                           int dim, int LQP, float* __restrict__ Qt,
                                                           __bf16* __restrict__ Qb, __bf16* __restrict__ Qb_lo,
                                                           float cmax, float* __restrict__ qinv,
                                                           uint32_t* __restrict__ qflag) {
  __shared__ float s_m[4];
  __shared__ int s_bad;
  const int b = blockIdx.x;
  const int t0 = qoff[b], Lq = qoff[b + 1] - t0;
  const int n = dim * LQP;
  if (threadIdx.x == 0) s_bad = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    {  // k-major write, coalesced over q
      int k = i / LQP, qq = i - k * LQP;
      float v = (qq < Lq) ? q[(int64_t)(t0 + qq) * dim + k] : 0.0f;
      Qt[(int64_t)b * n + i] = v;
    }
    {  // row-major bf16
      int qq = i / dim, k = i - qq * dim;
      float v = (qq < Lq) ? q[(int64_t)(t0 + qq) * dim + k] : 0.0f;
      const __bf16 hi = (__bf16)v;
      Qb[(int64_t)b * n + i] = hi;
      Qb_lo[(int64_t)b * n + i] = (__bf16)(v - (float)hi);   // q = hi + lo to ~2^-17 relative
    }
  }
  // Scale of this query's u8 score table (S4 upper-bound filter): every finite Q.C^T value obeys
  // |QC[q,c]| <= (1 + 128 * 2^-24) * ||q|| * ||c|| <= s = 1.001 * max_q ||q|| * cmax (Cauchy-Schwarz; the f32 norms and
  // the k-ordered FMA chain are each within ~1e-5 relative).  qinv = 1 / s; a query with a non-finite value is flagged
  // and keeps the unfiltered path.
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = 0.f;
  int bad = 0;
  for (int qq = wave; qq < Lq; qq += 4) {
    float ss = 0.f;
    for (int k = lane; k < dim; k += 64) {
      const float v = q[(int64_t)(t0 + qq) * dim + k];
      ss = fmaf(v, v, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (!finitef(ss)) bad = 1;
    else mx = fmaxf(mx, ss);
  }
  if (lane == 0) {
    s_m[wave] = mx;
    if (bad) atomicOr(&s_bad, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0 && qinv) {
    const float m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    float sc = sqrtf(m) * cmax * 1.001f;
    int isbad = s_bad;
    if (!finitef(sc) || !(sqrtf(m) < 1e12f)) isbad = 1;   // huge norms: products could overflow (S6 fast epilogue)
    if (!(sc > 0.f)) sc = 1.0f;   // an all-zero query: every score is 0
    qinv[b] = isbad ? 0.f : 1.0f / sc;
    qflag[b] = isbad ? 1u : 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// S1  Q.C^T  (exact f32 MFMA).  One wave = 64 centroids (two 32-row A fragments held in
// registers for the whole kernel) x every 32-token query tile of the batch.
// D[c][q]: lane holds q = lane&31 and centroid rows mfma_row(r, lane>>5).
// The k loop feeds k = 2s + (lane>>5) in ascending s, so each output is the k-ordered FMA chain.
// ---------------------------------------------------------------------------------------------
template <int DIM, int CPW>   // CPW = 32-centroid A fragments per wave
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CPW == 1 ? 3 : 1))) qc_gemm_kernel(const float* __restrict__ C, int64_t K, int64_t KP,
                                                      const float* __restrict__ Qt, int B, int LQP,
                                                      float* __restrict__ QCT, uint32_t* __restrict__ gmax,
                                                      uint8_t* __restrict__ QCU, int RB /* u8 row bytes: power of two >= LQP */,
                                                      const float* __restrict__ qinv, const int32_t* __restrict__ qoff) {
  // The block's 4 waves walk the same sequence of 32-token query tiles ([DIM][32] f32, k-major).  Tile t+1 is
  // copied global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers) while tile t feeds
  // the MFMAs as conflict-free ds_read B operands; tile t-1's epilogue (QCT stores, group maxima) is issued
  // at the top of iteration t, so by the barrier that closes the iteration neither the DMA nor the stores
  // are still outstanding and s_waitcnt vmcnt(0) costs nothing.
  constexpr int NV = DIM / 32;  // 1-KiB DMA pieces per wave per tile
  __shared__ float sQ[2][DIM * 32];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * CPW);
  const bool active = c0 < KP;
  float a[CPW][DIM / 2];
#pragma unroll
  for (int f = 0; f < CPW; ++f) {
    const int64_t r0 = c0 + 32 * f + li;
#pragma unroll
    for (int m = 0; m < DIM / 4; ++m) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 < K) v0 = *reinterpret_cast<const float4*>(C + r0 * DIM + 4 * m);
      a[f][2 * m] = kk ? v0.y : v0.x;
      a[f][2 * m + 1] = kk ? v0.w : v0.z;
    }
  }
  const int nqt = LQP >> 5;
  const int ntiles = B * nqt;
  const int64_t G = KP >> 5;
  auto dma_tile = [&](int tile, int buf) {
    const int b = tile / nqt, qt = tile - b * nqt;
    const float* src = Qt + (int64_t)b * DIM * LQP + qt * 32;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i4 = j * 256 + tid;   // float4 index inside the tile; the wave's 64 lanes fill 1 KiB of LDS linearly
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + (int64_t)(i4 >> 3) * LQP + (i4 & 7) * 4),
          (__attribute__((address_space(3))) void*)(&sQ[buf][(j * 256 + wave * 64) * 4]), 16, 0, 0);
    }
  };
  // Epilogue through LDS.  A lane holds one token (li) of 16 centroid rows, so storing straight from the accumulators is
  // 16 dword + 16 byte store instructions per tile, each a full 64-lane address pass of the CU's single vector-memory
  // port.  The tile is transposed in LDS instead (16 ds_write_b32 + 16 ds_write_b8, wave-private) and leaves as 4 dwordx4
  // stores of 8 rows each plus ONE dwordx4 store of the 32 u8 rows: 6 vector-memory instructions per tile instead of 33
  // (S1 0.48 -> 0.46 ms).  What remains (cycle-stamped build): the ~250 VALU / LDS instructions of the epilogue issue at
  // one per 25-30 cycles while the sibling wave's back-to-back f32 MFMA chain holds the SIMD -- per SIMD the kernel is
  // MFMA time plus epilogue time; placing element r of tile t-1 after MFMA 4r+3 of tile t (the wave's own MFMA shadow) did
  // not change that (0.48 ms).
  // Round 6: the f32 tile goes through LDS in two halves of 16 centroid rows (one more wave barrier per fragment) and the kernel is
  // compiled for three waves per SIMD: at DIM = 128 it stood at 172 registers and 54 KB of LDS per workgroup -- both just past
  // what three workgroups per CU allow (168 / 53.3 KB).  With a third wave to run its MFMA chain while two are in their
  // epilogues: S1 0.370 -> 0.366 ms at K = 2^16, 2.95 - 3.05 -> 2.77 ms at K = 2^19 (63 % of the f32 MFMA peak).
  constexpr int TS = 36;   // f32 tile row stride in words: 16-B aligned rows, halves land on different banks
  __shared__ __attribute__((aligned(16))) float sT[4][16 * TS];   // half a tile (16 centroid rows) at a time
  __shared__ __attribute__((aligned(16))) uint8_t sU[4][32 * 32];
  auto epilogue = [&](const f32x16 (&acc)[CPW], int tile) {
    // (Round 5, measured and removed: s_setprio 2 for the epilogue -- VALU issue between the waves of a SIMD is arbitrated by
    // priority, then age -- left S1 where it was: 0.43 vs 0.44 ms at K = 2^16, 3.25 vs 3.37 ms at K = 2^19.)
    const int b = nqt == 1 ? tile : tile / nqt, qt = tile - b * nqt;
    const int h = lane >> 5, rr = (lane & 31) >> 3, c4 = lane & 7;   // store role: half, row in the group of 4, 16-B chunk
#pragma unroll
    for (int f = 0; f < CPW; ++f) {
      // group key maximum.  Fragments wholly inside [0, K) whose 16 values are finite in every lane (their sum is) take
      // the key of the float maximum: okey() is strictly increasing on finite values and the accumulators start at +0,
      // so no -0 can appear.  Anything else (rows past K, a non-finite value) takes the key of every element.
      float vsum = 0.f, vmax = acc[f][0];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        vsum += acc[f][r];
        vmax = fmaxf(vmax, acc[f][r]);
      }
      uint32_t k0;
      if (c0 + 32 * f + 32 <= K && __ballot(!finitef(vsum)) == 0ull) {
        k0 = okey(vmax);
      } else {
        k0 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) k0 = max(k0, (c0 + 32 * f + mfma_row(r, kk) < K) ? okey(acc[f][r]) : 0u);
      }
      k0 = max(k0, (uint32_t)__shfl_xor((int)k0, 32));
      if (kk == 0) gmax[((int64_t)b * G + ((c0 >> 5) + f)) * LQP + qt * 32 + li] = k0;
      if (QCU) {
        // u8 UPPER bound of every score for the S4 filter (round 5: the table spans the POSITIVE scores only, twice the
        // resolution): u = floor(max(x, 0) / s * 254) + 1 in [1, 254] (x <= s / 1.001), monotone in x, so max over a
        // document's codes commutes with it; u = 1 is the clipped bottom entry (x < s / 254: the score may be anything down
        // to -s, which the LOWER bounds account for -- approx_ub_kernel); 0 marks the padding tokens q >= Lq.  One RB-byte
        // row per centroid (bytes LQP .. RB-1 are zeroed by the host).
        const float inv = qinv[b];
        const bool qv = qt * 32 + li < qoff[b + 1] - qoff[b];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float hh = fmaxf(acc[f][r] * inv, 0.f) * 254.0f;    // < 253.8: u <= 254 without a clamp (NaN -> 0: flagged queries
          const uint32_t u = qv ? (uint32_t)hh + 1u : 0u;            // never reach the filter)
          sU[wave][mfma_row(r, kk) * 32 + li] = (uint8_t)u;
        }
      }
      float* outb = QCT + ((int64_t)b * KP + c0 + 32 * f) * LQP + qt * 32 + 4 * c4;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {   // rows 0..15 (registers 0..7), then rows 16..31, through the same half tile
#pragma unroll
        for (int r = 8 * hp; r < 8 * hp + 8; ++r) sT[wave][(mfma_row(r, kk) - 16 * hp) * TS + li] = acc[f][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the wave's own LDS writes are visible to its other lanes
#pragma unroll
        for (int g = 2 * hp; g < 2 * hp + 2; ++g) {
          const int row = 8 * g + 4 * h + rr;
          const float4 v = *reinterpret_cast<const float4*>(&sT[wave][(row - 16 * hp) * TS + 4 * c4]);
          *reinterpret_cast<float4*>(outb + (int64_t)row * LQP) = v;
        }
        __builtin_amdgcn_wave_barrier();   // the second half / the next fragment overwrites sT
      }
      if (QCU) {
        const int row = lane >> 1, half = lane & 1;   // two lanes per 32-B row piece
        const uint4 v = *reinterpret_cast<const uint4*>(&sU[wave][row * 32 + 16 * half]);
        *reinterpret_cast<uint4*>(QCU + ((int64_t)b * KP + c0 + 32 * f + row) * RB + qt * 32 + 16 * half) = v;
      }
      __builtin_amdgcn_wave_barrier();   // the next fragment / tile overwrites sT and sU
    }
  };
  if (ntiles > 0) dma_tile(0, 0);
  __syncthreads();
  f32x16 prev[CPW];
#pragma unroll
  for (int f = 0; f < CPW; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[f][r] = 0.f;
  for (int tile = 0; tile < ntiles; ++tile) {
    if (tile + 1 < ntiles) dma_tile(tile + 1, (tile + 1) & 1);
    if (active) {
      if (tile > 0) epilogue(prev, tile - 1);
      const float* qb = &sQ[tile & 1][kk * 32 + li];   // [2s+kk][q]
      f32x16 acc[CPW];
#pragma unroll
      for (int f = 0; f < CPW; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
#pragma unroll
      for (int s = 0; s < DIM / 2; ++s) {  // fully unrolled: the A fragments must stay in registers
        const float bq = qb[s * 64];
#pragma unroll
        for (int f = 0; f < CPW; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f][s], bq, acc[f], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < CPW; ++f) prev[f] = acc[f];
    }
    __syncthreads();
  }
  if (active && ntiles > 0) epilogue(prev, ntiles - 1);
}

// ---------------------------------------------------------------------------------------------
// S1, split-bf16 form (round 4; OPT-IN: precision >= 1 and K > centroid_batch_size, the crate's large-K regime).
// At K = 2^19 (what kmeans.rs:303-309 picks for 10 M x 300-token documents) Q.C^T is 275 GFLOP per batch of 64 queries:
// 1.75 ms at the exact-f32 MFMA peak, 3.5 ms measured -- the largest stage of that regime.  Here both operands are split
// into bf16 hi + lo (x = hi + lo to ~2^-17 relative) and the product is hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16,
// f32 accumulation: three MFMAs at 16x the f32 rate, |error| <~ 2^-16 |q||c| per score (the dropped lo.lo term and the
// bf16 rounding of lo) -- the arithmetic of the precision-2 MaxSim.  The values are NOT the k-ordered f32 FMA chain any
// more, so S1-S5 are no longer bit-equal to the oracle in this mode: probed cells / candidates can differ at near-ties
// closer than ~1e-5, rankings are checked with assert_ranking_close like the exact stage (tests/test_gpu_large_k.py).
// precision 0 never takes this path.  Structure as qc_gemm_kernel: a wave keeps one 32-centroid A fragment (hi and lo, 64
// VGPRs) for the whole kernel and walks every 32-token query tile; tiles arrive by global_load_lds into a double buffer,
// laid out [hi | lo][k-step][k-half][token] in 16-byte pieces so that the B operand reads are conflict-free.
// ---------------------------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(256) qc_gemm_b3_kernel(const float* __restrict__ C, int64_t K, int64_t KP,
                                                         const __bf16* __restrict__ Qb, const __bf16* __restrict__ Qbl, int B,
                                                         int LQP, float* __restrict__ QCT, uint32_t* __restrict__ gmax,
                                                         uint8_t* __restrict__ QCU, int RB, const float* __restrict__ qinv,
                                                         const int32_t* __restrict__ qoff) {
  constexpr int NS = DIM / 16;                 // k-steps of 16
  constexpr int TILE_B = DIM * 32 * 2;         // bytes of one bf16 tile (hi or lo)
  constexpr int NV = 2 * TILE_B / (256 * 16);  // 16-byte DMA pieces per thread per tile
  __shared__ __attribute__((aligned(16))) char sQ[2][2 * TILE_B];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const bool active = c0 < KP;
  bf16x8 ah[NS], al[NS];
  {
    const int64_t r0 = c0 + li;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float v[8];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 < K) f = *reinterpret_cast<const float4*>(C + r0 * DIM + 16 * s + 8 * kk + 4 * m);
        v[4 * m] = f.x; v[4 * m + 1] = f.y; v[4 * m + 2] = f.z; v[4 * m + 3] = f.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        ah[s][e] = h;
        al[s][e] = (__bf16)(v[e] - (float)h);
      }
    }
  }
  const int nqt = LQP >> 5;
  const int ntiles = B * nqt;
  const int64_t G = KP >> 5;
  auto dma_tile = [&](int tile, int buf) {
    const int b = tile / nqt, qt = tile - b * nqt;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int p = j * 256 + tid;                    // 16-byte piece of the buffer: [half][s][kk][token]
      const int half = p / (TILE_B / 16), pp = p - half * (TILE_B / 16);
      const int s = pp >> 6, k2 = (pp >> 5) & 1, tok = pp & 31;
      const __bf16* src = (half ? Qbl : Qb) + ((int64_t)b * LQP + qt * 32 + tok) * DIM + 16 * s + 8 * k2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(&sQ[buf][(j * 256 + wave * 64) * 16]), 16, 0, 0);
    }
  };
  constexpr int TS = 36;   // f32 tile row stride in words (see qc_gemm_kernel)
  __shared__ __attribute__((aligned(16))) float sT[4][32 * TS];
  __shared__ __attribute__((aligned(16))) uint8_t sU[4][32 * 32];
  auto epilogue = [&](const f32x16& acc, int tile) {
    const int b = nqt == 1 ? tile : tile / nqt, qt = tile - b * nqt;
    const int h = lane >> 5, rr = (lane & 31) >> 3, c4 = lane & 7;
    float vsum = 0.f, vmax = acc[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sT[wave][mfma_row(r, kk) * TS + li] = acc[r];
      vsum += acc[r];
      vmax = fmaxf(vmax, acc[r]);
    }
    uint32_t k0;
    if (c0 + 32 <= K && __ballot(!finitef(vsum)) == 0ull) {
      k0 = okey(vmax);
    } else {
      k0 = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) k0 = max(k0, (c0 + mfma_row(r, kk) < K) ? okey(acc[r]) : 0u);
    }
    k0 = max(k0, (uint32_t)__shfl_xor((int)k0, 32));
    if (kk == 0) gmax[((int64_t)b * G + (c0 >> 5)) * LQP + qt * 32 + li] = k0;
    if (QCU) {
      const float inv = qinv[b];
      const bool qv = qt * 32 + li < qoff[b + 1] - qoff[b];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // the split product can exceed the Cauchy-Schwarz bound of the f32 chain by its own error: clamp to the table's range
        const float hh = fminf(fmaxf(acc[r] * inv, 0.f) * 254.0f, 254.5f);
        const uint32_t u = qv ? (uint32_t)hh + 1u : 0u;
        sU[wave][mfma_row(r, kk) * 32 + li] = (uint8_t)u;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* outb = QCT + ((int64_t)b * KP + c0) * LQP + qt * 32 + 4 * c4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 8 * g + 4 * h + rr;
      const float4 v = *reinterpret_cast<const float4*>(&sT[wave][row * TS + 4 * c4]);
      *reinterpret_cast<float4*>(outb + (int64_t)row * LQP) = v;
    }
    if (QCU) {
      const int row = lane >> 1, half = lane & 1;
      const uint4 v = *reinterpret_cast<const uint4*>(&sU[wave][row * 32 + 16 * half]);
      *reinterpret_cast<uint4*>(QCU + ((int64_t)b * KP + c0 + row) * RB + qt * 32 + 16 * half) = v;
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (ntiles > 0) dma_tile(0, 0);
  __syncthreads();
  f32x16 prev;
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  for (int tile = 0; tile < ntiles; ++tile) {
    if (tile + 1 < ntiles) dma_tile(tile + 1, (tile + 1) & 1);
    if (active) {
      if (tile > 0) epilogue(prev, tile - 1);
      const char* qb = &sQ[tile & 1][0];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(qb + 16 * ((s * 2 + kk) * 32 + li));
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(qb + TILE_B + 16 * ((s * 2 + kk) * 32 + li));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc, 0, 0, 0);
      }
      prev = acc;
    }
    __syncthreads();
  }
  if (active && ntiles > 0) epilogue(prev, ntiles - 1);
}

// ---------------------------------------------------------------------------------------------
// S2 helpers: block-level radix select for QW query tokens at once.
// Thread (r = tid / QW, q = tid % QW); `enumerate(cb)` calls cb(key) for this thread's items of
// token q.  On return s_prefix[q] = key of the want-th largest item, s_rem[q] = how many items
// equal to that key belong to the top `want`.
// ---------------------------------------------------------------------------------------------
template <int QW, class Enum>
__device__ __forceinline__ void radix_select(Enum&& enumerate, uint32_t want, uint32_t* hist /*[256*QW]*/,
                                             uint32_t* part /*[256]*/, uint32_t* s_prefix, uint32_t* s_rem, int tid) {
  constexpr int NR = 256 / QW;      // thread rows
  constexpr int BPR = 256 / NR;     // histogram bins summed per row-thread (= QW)
  const int q = tid % QW, r = tid / QW;
  if (r == 0) {
    s_prefix[q] = 0;
    s_rem[q] = want;
  }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 256 * QW; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t pre = s_prefix[q];
    enumerate([&](uint32_t key) {
      if (pass == 0 || (key >> (shift + 8)) == pre) atomicAdd(&hist[((key >> shift) & 255u) * QW + q], 1u);
    });
    __syncthreads();
    uint32_t sum = 0;
    for (int i = 0; i < BPR; ++i) sum += hist[(r * BPR + i) * QW + q];
    part[r * QW + q] = sum;
    __syncthreads();
    if (r == 0) {
      const uint32_t rem = s_rem[q];
      uint32_t cum = 0;
      int R = NR - 1;
      for (; R > 0; --R) {
        const uint32_t pr = part[R * QW + q];
        if (cum + pr >= rem) break;
        cum += pr;
      }
      int bin = R * BPR + BPR - 1;
      for (; bin > R * BPR; --bin) {
        const uint32_t h = hist[bin * QW + q];
        if (cum + h >= rem) break;
        cum += h;
      }
      s_prefix[q] = (pre << 8) | (uint32_t)bin;
      s_rem[q] = rem - cum;
    }
    __syncthreads();
  }
}

struct ProbeP {
  const float* QCT;        // [B][KP][LQP]
  const uint32_t* gmax;    // [B][KP/32][LQP]  (masked by `elig` when a subset is given)
  const int32_t* qoff;     // [B+1]
  int64_t K, KP;
  int LQP;
  int nprobe;              // params.n_ivf_probe
  const int32_t* nprobe_dev;   // effective nprobe with a subset (search.rs:370-382), else NULL
  const uint32_t* elig;    // eligible-centroid bitmap [KP/32] (search.rs:350-364), else NULL
  const int32_t* n_elig;   // popcount of elig, else NULL
  int has_thr;
  float thr;
  int lds_gm;              // probe_mark_kernel<4> was launched with KP/32 * 4 * 4 bytes of dynamic LDS
  int64_t slab;            // > 0: batched-probe semantics (search.rs:140-254) with this centroid_batch_size
  uint32_t* cellbits;      // [B][KP/32] zeroed
  uint32_t* tauq;          // [B][LQP] per token: okey of its n_probe-th best centroid (0 = everything)
  uint32_t* cells_tmp;     // [B][KP]
  uint32_t* cells;         // [B][KP]
  int32_t* n_cells;        // [B]
  Counters* ctr;
};

#define NP_PROBE_CAPG 64   // surviving 32-centroid groups per token whose keys fit one wave's registers

// Per-token tail of the probe: among the elements of the surviving groups find the n_probe best and
// mark them.  One WAVE per token; element keys (+1, 0 = absent) live in 32 registers per lane and the
// n_probe-th largest is found by a 32-step bitwise search with ballot/popcount counting.
// Number of (slot, lane) keys satisfying `pred`, over the whole wave.  Each lane counts its own slots with VALU
// compares (<= 32, six bits), then six ballots weigh the bits: one ballot -> s_bcnt1 round trip per SLOT (32 per
// counting step, each ~100 cycles of VALU -> SALU latency with one wave per SIMD) made this search 75 % of
// probe_mark_kernel's time.
template <class Pred>
__device__ __forceinline__ uint32_t wave_count(int nslots, const uint32_t (&keys)[32], Pred&& pred) {
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 32; ++j) mine += (j < nslots && pred(keys[j])) ? 1u : 0u;
  uint32_t cnt = 0;
#pragma unroll
  for (int b = 0; b < 6; ++b) cnt += (uint32_t)__popcll(__ballot((mine >> b) & 1u)) << b;
  return cnt;
}

__device__ __forceinline__ void wave_select_mark(int nslots, uint32_t n_probe, const uint32_t (&keys)[32],
                                                 uint32_t& tau, uint32_t& rem) {
  uint32_t prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t trial = prefix | (1u << bit);
    if (wave_count(nslots, keys, [&](uint32_t k) { return k >= trial; }) >= n_probe) prefix = trial;
  }
  const uint32_t gt = wave_count(nslots, keys, [&](uint32_t k) { return k > prefix; });
  tau = prefix;
  rem = n_probe > gt ? n_probe - gt : 0u;
}

// QW = query tokens per probe block, grid = (LQP / QW, B).  4 (one token per wave, two blocks per CU) with the group maxima
// in LDS; 8 when they are re-read from memory (K > 65536: measured 0.36 vs 0.51 ms at K = 2^18).
template <int QW>
__global__ void __launch_bounds__(256) probe_mark_kernel(ProbeP p) {
  __shared__ uint32_t hist[256 * QW];
  __shared__ uint32_t part[256];
  __shared__ uint32_t s_prefix[QW], s_rem[QW], s_taug[QW], s_gcnt[QW];
  __shared__ uint32_t s_glist[QW * NP_PROBE_CAPG];
  extern __shared__ uint32_t s_gm[];   // [G][QW] group maxima of the block's tokens when the launch provides the space
  constexpr int NR = 256 / QW;
  const bool lds_gm = p.lds_gm != 0;
  const int b = blockIdx.y, qc = blockIdx.x, tid = threadIdx.x, q = tid % QW, r = tid / QW;
  const int wave = tid >> 6, lane = tid & 63;
  const int Lq = p.qoff[b + 1] - p.qoff[b];
  const int64_t G = p.KP >> 5;
  const int LQP = p.LQP;
  const float* QCT = p.QCT + (int64_t)b * p.KP * LQP;
  const uint32_t* gm = p.gmax + (int64_t)b * G * LQP;
  uint32_t* bits = p.cellbits + (int64_t)b * G;
  const int64_t pool = p.elig ? (int64_t)*p.n_elig : p.K;
  const int64_t eff = p.nprobe_dev ? (int64_t)*p.nprobe_dev : (int64_t)p.nprobe;
  const uint32_t n_probe = (uint32_t)min(eff, pool);  // search.rs:405
  const bool take_all = pool <= (int64_t)n_probe;
  uint32_t* tauq = p.tauq + (int64_t)b * LQP;

  if (n_probe > 0 && take_all && Lq > 0) {
    if (qc != 0) return;   // one block marks the whole pool
    // every pooled centroid is selected by every token (search.rs:406: len <= n_probe)
    for (int64_t w = tid; w < G; w += 256) {
      uint32_t m = p.elig ? p.elig[w] : 0xFFFFFFFFu;
      const int64_t c0 = w * 32;
      if (c0 + 32 > p.K) m &= (c0 >= p.K) ? 0u : ((1u << (p.K - c0)) - 1u);
      if (m) atomicOr(&bits[w], m);
    }
  } else if (n_probe > 0) {
    {
      if (qc * QW >= Lq) return;
      const int qq = qc * QW + q;
      const bool qvalid = qq < Lq;
      // ---- phase 1 (block, QW tokens at once): tau_g[q] = n_probe-th largest group maximum
      // The group maxima are read five times (four radix passes + the survivor list): with lds_gm (K <= 65536: the
      // block's QW x G keys are 64 KB) they come from memory ONCE, 16 loads in flight per thread, and the passes read
      // LDS (s_gm[g * QW + q]: conflict-free) -- the kernel was bound by the latency of those strided re-reads.
      const bool use_groups = G > (int64_t)n_probe;
      if (lds_gm) {
        for (int64_t g0 = r; g0 < G; g0 += 16 * NR) {
          uint32_t kv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) kv[u] = (qvalid && g0 + NR * u < G) ? gm[(g0 + NR * u) * LQP + qq] : 0u;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (g0 + NR * u < G) s_gm[(g0 + NR * u) * QW + q] = kv[u];
        }
        __syncthreads();
      }
      auto each_group = [&](auto&& cb) {   // cb(group, key) over this thread's groups of its token
        if (!qvalid) return;
        if (lds_gm) {
          for (int64_t g = r; g < G; g += NR) cb(g, s_gm[g * QW + q]);
        } else {
          for (int64_t g = r; g < G; g += 8 * NR) {   // 8 independent loads in flight
            uint32_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kv[u] = (g + NR * u < G) ? gm[(g + NR * u) * LQP + qq] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (g + NR * u < G) cb(g + NR * u, kv[u]);
          }
        }
      };
      if (use_groups) {
        radix_select<QW>([&](auto&& cb) { each_group([&](int64_t, uint32_t key) { cb(key); }); }, n_probe, hist, part,
                         s_prefix, s_rem, tid);
        if (r == 0) s_taug[q] = s_prefix[q];
      } else if (r == 0) {
        s_taug[q] = 0;
      }
      if (r == 0) s_gcnt[q] = 0;
      __syncthreads();
      // ---- surviving groups -> per-token lists (order irrelevant)
      {
        const uint32_t taug = s_taug[q];
        each_group([&](int64_t g, uint32_t key) {
          if (key >= taug) {
            const uint32_t pos = atomicAdd(&s_gcnt[q], 1u);
            if (pos < NP_PROBE_CAPG) s_glist[q * NP_PROBE_CAPG + pos] = (uint32_t)g;
          }
        });
      }
      __syncthreads();
      // ---- phases 2+3: one wave per token
      for (int t = wave; t < QW; t += 4) {
        const int tq = qc * QW + t;
        if (tq >= Lq) break;
        const uint32_t ng = s_gcnt[t];
        const float* col = QCT + tq;
        if (ng <= NP_PROBE_CAPG) {
          // element slot e = j*64 + lane -> (group list[e>>5], member e&31); key' = okey+1, 0 = absent
          const int nslots = (int)((ng * 32 + 63) / 64);
          uint32_t keys[32];
          uint32_t cids[32];
          // all of the lane's element loads are issued before the first is used (unconditional, clamped addresses): a
          // load inside the validity branch was waited for on the spot, 32 memory round trips per token
          float raw[32];
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            const uint32_t e = (uint32_t)j * 64 + lane;
            const bool in = j < nslots && e < ng * 32;
            const int64_t c = in ? (int64_t)s_glist[t * NP_PROBE_CAPG + (e >> 5)] * 32 + (e & 31) : 0;
            cids[j] = (in && c < p.K) ? (uint32_t)c : 0xFFFFFFFFu;
            raw[j] = col[(cids[j] != 0xFFFFFFFFu ? (int64_t)cids[j] : 0) * LQP];
          }
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            const uint32_t c = cids[j];
            const bool ok = c != 0xFFFFFFFFu && (!p.elig || ((p.elig[c >> 5] >> (c & 31)) & 1u));
            keys[j] = ok ? okey(raw[j]) + 1u : 0u;
            cids[j] = ok ? c : 0u;
          }
          uint32_t tau, rem;
          wave_select_mark(nslots, n_probe, keys, tau, rem);
          if (lane == 0) tauq[tq] = tau ? tau - 1u : 0u;
          // ties at the cut: select_nth_unstable leaves them unspecified (search.rs:405-409); the batched path's
          // heaps keep the LOWEST centroid ids (entries are (Reverse(score), id), search.rs:164-199), so both
          // paths take the `rem` smallest ids among the equal scores
          const uint32_t neq = wave_count(nslots, keys, [&](uint32_t k) { return k == tau && tau != 0; });
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            if (j < nslots) {
              const bool gt = keys[j] > tau && keys[j] != 0;
              const bool eq = keys[j] == tau && tau != 0;
              if (gt || (eq && neq <= rem)) atomicOr(&bits[cids[j] >> 5], 1u << (cids[j] & 31));
            }
          }
          if (neq > rem) {
            uint32_t last = 0;
            for (uint32_t r = 0; r < rem; ++r) {
              uint32_t m = 0xFFFFFFFFu;
#pragma unroll
              for (int j = 0; j < 32; ++j)
                if (j < nslots && keys[j] == tau && tau != 0 && (r == 0 || cids[j] > last)) m = min(m, cids[j]);
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
              if (lane == 0) atomicOr(&bits[m >> 5], 1u << (m & 31));
              last = m;
            }
          }
        } else {
          // degenerate tie-heavy input: too many surviving groups for registers.  Same selection with
          // the keys re-read from memory in every counting step (slow, exact).
          const uint32_t taug = s_taug[t];
          auto count_ge = [&](uint32_t trial, bool strict) {
            uint32_t cnt = 0;
            for (int64_t g0 = 0; g0 < G; g0 += 64) {
              const int64_t g = g0 + lane;
              uint32_t mine = 0;
              if (g < G && gm[g * LQP + tq] >= taug) {
                const uint32_t em = p.elig ? p.elig[g] : 0xFFFFFFFFu;
                for (int i = 0; i < 32; ++i) {
                  const int64_t c = g * 32 + i;
                  if (c < p.K && ((em >> i) & 1u)) {
                    const uint32_t k1 = okey(col[c * LQP]) + 1u;
                    mine += strict ? (k1 > trial) : (k1 >= trial);
                  }
                }
              }
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
              cnt += mine;
            }
            return cnt;
          };
          uint32_t prefix = 0;
          for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);
            if (count_ge(trial, false) >= n_probe) prefix = trial;
          }
          const uint32_t gt = count_ge(prefix, true);
          const uint32_t rem = n_probe > gt ? n_probe - gt : 0u;
          if (lane == 0) tauq[tq] = prefix ? prefix - 1u : 0u;
          uint32_t taken = 0;   // ties before this block of 64 groups; inside it they rank by (group, member) = ascending id
          for (int64_t g0 = 0; g0 < G; g0 += 64) {
            const int64_t g = g0 + lane;
            const bool gv = g < G && gm[g * LQP + tq] >= taug;
            const uint32_t em = gv ? (p.elig ? p.elig[g] : 0xFFFFFFFFu) : 0u;
            uint32_t eqm = 0, gtm = 0;
            for (int i = 0; i < 32; ++i) {
              const int64_t c = g * 32 + i;
              const bool ok = gv && c < p.K && ((em >> i) & 1u);
              const uint32_t k1 = ok ? okey(col[c * LQP]) + 1u : 0u;
              if (ok && k1 > prefix) gtm |= 1u << i;
              if (ok && k1 == prefix && prefix != 0) eqm |= 1u << i;
            }
            uint32_t incl = (uint32_t)__popc(eqm);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
              if (lane >= o) incl += v;
            }
            uint32_t rank = taken + incl - (uint32_t)__popc(eqm);
            uint32_t take = gtm;
            for (uint32_t m = eqm; m; m &= m - 1) {
              if (rank < rem) take |= m & (0u - m);
              ++rank;
            }
            if (take) atomicOr(&bits[g], take);
            taken += (uint32_t)__shfl((int)incl, 63);
          }
        }
      }
      __syncthreads();
    }
  }
}

#define NP_PROBE_NF 8   // probe_finish workgroups per query (each owns 1/NF of the marked-cell bitmap)
__global__ void __launch_bounds__(256) probe_finish_kernel(ProbeP p) {
  __shared__ uint32_t s_ntmp, s_nfinal, s_obase;
  const int b = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int Lq = p.qoff[b + 1] - p.qoff[b];
  const int64_t G = p.KP >> 5;
  const int LQP = p.LQP;
  const float* QCT = p.QCT + (int64_t)b * p.KP * LQP;
  uint32_t* bits = p.cellbits + (int64_t)b * G;
  const uint32_t* s_tauq = p.tauq + (int64_t)b * LQP;
  const int64_t pool = p.elig ? (int64_t)*p.n_elig : p.K;
  const int64_t eff = p.nprobe_dev ? (int64_t)*p.nprobe_dev : (int64_t)p.nprobe;
  const uint32_t n_probe = (uint32_t)min(eff, pool);
  if (tid == 0) { s_ntmp = 0; s_nfinal = 0; }
  __syncthreads();
  // ---- compact the marked cells of this workgroup's slice of the bitmap (words [w_lo, w_hi))
  const int64_t wper = (G + NP_PROBE_NF - 1) / NP_PROBE_NF;
  const int64_t w_lo = (int64_t)f * wper, w_hi = min(G, w_lo + wper);
  uint32_t* tmp = p.cells_tmp + (int64_t)b * p.KP + w_lo * 32;   // at most 32 cells per word: the slice cannot overflow
  for (int64_t w = w_lo + tid; w < w_hi; w += 256) {
    uint32_t m = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      tmp[atomicAdd(&s_ntmp, 1u)] = (uint32_t)(w * 32 + bit);
    }
  }
  __syncthreads();
  // ---- threshold.
  // dense path (search.rs:417-425): keep c iff finite-first max_q QC[q,c] >= t_cs; max_by keeps the
  //   LAST of equal maxima, so an all-non-finite column yields QC[Lq-1, c].  One LANE per cell.
  // batched path (search.rs:184-196,243-251): the max runs only over (q,c) pairs that were ever
  //   pushed into token q's slab-local heap: pushed <=> fewer than n_probe earlier centroids of the
  //   same slab score >= QC[q,c].  Pairs inside token q's global top-n_probe are always pushed, so
  //   the slab prefix is only counted for a token with QC[q,c] >= t_cs outside its top-n_probe.  One WAVE per cell.
  const uint32_t ntmp = s_ntmp;
  uint32_t* lst = tmp;   // survivors are compacted in place (reads of entry i happen before any write to slot <= i)
  if (!p.has_thr) {
    if (tid == 0) s_nfinal = ntmp;
  } else if (p.slab <= 0) {
    for (uint32_t i0 = 0; i0 < ntmp; i0 += 256) {
      const uint32_t i = i0 + tid;
      uint32_t c = 0;
      bool pass = false;
      if (i < ntmp) {
        c = tmp[i];
        const float4* row4 = reinterpret_cast<const float4*>(QCT + (int64_t)c * LQP);
        uint32_t km = 0;
        for (int q4 = 0; 4 * q4 < Lq; ++q4) {
          const float4 v = row4[q4];
          km = max(km, okey(v.x));
          if (4 * q4 + 1 < Lq) km = max(km, okey(v.y));
          if (4 * q4 + 2 < Lq) km = max(km, okey(v.z));
          if (4 * q4 + 3 < Lq) km = max(km, okey(v.w));
        }
        float mx;
        if (km != 0) mx = unkey(km);
        else mx = (Lq > 0) ? QCT[(int64_t)c * LQP + Lq - 1] : NP_NEG_INF;
        pass = mx >= p.thr;
      }
      __syncthreads();   // every read of this step's entries is done
      if (pass) lst[atomicAdd(&s_nfinal, 1u)] = c;
      __syncthreads();
    }
  } else {
    for (uint32_t i0 = 0; i0 < ntmp; i0 += 4) {
      const uint32_t i = i0 + wave;
      bool pass = false;
      uint32_t c = 0;
      if (i < ntmp) {
        c = tmp[i];
        const float* row = QCT + (int64_t)c * LQP;
        const int64_t slab0 = ((int64_t)c / p.slab) * p.slab;
        uint32_t km = 0;       // best finite score among the always-pushed pairs
        int first_sel = 0x7FFFFFFF;
        for (int q0 = 0; q0 < Lq && !pass; q0 += 64) {
          const int qx = q0 + lane;
          const bool qv = qx < Lq;
          const float v = qv ? row[qx] : 0.f;
          const uint32_t k = qv ? okey(v) : 0u;
          const bool sel = qv && k >= s_tauq[qx];
          uint32_t kk2 = sel ? k : 0u;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) kk2 = max(kk2, (uint32_t)__shfl_xor((int)kk2, o));
          km = max(km, kk2);
          const unsigned long long selb = __ballot(sel);
          if (selb && first_sel == 0x7FFFFFFF) first_sel = q0 + (__ffsll((long long)selb) - 1);
          if (km != 0 && unkey(km) >= p.thr) { pass = true; break; }
          // tokens outside their top-n_probe whose score alone would pass: count the slab prefix
          unsigned long long cand = __ballot(qv && !sel && finitef(v) && v >= p.thr);
          while (cand && !pass) {
            const int ql = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const int q2 = q0 + ql;
            const uint32_t k2 = __shfl((int)k, ql);
            uint32_t cnt = 0;
            for (int64_t c0 = slab0; c0 < (int64_t)c && cnt < n_probe; c0 += 64) {
              const int64_t cc = c0 + lane;
              const bool ge = cc < (int64_t)c && okey(QCT[cc * LQP + q2]) >= k2;
              cnt += (uint32_t)__popcll(__ballot(ge));
            }
            if (cnt < n_probe) pass = true;   // pushed, finite and >= t_cs
          }
        }
        if (!pass && km == 0 && first_sel != 0x7FFFFFFF) {
          // every always-pushed score is non-finite: max_score() keeps the first inserted one
          pass = row[first_sel] >= p.thr;
        }
      }
      __syncthreads();
      if (pass && lane == 0) lst[atomicAdd(&s_nfinal, 1u)] = c;
      __syncthreads();
    }
  }
  __syncthreads();
  // ---- append this slice's cells to the query's list
  const uint32_t nfin = s_nfinal;
  if (tid == 0) {
    s_obase = nfin ? (uint32_t)atomicAdd(&p.n_cells[b], (int32_t)nfin) : 0u;
    if (nfin && p.ctr) atomicAdd(&p.ctr->n_cells, (unsigned long long)nfin);
  }
  __syncthreads();
  uint32_t* outc = p.cells + (int64_t)b * p.KP + s_obase;
  for (uint32_t i = tid; i < nfin; i += 256) outc[i] = lst[i];
}

// Group maxima restricted to eligible centroids (subset path): gmax[b][g][q] = max over eligible
// members of okey(QCT[b][g*32+i][q]).  One wave per (b, g); lanes = query tokens.
__global__ void __launch_bounds__(256) masked_gmax_kernel(const float* __restrict__ QCT, int64_t KP, int64_t K, int LQP,
                                                          const uint32_t* __restrict__ elig,
                                                          uint32_t* __restrict__ gmax) {
  const int64_t G = KP >> 5;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  if (g >= G) return;
  const uint32_t em = elig[g];
  for (int q0 = 0; q0 < LQP; q0 += 64) {
    const int q = q0 + lane;
    if (q >= LQP) break;
    uint32_t km = 0;
    for (int i = 0; i < 32; ++i) {
      const int64_t c = g * 32 + i;
      if (c < K && ((em >> i) & 1u)) km = max(km, okey(QCT[((int64_t)b * KP + c) * LQP + q]));
    }
    gmax[((int64_t)b * G + g) * LQP + q] = km;
  }
}

// ---------------------------------------------------------------------------------------------
// subset pre-filter (search.rs:350-382, 434-437)
// ---------------------------------------------------------------------------------------------
// one wave per subset doc: doc bitmap (shard-local) + eligible-centroid bitmap
__global__ void __launch_bounds__(256) subset_kernel(const int64_t* __restrict__ subset, int64_t n, int64_t doc_begin,
                                                     int64_t n_docs, const int64_t* __restrict__ doc_off,
                                                     CodeArr codes, uint32_t* __restrict__ docbits,
                                                     uint32_t* __restrict__ elig) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t d = subset[i] - doc_begin;
  if (d < 0 || d >= n_docs) return;
  if (lane == 0 && docbits) atomicOr(&docbits[d >> 5], 1u << (d & 31));
  if (elig) {
    const int64_t s = doc_off[d], e = doc_off[d + 1];
    for (int64_t t = s + lane; t < e; t += 64) {
      const uint32_t c = codes[t];
      atomicOr(&elig[c >> 5], 1u << (c & 31));
    }
  }
}

// n_elig + effective nprobe = clamp(nprobe * N / |subset|, nprobe, n_elig)  (search.rs:370-382)
__global__ void __launch_bounds__(256) subset_nprobe_kernel(const uint32_t* __restrict__ elig, int64_t words,
                                                            int nprobe, int64_t n_total, int64_t subset_len,
                                                            int32_t* __restrict__ n_elig, int32_t* __restrict__ eff) {
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int c = 0;
  for (int64_t w = threadIdx.x; w < words; w += 256) c += __popc(elig[w]);
  atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ne = s_cnt;
    *n_elig = ne;
    long long e = nprobe;
    if (ne > 0) {
      unsigned long long scaled =
          subset_len > 0 ? (unsigned long long)nprobe * (unsigned long long)n_total / (unsigned long long)subset_len
                         : (unsigned long long)nprobe;
      long long sc = scaled > 0x7FFFFFFFull ? 0x7FFFFFFFll : (long long)scaled;
      if (sc < nprobe) sc = nprobe;
      if (sc > ne) sc = ne;
      e = sc;
    }
    *eff = (int32_t)e;
  }
}

// ---------------------------------------------------------------------------------------------
// S3  candidates: union of the probed posting lists as a per-query bitmap over shard docs
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mark_candidates_kernel(const uint32_t* __restrict__ cells,
                                                              const int32_t* __restrict__ n_cells, int64_t KP,
                                                              const int64_t* __restrict__ ivf_off,
                                                              const uint32_t* __restrict__ ivf,
                                                              const uint32_t* __restrict__ subset_bits, int64_t NW,
                                                              uint32_t* __restrict__ docbits, Counters* ctr) {
  const int b = blockIdx.y;
  const int nc = n_cells[b];
  uint32_t* bits = docbits + (int64_t)b * NW;
  unsigned long long ids = 0;
  for (int i = blockIdx.x; i < nc; i += gridDim.x) {
    const uint32_t c = cells[(int64_t)b * KP + i];
    const int64_t s = ivf_off[c], e = ivf_off[c + 1];
    if (threadIdx.x == 0) ids += (unsigned long long)(e - s);
    for (int64_t j = s + threadIdx.x; j < e; j += 256) {
      const uint32_t d = ivf[j];
      const uint32_t m = 1u << (d & 31);
      if (subset_bits && !(subset_bits[d >> 5] & m)) continue;   // search.rs:434-437
      if (!(bits[d >> 5] & m)) atomicOr(&bits[d >> 5], m);      // stale read only costs an extra atomic
    }
  }
  if (threadIdx.x == 0 && ids) atomicAdd(&ctr->n_ivf_ids, ids);
}

#define NP_CHUNK_WORDS 1024  // bitmap words per compaction block (32768 docs)

// S3 without memory atomics: block (slice, b) owns a contiguous range of the query's document bitmap (a whole number
// of compaction chunks, <= 128 KB) IN LDS, sweeps ALL of the query's posting lists and keeps the ids that fall into
// its range (the lists come out of L2 for every slice after the first), then writes the range out coalesced together
// with its chunk populations: no bitmap memset, no device-scope atomicOr over an 80 MB bitmap array (10 M documents
// x 64 queries: those ran at ~30 G atomics/s), no separate count pass.  Work items are 512 consecutive entries of one
// list, dealt round-robin to the 16 waves.
#define NP_MARK_CELLS 1024   // probed cells staged per pass
__global__ void __launch_bounds__(1024) mark_slices_kernel(const uint32_t* __restrict__ cells,
                                                           const int32_t* __restrict__ n_cells, int64_t KP,
                                                           const int64_t* __restrict__ ivf_off,
                                                           const uint32_t* __restrict__ ivf,
                                                           const uint32_t* __restrict__ subset_bits, int64_t NW,
                                                           int slice_chunks, int nchunks, uint32_t* __restrict__ docbits,
                                                           int32_t* __restrict__ chunk_counts, Counters* ctr,
                                                           int sorted_lists /* every posting list ascends: a block reads only the
                                                                               part of each list inside its range (two bisections per
                                                                               list) instead of sweeping all of it -- with ns ranges
                                                                               per query the sweep reads every list ns times */) {
  extern __shared__ uint32_t s_bits[];   // slice_chunks * NP_CHUNK_WORDS words
  __shared__ int64_t s_start[NP_MARK_CELLS];
  __shared__ uint32_t s_len[NP_MARK_CELLS];
  __shared__ uint32_t s_items[NP_MARK_CELLS + 1];   // exclusive prefix of the lists' item counts
  __shared__ uint32_t s_wsum[16];
  const int b = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t w0 = (int64_t)sl * slice_chunks * NP_CHUNK_WORDS;
  const int nw = (int)min((int64_t)slice_chunks * NP_CHUNK_WORDS, NW - w0);
  const uint32_t lo = (uint32_t)(w0 * 32), span = (uint32_t)nw * 32u;
  for (int i = tid; i < slice_chunks * NP_CHUNK_WORDS; i += 1024) s_bits[i] = 0;
  const int nc = n_cells[b];
  unsigned long long ids = 0;
  for (int c0 = 0; c0 < nc; c0 += NP_MARK_CELLS) {
    __syncthreads();   // the previous pass's tables are no longer read; s_bits zeroed
    const int m = min(NP_MARK_CELLS, nc - c0);
    uint32_t len = 0;
    if (tid < m) {
      const uint32_t c = cells[(int64_t)b * KP + c0 + tid];
      int64_t s0 = ivf_off[c];
      len = (uint32_t)(ivf_off[c + 1] - s0);
      ids += len;
      if (sorted_lists && gridDim.x > 1 && len > 64u) {
        // entries with lo <= id < lo + span: [first id >= lo, first id >= lo + span)
        const uint32_t* L = ivf + s0;
        uint32_t a = 0, z = len;
        while (a < z) {
          const uint32_t mid = (a + z) >> 1;
          if (L[mid] < lo) a = mid + 1;
          else z = mid;
        }
        const uint32_t first = a;
        const uint64_t hi64 = (uint64_t)lo + (uint64_t)span;
        z = len;
        while (a < z) {
          const uint32_t mid = (a + z) >> 1;
          if ((uint64_t)L[mid] < hi64) a = mid + 1;
          else z = mid;
        }
        s0 += first;
        len = a - first;
      }
      s_start[tid] = s0;
      s_len[tid] = len;
    }
    // exclusive scan of the item counts over the block
    const uint32_t items = (len + 511u) >> 9;
    uint32_t incl = items;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; ++k) woff += s_wsum[k];
    s_items[tid] = woff + incl - items;
    if (tid == 1023) s_items[1024] = woff + incl;
    __syncthreads();
    const uint32_t total = s_items[1024];
    for (uint32_t it = (uint32_t)wave; it < total; it += 16) {
      int a = 0, z = m;   // the list this item belongs to: largest t with s_items[t] <= it (empty lists have no items)
      while (z - a > 1) {
        const int mid = (a + z) >> 1;
        if (s_items[mid] <= it) a = mid;
        else z = mid;
      }
      const uint32_t off = (it - s_items[a]) << 9, ln = s_len[a];
      const uint32_t* src = ivf + s_start[a] + off;
      uint32_t d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t j = k * 64 + lane;
        d[k] = (off + j < ln) ? src[j] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t rel = d[k] - lo;
        if (d[k] != 0xFFFFFFFFu && rel < span) {
          const uint32_t bit = 1u << (d[k] & 31);
          if (subset_bits && !(subset_bits[d[k] >> 5] & bit)) continue;   // search.rs:434-437
          atomicOr(&s_bits[rel >> 5], bit);
        }
      }
    }
  }
  __syncthreads();
  // write the range out and count its chunks: wave w takes chunks w, w + 16, ...
  uint32_t* bits = docbits + (int64_t)b * NW + w0;
  for (int ch = wave; ch < slice_chunks; ch += 16) {
    const int64_t gch = (int64_t)sl * slice_chunks + ch;
    if (gch >= nchunks) break;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < NP_CHUNK_WORDS / 64; ++k) {
      const int w = ch * NP_CHUNK_WORDS + k * 64 + lane;
      if (w < nw) {
        const uint32_t v = s_bits[w];
        bits[w] = v;
        cnt += __popc(v);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) chunk_counts[(int64_t)b * nchunks + gch] = cnt;
  }
  if (sl == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ids += __shfl_xor(ids, o);
    if (lane == 0 && ids) atomicAdd(&ctr->n_ivf_ids, ids);
  }
}

__global__ void __launch_bounds__(256) count_chunks_kernel(const uint32_t* __restrict__ docbits, int64_t NW,
                                                           int nchunks, int32_t* __restrict__ chunk_counts) {
  __shared__ int s_cnt;
  const int b = blockIdx.y, ch = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t* bits = docbits + (int64_t)b * NW;
  int c = 0;
  for (int k = 0; k < NP_CHUNK_WORDS / 256; ++k) {
    const int64_t w = (int64_t)ch * NP_CHUNK_WORDS + k * 256 + threadIdx.x;
    if (w < NW) c += __popc(bits[w]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) chunk_counts[(int64_t)b * nchunks + ch] = s_cnt;
}

// Candidate pool.  The candidate arrays (doc id, 16-B meta record, approximate score) are ONE pool of P entries per
// workspace instead of B x n_docs strides: query b's candidates occupy [cand_base[b], cand_base[b] + n_cand[b]).
// A batch whose candidates do not fit the pool together is processed in ROUNDS: plan_rounds_kernel packs the
// queries first-fit in order (a query never exceeds n_docs <= P entries), S3-compaction / S4 / S5 then run once
// per round on the queries of that round (round_of[b]); the host enqueues the worst-case number of rounds and the
// kernels of a round nobody was assigned to exit at once.
struct RoundPlan {
  int32_t* n_cand;      // [B]
  int64_t* cand_base;   // [B] first pool entry of query b (inside its round)
  int32_t* round_of;    // [B]
  int32_t* round_tab;   // [2 * max_rounds]: first query, one-past-last query of each round; [2*max_rounds] = n_rounds
  int32_t* order;       // [B] the queries of every round, heaviest (most candidates) first, at the round's query range
};

__global__ void __launch_bounds__(256) plan_rounds_kernel(const int32_t* __restrict__ chunk_counts, int nchunks, int B,
                                                          int64_t pool, int max_rounds, RoundPlan rp, Counters* ctr,
                                                          const int32_t* __restrict__ n_direct = nullptr /* [B]: the queries' candidate
                                                              counts as they are (the zeroth level has counted -- and charged --
                                                              them: gain_count_kernel) instead of sums of chunk populations */) {
  __shared__ int s_n[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = wave; b < B; b += 4) {
    int c = 0;
    if (n_direct) c = lane == 0 ? n_direct[b] : 0;
    else
    for (int j = lane; j < nchunks; j += 64) c += chunk_counts[(int64_t)b * nchunks + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) s_n[b] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int64_t cur = 0;
    unsigned long long total = 0;
    int r = 0;
    for (int i = 0; i < 2 * max_rounds; ++i) rp.round_tab[i] = 0;
    rp.round_tab[0] = 0;
    for (int b = 0; b < B; ++b) {
      const int n = s_n[b];
      if (cur + n > pool && r + 1 < max_rounds) {   // close the round (the host's max_rounds bound is never hit)
        rp.round_tab[2 * r + 1] = b;
        ++r;
        rp.round_tab[2 * r] = b;
        cur = 0;
      }
      rp.n_cand[b] = (cur + n > pool) ? (int)max((int64_t)0, pool - cur) : n;
      rp.cand_base[b] = cur;
      rp.round_of[b] = r;
      cur += rp.n_cand[b];
      total += (unsigned long long)rp.n_cand[b];
    }
    rp.round_tab[2 * r + 1] = B;
    rp.round_tab[2 * max_rounds] = r + 1;
    if (!n_direct) atomicAdd(&ctr->n_candidates, total);
    atomicMax(&ctr->n_rounds, (unsigned long long)(r + 1));
  }
  __syncthreads();
  // heaviest-first order inside each round (rank by counting; rounds are contiguous query ranges): the per-XCD
  // kernels hand queries to XCDs in this order, first come first served, so no XCD is left with the long tail
  for (int b = tid; b < B; b += 256) {
    const int r = rp.round_of[b], rb = rp.round_tab[2 * r], re = rp.round_tab[2 * r + 1];
    const int n = rp.n_cand[b];
    int rank = 0;
    for (int j = rb; j < re; ++j) {
      const int nj = rp.n_cand[j];
      rank += (nj > n || (nj == n && j < b)) ? 1 : 0;
    }
    rp.order[rb + rank] = b;
  }
}

// One query per XCD at a time, handed out dynamically.  Workgroup w runs on XCD w % 8 (tools/probes/xcc_probe.hip);
// at step j every workgroup of XCD x works on the query in slot[x][j], which the first of them to arrive fills with
// the next entry of the heaviest-first order (global ticket).  -1 = empty, -2 = being filled, -3 = no query left.
// The workgroup that fills a slot is running, so nobody waits on a workgroup that is not resident.
// Once the order is exhausted the filling workgroup calls `steal()`: it may name a query that another XCD is still
// working on (its documents are claimed from a shared cursor, so any number of XCDs can share one query) or return -3.
template <class Steal>
__device__ __forceinline__ int xcd_next_query(int32_t* slots, int32_t* ticket, int x, int step, int B,
                                              const int32_t* order, int rb, int re, Steal&& steal) {
  if (step > B) return -3;   // slots hold B + 1 steps per XCD
  int32_t* slot = slots + (int64_t)x * (B + 1) + step;
  int v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == -1 && atomicCAS(slot, -1, -2) == -1) {
    const int t = atomicAdd(ticket, 1) + 1;   // the ticket starts at -1 like the slots (one 0xFF fill clears both)
    v = (t < re - rb) ? order[rb + t] : steal();
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  }
  while (v == -1 || v == -2) {
    __builtin_amdgcn_s_sleep(4);
    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

// writes ascending doc ids; thread t owns words [4t, 4t+4) of the chunk so the order is preserved
__global__ void __launch_bounds__(256) compact_kernel(const uint32_t* __restrict__ docbits, int64_t NW, int nchunks,
                                                      const int32_t* __restrict__ chunk_counts,
                                                      uint32_t* __restrict__ cand, RoundPlan rp, int round,
                                                      const uint4* __restrict__ doc_meta, uint4* __restrict__ cand_meta) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (rp.round_of[b] != round) return;
  const int32_t* cc = chunk_counts + (int64_t)b * nchunks;
  {
    int part = 0;
    for (int j = tid; j < ch; j += 256) part += cc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) s_wave[wave] = part;
    __syncthreads();
    if (tid == 0) s_base = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  const int base = s_base;
  __syncthreads();
  const uint32_t* bits = docbits + (int64_t)b * NW;
  const int64_t w0 = (int64_t)ch * NP_CHUNK_WORDS + tid * 4;
  uint32_t w[4];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w[k] = (w0 + k < NW) ? bits[w0 + k] : 0u;
    cnt += __popc(w[k]);
  }
  // exclusive scan of cnt over the block
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < wave; ++k) woff += s_wave[k];
  int pos = base + woff + incl - cnt;
  const int limit = rp.n_cand[b];
  uint32_t* out = cand ? cand + rp.cand_base[b] : nullptr;
  uint4* outm = cand_meta + rp.cand_base[b];
  // (gathering a thread's records 8 at a time with independent loads measured slower, 457 vs 388 us at 10 M documents:
  // the kernel is bound by the 16-B-of-a-line record gather itself, 1.9 % of the documents are candidates)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t m = w[k];
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const uint32_t d = (uint32_t)((w0 + k) * 32 + bit);
      if (pos < limit) {
        if (doc_meta) outm[pos] = doc_meta[d];   // {doc, distinct codes, list offset lo, offset hi | doc length << 8}: one 16-B gather
        if (out) out[pos] = d;     // the bare id list: the hot level of the filter (it finds a document's list block from the id and
      }                            // writes the records itself), the unfiltered selection, the debug trace
      ++pos;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// S3, ZEROTH filter level (round 6): an upper bound of a candidate's approximate score from the probed cells alone -- from
// what S3 sees while it unions the posting lists, before any list block is read.
//   top(q)  = token q's top-n_probe centroids (search.rs:388-414); theta_q = its n_probe-th best score (tauq, S2)
//   P       = the probed cells = union of top(q).  Without a centroid_score_threshold every probed cell is kept, and a centroid
//             OUTSIDE P is in no token's top-n_probe: QC[q, c] <= theta_q for every token q
//   in table units (u = S1's monotone u8 table):  ut_q = u(theta_q),   G(c) = sum_q max(0, u[q, c] - ut_q)   for c in P
//   U0(d)   = sum_q ut_q + sum_{c in P, d in list(c)} G(c)
//          >= sum_q max(ut_q, max_{c in P & codes(d)} u[q, c])  >=  sum_q max_{c in codes(d)} u[q, c] = U(d)
// (a sum over the document's probed cells instead of a per-token maximum: ONE accumulator per document).  d is in list(c) exactly when c is one of its codes, so the accumulator
// is a scatter-add of G(c) + 1 over the probed posting lists -- what mark_slices_kernel does with one bit.  U0 is an upper
// bound of the exact integer bound U, so it may stand in front of the existing cuts: with tau0 = the n_sel-th largest LOWER
// bound L among ANY set S0 of candidates (approx_ub_kernel on the ~3 n_sel documents with the largest U0), a document with
// U0 < tau0 - slack cannot be among the n_sel best (np_kernels.h "S4, upper-bound filter": the same argument with U0 >= U).
// CPU simulation on the metric corpus first (tools/sim/s3_gain_sim.py, profiles/r06_sim_s3_gain_*.txt): with t_cs = None the
// bound keeps 0.2-2 % of the 6.1 M candidates per query at nprobe 32 (every query) and 1-9 % of the 2.06 M at nprobe 8 for
// three queries in four (no pruning for the fourth: sum_q theta_q reaches its tau); with the default threshold 0.4 the ~900
// REMOVED cells of a query (search.rs:417-425) would have to enter theta (their scores reach 0.4) and sum theta' = 11.5
// exceeds tau ~ 10-12: nothing to gain there, so the level runs only where no threshold is set.
//
// WITH A THRESHOLD (search.rs:417-425) the cells it removes are still in some token's top-n_probe: folding them into theta lifts
// the floor to the cut (sum theta' = 11.5 on the metric corpus: nothing pruned).  Swept as BOUND-ONLY cells instead -- their lists
// add gains, only the KEPT cells' documents are candidates -- the floor stays sum theta_q and the level keeps 5-17 % of the
// candidates at 0.5 distinct codes per token and 8-39 % at 0.8 (profiles/r06_sim_s3_gain_thr_*.txt); on the metric corpus
// (0.23: 186 k candidates per query against 9.4 M posting entries to sweep) it costs more than it saves, which the run / skip
// rule finds out by itself.
// DEEPER THAN THE PROBE (n_ivf_probe < 32): the bound's floor sum_q ut_q falls with the depth (a token's 32nd best score instead
// of its 8th: 10.3 -> 9.2 score units on the metric corpus, against cuts of 10-12.5: one query in four had NO pruning at depth 8,
// every query at depth 32).  So the level probes on its own to depth max(n_ivf_probe, 32) (the S2 kernels once more): the cells
// beyond the search's own are BOUND-ONLY -- their lists add gains but make no candidates.  An accumulator holds 2 x the sum and
// bit 0 = "in a cell the search probed" (ds_or): a candidate is an accumulator with bit 0 set.
//
// One sweep over the probed posting lists, each block (range r, query b) holding the u16 accumulators of 32768 documents
// in LDS (two per dword, ds_add_u32; the gains are scaled so that twice the sum over ALL probed cells fits 16 bits: no carry
// into the neighbour), writes the accumulators out; three streaming passes over them count the candidates at the cut, emit S0
// (records for the exact bound) and emit the candidates that pass the cut (bare ids, ascending inside a range, ranges in any
// order: the hot level takes a claim's block offsets from its smallest id).  A range's part of a posting list comes from a
// static table built at open (ivf_split[c][r] = entries of list c with id < 32768 r).
// ---------------------------------------------------------------------------------------------
#define NP_UB_BINS 2048    // histogram bins of the integer bounds: bin = U >> hshift, hshift = log2(ROWB / 32) + 2 (U <= 255 * ROWB);
                           // u32 counters in LDS (8 KB): a workgroup's share of a query's documents is unbounded
#define NP_GAIN_RANGE NP_IVF_SPLIT_RANGE   // documents per accumulator range = the granularity of the posting lists' range table (np_internal.h)
#define NP_GAIN_CELLS 256        // probed cells staged per pass
#define NP_GAIN_ITEM 32          // posting entries per work item (half a wave)

// per query: ut_q, base = sum ut_q, G(c) + 1 per probed cell (scaled by 2^-shift, rounded up, when their sum would not fit
// 16 bits), and the trivial round plan of the S0 exact-bound launch (one round, identity order, fixed-size slices)
template <int RB>
__global__ void __launch_bounds__(256) gain_prep_kernel(const uint8_t* __restrict__ QCU, int64_t KP,
                                                        const uint32_t* __restrict__ cells, const int32_t* __restrict__ n_cells,
                                                        const uint32_t* __restrict__ tauq, int LQP, const float* __restrict__ qinv,
                                                        const int32_t* __restrict__ qoff, uint16_t* __restrict__ gain /* [B][KP] */,
                                                        uint32_t* __restrict__ gbase /* [B][4]: base, shift, level shift r, b0 */, int B, int s0cap,
                                                        RoundPlan rp0, int hshift, const uint32_t* __restrict__ real_bits /* [B][KP / 32] the cells the
                                                            search probes (S2's marks) when `cells` goes deeper; NULL: all of them */) {
  static_assert(RB == 32 || RB == 64, "rows of 32 or 64 query tokens");
  __shared__ uint32_t s_ut[RB];
  __shared__ uint32_t s_red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Lq = qoff[b + 1] - qoff[b];
  const float inv = qinv[b];
  if (tid < RB) {
    uint32_t u = 0;
    if (tid < Lq) {
      const uint32_t tau = tauq[(int64_t)b * LQP + tid];
      const float x = tau ? unkey(tau) : 0.f;     // 0: fewer than n_probe scored centroids (every one is probed)
      u = min((uint32_t)(fmaxf(x * inv, 0.f) * 254.0f) + 1u, 255u);   // the table's own rounding (qc_gemm_kernel)
    }
    s_ut[tid] = u;
  }
  if (tid == 0) {
    rp0.cand_base[b] = (int64_t)b * s0cap;
    rp0.round_of[b] = 0;
    rp0.order[b] = b;
    if (b == 0) {
      rp0.round_tab[0] = 0;
      rp0.round_tab[1] = B;
      rp0.round_tab[2] = 1;
    }
  }
  __syncthreads();
  const int nc = n_cells[b];
  uint16_t* gb = gain + (int64_t)b * KP;
  uint32_t tot = 0;
  for (int i = tid; i < nc; i += 256) {
    const uint32_t c = cells[(int64_t)b * KP + i];
    const uint4* row = reinterpret_cast<const uint4*>(QCU + ((int64_t)b * KP + c) * RB);
    uint32_t g = 0;
#pragma unroll
    for (int j = 0; j < RB / 16; ++j) {
      const uint4 v = row[j];
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = (w4[e] >> (8 * k)) & 0xFFu, t = s_ut[16 * j + 4 * e + k];
          g += max(u, t) - t;                      // padding tokens: u = t = 0
        }
    }
    gb[i] = (uint16_t)g;                           // <= 254 * 64
    tot += g;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o);
  if (lane == 0) s_red[wave] = tot;
  __syncthreads();
  tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  uint32_t sh = 0;
  // sum of ceil(g / 2^sh) <= tot / 2^sh + nc; the accumulators hold twice that.  (nc <= depth x query tokens <= 16384: the host
  // runs the level only then -- with more cells than accumulator units no scaling could fit)
  while (sh < 24u && (tot >> sh) + (uint32_t)nc > 32767u) ++sh;
  // stored: the scaled gain (< 2^15) | bit 15 = the search probes this cell itself (its documents are candidates)
  for (int i = tid; i < nc; i += 256) {
    const uint32_t c = cells[(int64_t)b * KP + i];
    const uint32_t real = real_bits ? (real_bits[(int64_t)b * (KP >> 5) + (c >> 5)] >> (c & 31)) & 1u : 1u;
    gb[i] = (uint16_t)((((uint32_t)gb[i] + (1u << sh) - 1u) >> sh) | (real << 15));
  }
  __syncthreads();
  // LEVELS.  Everything behind the sweep works on 8-bit levels of the bound instead of its 2048 histogram bins:
  //     level(U0) = 1 + min(254, ((U0 >> hshift) - b0) >> r),   b0 = base >> hshift   (0 = not a candidate)
  // monotone in U0, so "level >= level(cut)" keeps a superset of "bin >= cut" -- what a cut on an upper bound tolerates -- and the
  // documents' levels are ONE byte each in memory (the emission passes stream them) and 256 counters in the sweep.  r spreads
  // four times the largest single cell's gain over the 254 levels (a document in a handful of strong cells still resolves;
  // beyond that the level saturates: such documents are always kept).
  __shared__ uint32_t s_gmax;
  if (tid == 0) s_gmax = 0;
  __syncthreads();
  uint32_t gm = 0;
  for (int i = tid; i < nc; i += 256) gm = max(gm, (uint32_t)gb[i] & 0x7FFFu);
  if (gm) atomicMax(&s_gmax, gm);
  __syncthreads();
  if (tid == 0) {
    uint32_t base = 0;
    for (int q = 0; q < RB; ++q) base += s_ut[q];
    const uint32_t span = ((4u * s_gmax) << sh) >> hshift;
    uint32_t r = 0;
    while (r < 11u && (span >> r) > 254u) ++r;
    gbase[4 * b] = base;
    gbase[4 * b + 1] = sh;
    gbase[4 * b + 2] = r;
    gbase[4 * b + 3] = base >> hshift;
  }
}

// level of a document's accumulator (bit 0 = candidate, bits 1.. = the scaled gain sum); 0 = not a candidate
__device__ __forceinline__ uint32_t gain_level(uint32_t a, uint32_t base, uint32_t sh, uint32_t r, uint32_t b0, int hshift) {
  if (!(a & 1u)) return 0u;
  const uint32_t bin = min((base + ((a >> 1) << sh)) >> hshift, (uint32_t)(NP_UB_BINS - 1));
  return 1u + min(254u, (bin - b0) >> r);
}
// the level a histogram-bin threshold maps to (monotone: level(bin(doc)) >= level_of_bin(t) for every doc with bin >= t)
__device__ __forceinline__ uint32_t gain_level_of_bin(uint32_t t, uint32_t r, uint32_t b0) {
  return t <= b0 ? 1u : 1u + min(254u, (t - b0) >> r);
}

struct GainP {
  const uint32_t* cells;      // [B][KP] probed cells (S2)
  const int32_t* n_cells;     // [B]
  int64_t KP;
  const int64_t* ivf_off;     // [K + 1]
  const uint32_t* ivf;
  const uint32_t* split;      // [K][R1]: split[c * R1 + r] = entries of list c with id < r * NP_GAIN_RANGE
  int R1;
  const uint16_t* gain;       // [B][KP] by cell position
  const uint32_t* gbase;      // [B][4]: base, gain shift, level shift r, b0 = base >> hshift
  int hshift;
  uint32_t* hist0;            // [B][256] documents per level (sweep)
  int32_t* n_raw;             // [B] candidates (mode 0)
  const uint32_t* thr;        // [B] mode 1: LEVEL threshold of S0 (0 = no S0); mode 2: level of the cut (<= 1: keep every candidate)
  uint4* s0_meta;             // mode 1: [B][s0cap] records
  int s0cap;
  const void* ucodes;         // list blocks (headers): mode 1 builds the records itself
  int code_wide, ublock_stride;
  int64_t ovf_base;
  uint32_t* cand;             // mode 2: [pool] ids at cand_base[b]
  int32_t* n_emit;            // [B] zeroed: mode 1 slots handed out above the marginal bin; mode 2 (zeroed again) candidates emitted
  int32_t* n_marg;            // [B] zeroed: mode 1 slots handed out to the marginal bin
  const int32_t* n_hi;        // [B] documents above the marginal bin of S0 (gain_thr_kernel)
  uint8_t* lvl;               // [B][n_ranges * NP_GAIN_RANGE] the documents' levels, 0 = not a candidate (sweep -> emission passes)
  int n_ranges;
  RoundPlan rp;
  Counters* ctr;
};

// Sweep: block (range r, query b) scatter-adds the gains of the probed cells over its 32768 documents in LDS, then turns every
// accumulator into its 8-bit level, counts the levels into the query's 256-bin histogram and writes the levels out
// (lvl[b][doc], one byte per document, coalesced): the passes below (S0, candidates) are streaming reads of that array instead
// of further sweeps of the posting lists (measured, REST default regime at 10 M documents: 0.83 ms per sweep against ~0.15 ms
// per streaming pass), and the histogram gives the round plan the exact number of candidates at any cut.
// A wave works through TASKS of 16 probed cells with no block-wide barrier in between: the 16 cells' parts of their posting
// lists (~34 entries each at K = 2^16) are 16 loads in flight per lane, and the metadata of the wave's next task (cell id ->
// range table, list offset, gain: two dependent round trips) travels during the current one.  (First version: per pass of 256
// cells a prefix sum of 32-entry items and three barriers, four items in flight per half-wave -- one memory round trip per
// 128 items and ~5 us of barriers and dependent loads per pass: 2.4 ms per batch at t_cs = None, nprobe 32.)
// Levels 1 and 2 -- the documents of ONE weak cell, the bulk of the candidates -- are counted in registers: 64 lanes adding to
// the same two LDS counters at once serialise (a complete 2048-bin histogram of every candidate cost the sweep ~10 us per
// block).
#define NP_GAIN_TASK 16
__global__ void __launch_bounds__(1024) gain_sweep_kernel(GainP p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_acc[];   // NP_GAIN_RANGE / 2 words: two u16 accumulators each
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_wsum[16];
  const int b = blockIdx.y, r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = p.n_cells[b];
  const uint32_t lo = (uint32_t)r * NP_GAIN_RANGE;
  {
    uint4* a4 = reinterpret_cast<uint4*>(s_acc);
    for (int i = tid; i < NP_GAIN_RANGE / 8; i += 1024) a4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 256) s_hist[tid] = 0;
  }
  __syncthreads();
  const uint32_t* cellb = p.cells + (int64_t)b * p.KP;
  const uint16_t* gainb = p.gain + (int64_t)b * p.KP;
  unsigned long long ids = 0;
  constexpr int STEP = 16 * NP_GAIN_TASK;   // cells between a wave's consecutive tasks
  const int sub = lane & (NP_GAIN_TASK - 1);   // every quarter of the wave holds a copy of the task's 16 cells
  auto load_cell = [&](int t0) { return t0 + sub < nc ? cellb[t0 + sub] : 0u; };
  auto load_meta = [&](int t0, uint32_t c, int64_t& start, uint32_t& len, uint32_t& g) {
    start = 0;
    len = 0;
    g = 0;
    if (t0 + sub < nc) {
      const uint32_t* sp = p.split + (int64_t)c * p.R1 + r;
      const uint32_t s = sp[0], e = sp[1];
      const int64_t o0 = p.ivf_off[c];
      start = o0 + s;
      len = e - s;
      g = (uint32_t)gainb[t0 + sub];
      if (r == 0 && lane < NP_GAIN_TASK) ids += (unsigned long long)(p.ivf_off[c + 1] - o0);
    }
  };
  int t0 = wave * NP_GAIN_TASK;
  uint32_t c_next = 0, c_next2 = 0;
  int64_t st_n = 0;
  uint32_t ln_n = 0, g_n = 0;
  if (t0 < nc) {
    c_next = load_cell(t0);
    c_next2 = load_cell(t0 + STEP);
    load_meta(t0, c_next, st_n, ln_n, g_n);
  }
  while (t0 < nc) {
    const int64_t st = st_n;
    const uint32_t ln = ln_n, g = g_n;
    load_meta(t0 + STEP, c_next2, st_n, ln_n, g_n);    // the next task's tables and the cell ids of the one after it: on their
    c_next2 = load_cell(t0 + 2 * STEP);                 // way while this task's entries are read
    uint32_t d[NP_GAIN_TASK];
    const uint32_t st_lo = (uint32_t)st, st_hi = (uint32_t)((uint64_t)st >> 32);
    auto part = [&](int k) {   // cell k of the task: its part's address (wave-uniform: readlane)
      const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)st_lo, k), shi = (uint32_t)__builtin_amdgcn_readlane((int)st_hi, k);
      return p.ivf + (int64_t)(((uint64_t)shi << 32) | slo);
    };
#pragma unroll
    for (int k = 0; k < NP_GAIN_TASK; ++k) {
      const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)ln, k);
      d[k] = (uint32_t)lane < lk ? part(k)[lane] : lo;
    }
    uint32_t lmax = 0;
#pragma unroll
    for (int k = 0; k < NP_GAIN_TASK; ++k) {
      const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)ln, k), gk = (uint32_t)__builtin_amdgcn_readlane((int)g, k);
      lmax = max(lmax, lk);
      if ((uint32_t)lane < lk) {
        const uint32_t rel = d[k] - lo, shl = (rel & 1u) * 16u;
        if (gk & 0x7FFFu) atomicAdd(&s_acc[rel >> 1], ((gk & 0x7FFFu) * 2u) << shl);
        if (gk & 0x8000u) atomicOr(&s_acc[rel >> 1], 1u << shl);
      }
    }
    if (lmax > 64u) {                                   // parts longer than a wave (popular centroids)
#pragma unroll 1
      for (int k = 0; k < NP_GAIN_TASK; ++k) {
        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)ln, k);
        if (lk <= 64u) continue;
        const uint32_t gk = (uint32_t)__builtin_amdgcn_readlane((int)g, k);
        const uint32_t* src = part(k);
        for (uint32_t off = 64; off < lk; off += 64)
          if (off + (uint32_t)lane < lk) {
            const uint32_t rel = src[off + lane] - lo, shl = (rel & 1u) * 16u;
            if (gk & 0x7FFFu) atomicAdd(&s_acc[rel >> 1], ((gk & 0x7FFFu) * 2u) << shl);
            if (gk & 0x8000u) atomicOr(&s_acc[rel >> 1], 1u << shl);
          }
      }
    }
    t0 += STEP;
  }
  __syncthreads();
  // ---- scan: levels out, their histogram, candidate count
  const uint32_t base = p.gbase[4 * b], sh = p.gbase[4 * b + 1], lr = p.gbase[4 * b + 2], b0 = p.gbase[4 * b + 3];
  uint32_t c1 = 0, c2 = 0, cnt = 0;
  {
    // thread t takes the 16-byte pieces t, t + 1024, ... of the range (8 documents each): conflict-free LDS reads, coalesced
    // 8-byte stores
    const uint4* a4 = reinterpret_cast<const uint4*>(s_acc) + tid;
    uint2* out = reinterpret_cast<uint2*>(p.lvl + ((int64_t)b * p.n_ranges + r) * NP_GAIN_RANGE) + tid;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint4 vv = a4[k * 1024];
      const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
      uint32_t o[2] = {0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t l = gain_level((w4[e] >> (16 * h)) & 0xFFFFu, base, sh, lr, b0, p.hshift);
          o[e >> 1] |= l << (8 * (2 * (e & 1) + h));
          if (l) {
            ++cnt;
            if (l == 1u) ++c1;
            else if (l == 2u) ++c2;
            else atomicAdd(&s_hist[l], 1u);
          }
        }
      out[k * 1024] = make_uint2(o[0], o[1]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += (uint32_t)__shfl_xor((int)cnt, o);
    c1 += (uint32_t)__shfl_xor((int)c1, o);
    c2 += (uint32_t)__shfl_xor((int)c2, o);
  }
  if (lane == 0) {
    s_wsum[wave] = cnt;
    if (c1) atomicAdd(&s_hist[1], c1);
    if (c2) atomicAdd(&s_hist[2], c2);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < 16; ++k) tot += s_wsum[k];
    if (tot) atomicAdd(&p.n_raw[b], (int32_t)tot);
  }
  if (tid < 256) {
    const uint32_t v = s_hist[tid];
    if (v) atomicAdd(&p.hist0[(int64_t)b * 256 + tid], v);
  }
  if (r == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ids += __shfl_xor(ids, o);
    if (lane == 0 && ids) atomicAdd(&p.ctr->n_ivf_ids, ids);
  }
}

// Emission passes over the stored levels: block (range of 32768 documents, query b).
//   MODE 1  S0: the documents whose level lies ABOVE thr[b] fill slots [0, n_hi[b]) of the query's record slice, the documents of
//           the marginal level thr[b] fill the slots behind them as far as the slice goes (any subset of the candidates is a valid
//           S0; gain_thr_kernel has set n_s0[b] = min(cap, #level >= thr)).  Records are built from the list blocks' headers.
//   MODE 2  candidates: level >= thr[b] (every candidate when thr[b] <= 1), bare ids, ascending inside a range, the ranges in any
//           order (the hot level takes a claim's block offsets from its smallest id).
template <int MODE>
__global__ void __launch_bounds__(256) gain_emit_kernel(GainP p, int round) {
  // thread t takes the 16-byte pieces t, t + 256, ... (16 documents each, 8 pieces: every load of a step is one contiguous 4 KB,
  // all 8 in flight), ONE slot reservation per block (the counters of a batch's queries share two cache lines: a reservation per
  // 8192 documents serialised 78 k atomics on them)
  constexpr int NPC = NP_GAIN_RANGE / 16 / 256;   // pieces per thread: 8
  __shared__ uint32_t s_out[2];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE == 2 && p.rp.round_of[b] != round) return;
  const uint32_t thr = p.thr[b];
  if (MODE == 1 && thr == 0u) return;
  const int64_t d_block = (int64_t)blockIdx.x * NP_GAIN_RANGE;
  uint32_t keep[NPC / 2], marg[NPC / 2];   // bit 16 (i & 1) + j of word i >> 1: document j of piece i
#pragma unroll
  for (int w = 0; w < NPC / 2; ++w) keep[w] = marg[w] = 0;
  {
    const uint4* a4 = reinterpret_cast<const uint4*>(p.lvl + (int64_t)b * p.n_ranges * NP_GAIN_RANGE + d_block) + tid;
    uint4 vv[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) vv[i] = a4[i * 256];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const uint32_t w4[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const uint32_t l = (w4[e] >> (8 * h)) & 0xFFu;
          const uint32_t bit = 1u << (16 * (i & 1) + 4 * e + h);
          if (l != 0u && l >= thr) keep[i >> 1] |= bit;
          if (MODE == 1 && l == thr) marg[i >> 1] |= bit;
        }
    }
  }
  // Slots in ASCENDING document order (piece-major: piece i * 256 + t): the hot level stages a claim's 32 list blocks together,
  // and ids scattered over the whole range cost it DRAM locality (measured: 4.09 vs 3.36 ms per batch in the REST default regime
  // with a thread-major order).  Per piece row i an exclusive scan over the 256 threads of (kept above the marginal level |
  // marginal ones << 16) -- at most 4096 each -- then the rows' totals.
  __shared__ __attribute__((aligned(16))) uint32_t s_c[NPC][256];
  __shared__ uint32_t s_row[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const uint32_t k16 = (keep[i >> 1] >> (16 * (i & 1))) & 0xFFFFu, m16 = (marg[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
    s_c[i][tid] = (uint32_t)__popc(k16 & ~m16) | ((uint32_t)__popc(m16) << 16);
  }
  __syncthreads();
  for (int i = wave; i < NPC; i += 4) {
    const uint4 c4 = *reinterpret_cast<const uint4*>(&s_c[i][4 * lane]);
    const uint32_t mine = c4.x + c4.y + c4.z + c4.w;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += v;
    }
    const uint32_t ex = incl - mine;
    *reinterpret_cast<uint4*>(&s_c[i][4 * lane]) = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);
    if (lane == 63) s_row[i] = incl;
  }
  __syncthreads();
  uint32_t rowbase[NPC], tot_hi = 0, tot_m = 0;   // (kept separately: a range holds up to 32768 of either, beyond a 16-bit field)
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    rowbase[i] = tot_hi | (tot_m << 16);   // < 32768 each while i < NPC
    tot_hi += s_row[i] & 0xFFFFu;
    tot_m += s_row[i] >> 16;
  }
  if (tot_hi + tot_m == 0) return;   // block-uniform
  if (tid == 0) {
    s_out[0] = tot_hi ? (uint32_t)atomicAdd(&p.n_emit[b], (int32_t)tot_hi) : 0u;
    if constexpr (MODE == 1) s_out[1] = tot_m ? (uint32_t)p.n_hi[b] + (uint32_t)atomicAdd(&p.n_marg[b], (int32_t)tot_m) : 0u;
  }
  __syncthreads();
  if constexpr (MODE == 1) {
    const int cb = p.code_wide ? 4 : 2, hdr = 16 / cb, fit = p.ublock_stride - hdr;
    uint4* out = p.s0_meta + (int64_t)b * p.s0cap;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const uint32_t pre = s_c[i][tid];
      uint32_t pos = s_out[0] + (rowbase[i] & 0xFFFFu) + (pre & 0xFFFFu), posm = s_out[1] + (rowbase[i] >> 16) + (pre >> 16);
      uint32_t m = (keep[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
      const uint32_t mg = (marg[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
      while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1;
        const uint32_t d = (uint32_t)d_block + (uint32_t)((i * 256 + tid) * 16 + j);
        const uint32_t at = ((mg >> j) & 1u) ? posm++ : pos++;
        if (at < (uint32_t)p.s0cap) {
          const uint4 hd = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.ucodes) + (int64_t)d * p.ublock_stride * cb);
          const int64_t cl = (int)hd.x > fit ? p.ovf_base + (int64_t)hd.z * 4 : (int64_t)d * p.ublock_stride + hdr;
          out[at] = make_uint4(d, hd.x, (uint32_t)(cl & 0xFFFFFFFFll), (uint32_t)((cl >> 32) & 0xFF) | (hd.y << 8));
        }
      }
    }
  } else {
    uint32_t* out = p.cand + p.rp.cand_base[b];
    const uint32_t limit = (uint32_t)p.rp.n_cand[b];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      uint32_t pos = s_out[0] + (rowbase[i] & 0xFFFFu) + (s_c[i][tid] & 0xFFFFu);
      uint32_t m = (keep[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
      while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1;
        if (pos < limit) out[pos] = (uint32_t)d_block + (uint32_t)((i * 256 + tid) * 16 + j);
        ++pos;
      }
    }
  }
}

// the kept cells of every query (S2's list AFTER the centroid_score_threshold) as a bitmap: with a threshold the level sweeps every
// probed cell, and only the kept ones make candidates (gain_prep_kernel's real_bits)
__global__ void __launch_bounds__(256) cells_to_bits_kernel(const uint32_t* __restrict__ cells, const int32_t* __restrict__ n_cells,
                                                            int64_t KP, uint32_t* __restrict__ bits /* [B][KP / 32] zeroed */) {
  const int b = blockIdx.x;
  const int n = n_cells[b];
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t c = cells[(int64_t)b * KP + i];
    atomicOr(&bits[(int64_t)b * (KP >> 5) + (c >> 5)], 1u << (c & 31));
  }
}

// S0 = the ~target documents with the largest U0: thr[b] = the LEVEL at which the count from the top reaches the target (0 = no
// S0: flagged query, or fewer candidates than the target); n_hi[b] = documents above it (all taken), n_s0[b] = min(cap, documents
// at levels >= thr): the marginal level fills what is left of the slice.  One wave per query.
__global__ void __launch_bounds__(64) gain_thr_kernel(const uint32_t* __restrict__ hist /* [B][256] */, int target, int cap,
                                                      const int32_t* __restrict__ n_raw, const uint32_t* __restrict__ qflag,
                                                      uint32_t* __restrict__ thr, int32_t* __restrict__ n_hi,
                                                      int32_t* __restrict__ n_s0) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (qflag[b] != 0 || n_raw[b] <= target) {
    if (lane == 0) {
      thr[b] = 0;
      n_hi[b] = 0;
      n_s0[b] = 0;
    }
    return;
  }
  // lane l owns levels 255 - 4 l .. 252 - 4 l (descending); suffix counts by a wave scan
  const uint32_t* hb = hist + (int64_t)b * 256;
  uint32_t h4[4], mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h4[k] = hb[255 - 4 * lane - k];
    mine += h4[k];
  }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += v;
  }
  const unsigned long long reach = __ballot(incl >= (uint32_t)target);
  if (reach == 0ull) {   // cannot happen (n_raw > target and every candidate has a level >= 1): no S0
    if (lane == 0) {
      thr[b] = 0;
      n_hi[b] = 0;
      n_s0[b] = 0;
    }
    return;
  }
  const int first = __ffsll((long long)reach) - 1;
  if (lane == first) {
    uint32_t cum = incl - mine;
    int k = 0;
    for (; k < 3; ++k) {
      if (cum + h4[k] >= (uint32_t)target) break;
      cum += h4[k];
    }
    const int level = 255 - 4 * lane - k;
    thr[b] = (uint32_t)max(level, 1);
    n_hi[b] = (int32_t)cum;                                        // < target <= cap
    n_s0[b] = (int32_t)min(cum + h4[k], (uint32_t)cap);
  }
}

// The cut in levels and the candidates it keeps: lcut[b] = level of the histogram-bin threshold tau0 - slack (ub_thr_kernel on
// S0's exact lower bounds; 0 there = the level does not apply: every candidate stays), n_out[b] = documents at levels >= lcut
// (the sweep's histogram is complete); the work counters and the host's report.  One wave per query, then block 0's lane 0.
__global__ void __launch_bounds__(64) gain_count_kernel(const uint32_t* __restrict__ cut_bin, const uint32_t* __restrict__ hist,
                                                        const uint32_t* __restrict__ gbase, const int32_t* __restrict__ n_raw,
                                                        uint32_t* __restrict__ lcut, int32_t* __restrict__ n_out, Counters* ctr,
                                                        unsigned long long* __restrict__ h_report /* pinned host words: [0] candidates << 32 |
                                                            kept over the batch's queries, [1] posting entries swept, for the host's
                                                            run / skip policy (read a batch later, never waited for) */,
                                                        unsigned long long* __restrict__ d_report /* [2] zeroed: the batch's sums */,
                                                        int B) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const uint32_t cb = cut_bin[b];
  uint32_t n = (uint32_t)n_raw[b], lc = 1u;
  if (cb) {
    lc = gain_level_of_bin(cb, gbase[4 * b + 2], gbase[4 * b + 3]);
    uint32_t c = 0;
    for (int l = lane; l < 256; l += 64) c += (uint32_t)l >= lc ? hist[(int64_t)b * 256 + l] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    n = c;
  }
  if (lane == 0) {
    lcut[b] = lc;
    n_out[b] = (int32_t)n;
    atomicAdd(&ctr->n_candidates, (unsigned long long)n_raw[b]);
    atomicAdd(&ctr->n_level0, (unsigned long long)n);
    atomicAdd(&d_report[0], (unsigned long long)n_raw[b]);
    atomicAdd(&d_report[1], (unsigned long long)n);
    __threadfence();
    if (atomicAdd(&d_report[2], 1ull) + 1ull == (unsigned long long)B && h_report) {   // the last query of the batch reports
      h_report[1] = ctr->n_ivf_ids;                       // posting entries the sweep read (what the level cost)
      __threadfence_system();
      h_report[0] = (min(d_report[0], 0xFFFFFFFFull) << 32) | min(d_report[1], 0xFFFFFFFFull);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// S4  approximate score: one wave per candidate document (search.rs:305-324).
//   score(d) = sum_q max_{c in codes(d)} QC[q, c]        max over tokens == max over DISTINCT codes
// QCT[b][c][:] is one contiguous row.  The chip's limit for this gather is vector-memory INSTRUCTIONS
// (a 64-lane 4-byte load costs the texture addresser as much as a 64-lane 16-byte load; measured:
// ~40 G row-gathers/s whether the table sits in L2, in the Infinity Cache, or the request is halved),
// so each lane fetches a float4: LPR lanes cover one row, one instruction fetches 64/LPR rows.
// max uses '>' semantics (NaN ignored, +inf kept); the sum runs in q order and skips tokens whose max
// stayed -inf, so the value is bit-identical to the reference loop.
// Candidates are walked in GLOBAL order (query after query) by all waves together, so one query's table
// is hot in the Infinity Cache at a time; per candidate ONE 16-byte meta record {doc, n_codes, offset,
// doc_len} (written by compact_kernel) and a 3-stage software pipeline across the wave's candidates
// (meta of w+2nw, codes of w+nw, gathers of w) keep the dependent-load chain off the critical path.
// ---------------------------------------------------------------------------------------------
// N0 row-gather instructions from code register creg0 plus N1 from creg1 (the document's second 64-code
// chunk), all in flight together: straight-line and unconditional (a load behind a branch makes the compiler
// drain vmcnt after each one), one fence (without it the scheduler folds every load into its max: one
// destination register and s_waitcnt vmcnt(0) per load), then the running max.  Lanes past the document's
// last code carry a DUPLICATE of a valid code, so no load needs masking.  Rows are addressed as a wave-uniform
// base + 32-bit byte offset (one query's table is < 4 GiB).
template <int N>
__device__ __forceinline__ void s4_fence(float4 (&v)[N]) {
  if constexpr (N == 1) asm volatile("" : "+v"(v[0].x));
  else if constexpr (N == 2) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x));
  else if constexpr (N == 3) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x));
  else if constexpr (N == 4) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x));
  else if constexpr (N == 5) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x));
  else if constexpr (N == 6) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x));
  else if constexpr (N == 7) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x));
  else if constexpr (N == 8) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
  else if constexpr (N == 9) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x), "+v"(v[8].x));
  else if constexpr (N == 10) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x), "+v"(v[8].x), "+v"(v[9].x));
  else if constexpr (N == 11) asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x), "+v"(v[8].x), "+v"(v[9].x), "+v"(v[10].x));
  else asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x), "+v"(v[8].x), "+v"(v[9].x), "+v"(v[10].x), "+v"(v[11].x));
}

template <int LPR, int N0, int N1>
__device__ __forceinline__ void s4_gather(const char* __restrict__ Tb, uint32_t row_bytes, uint32_t col_bytes, uint32_t creg0,
                                          uint32_t creg1, int s0, int grp, float& mx, float& my, float& mz, float& mw) {
  constexpr int RPI = 64 / LPR;
  static_assert(N0 + N1 >= 1 && N0 + N1 <= 12, "fence lists at most 12 operands");
  float4 v[N0 + N1];
#pragma unroll
  for (int u = 0; u < N0; ++u) {
    const uint32_t c = (uint32_t)__shfl((int)creg0, (s0 + u * RPI + grp) & 63);
    v[u] = *reinterpret_cast<const float4*>(Tb + (c * row_bytes + col_bytes));
  }
#pragma unroll
  for (int u = 0; u < N1; ++u) {
    const uint32_t c = (uint32_t)__shfl((int)creg1, (u * RPI + grp) & 63);
    v[N0 + u] = *reinterpret_cast<const float4*>(Tb + (c * row_bytes + col_bytes));
  }
  s4_fence<N0 + N1>(v);
#pragma unroll
  for (int u = 0; u < N0 + N1; ++u) {   // fmaxf == `if v > m`: NaN never wins, +inf does
    mx = fmaxf(mx, v[u].x);
    my = fmaxf(my, v[u].y);
    mz = fmaxf(mz, v[u].z);
    mw = fmaxf(mw, v[u].w);
  }
}

// exactly ceil(nt / RPI) instructions of one 64-code chunk
template <int LPR>
__device__ __forceinline__ void s4_chunk(const char* __restrict__ Tb, uint32_t row_bytes, uint32_t col_bytes, uint32_t creg,
                                         int grp, int nt, float& mx, float& my, float& mz, float& mw) {
  constexpr int RPI = 64 / LPR;
  int s0 = 0, ni = (nt + RPI - 1) / RPI;
  for (; ni > 8; ni -= 8, s0 += 8 * RPI) s4_gather<LPR, 8, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw);
  switch (ni) {
    case 8: s4_gather<LPR, 8, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 7: s4_gather<LPR, 7, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 6: s4_gather<LPR, 6, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 5: s4_gather<LPR, 5, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 4: s4_gather<LPR, 4, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 3: s4_gather<LPR, 3, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 2: s4_gather<LPR, 2, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    case 1: s4_gather<LPR, 1, 0>(Tb, row_bytes, col_bytes, creg, 0u, s0, grp, mx, my, mz, mw); break;
    default: break;
  }
}

#define NP_S4_MAXB 256    // queries per launch (prefix / Lq tables live in LDS)
template <int LPR>        // lanes per QCT row: 4*LPR >= LQP, power of two in {8,16,32,64}
__global__ void __launch_bounds__(256) approx_kernel(const float* __restrict__ QCT, int64_t KP, int LQP,
                                                     const int32_t* __restrict__ qoff,
                                                     const uint4* __restrict__ cand_meta, const int32_t* __restrict__ n_cand,
                                                     RoundPlan rp, int round, int max_rounds,
                                                     CodeArr codes, float* __restrict__ approx,
                                                     Counters* ctr) {
  constexpr int RPI = 64 / LPR;            // rows (codes) per gather instruction
  constexpr int RW = 4 * LPR + 1;          // LDS row stride in floats (+1: conflict-free column walks)
  constexpr int DPF = 256 / LPR;         // documents per flush (32 at Lq <= 32): ~17 KB of LDS per block
  __shared__ int64_t s_prefix[NP_S4_MAXB + 1];
  __shared__ int64_t s_pbase[NP_S4_MAXB];
  __shared__ int s_lq[NP_S4_MAXB];
  __shared__ float s_rows[4][DPF * RW];    // per wave: combined per-token maxima of the last DPF documents
  __shared__ int64_t s_out[4][DPF];        // their output positions
  __shared__ int s_olq[4][DPF];            // and query lengths
  // queries [rb, re) of this round: work item w = the (w - s_prefix[b])-th record of query b, stored at pool entry
  // cand_base[b] + that (the records of a query are contiguous; survivor lists leave gaps between queries)
  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], B = rp.round_tab[2 * round + 1] - rb;
  for (int k = threadIdx.x; k < B; k += 256) {
    s_pbase[k] = rp.cand_base[rb + k];
    s_lq[k] = qoff[rb + k + 1] - qoff[rb + k];
  }
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int k = 0; k < B; ++k) {
      s_prefix[k] = run;
      run += n_cand[rb + k];
    }
    s_prefix[B] = run;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jq = lane & (LPR - 1), grp = lane / LPR;   // this lane holds q = 4*jq .. 4*jq+3 of row `grp`
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  const uint32_t col_bytes = (4 * jq < LQP) ? (uint32_t)jq * 16u : 0u;
  const int64_t total = s_prefix[B];
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave;
  float* rows = s_rows[wave];
  int bc = 0;  // monotone cursor: largest b with prefix[b] <= w (empty queries are skipped)
  auto locate = [&](int64_t w, int& bo, int64_t& io) {
    while (bc < B - 1 && w >= s_prefix[bc + 1]) ++bc;
    bo = bc;
    io = w - s_prefix[bc];
  };
  // q-ordered sums (search.rs:308-321) of the buffered documents: lane d walks document d's row, so the 32
  // dependent adds are paid once per DPF documents instead of once per document
  auto flush = [&](int nbuf) {
    if (lane < nbuf) {
      const float* r = rows + lane * RW;
      const int lq = s_olq[wave][lane];
      float score = 0.f;
      for (int q = 0; q < lq; ++q) {
        const float x = r[q];
        if (x > NP_NEG_INF) score += x;
      }
      approx[s_out[wave][lane]] = score;
    }
  };
  unsigned long long toks = 0, ucodes = 0;
  uint4 m1 = make_uint4(0, 0, 0, 0), m2 = m1;
  int b1 = 0, b2 = 0;
  int64_t i1 = 0, i2 = 0;
  uint32_t c1 = 0, c1b = 0;   // first 128 distinct codes of the next candidate (padding lanes: a duplicate)
  auto fetch_codes = [&](const uint4& m) {
    const int64_t off1 = (int64_t)m.z | ((int64_t)(m.w & 0xFF) << 32);
    const int n = (int)m.y;
    c1 = n > 0 ? codes[off1 + min(lane, n - 1)] : 0u;
    c1b = n > 64 ? codes[off1 + min(64 + lane, n - 1)] : 0u;
  };
  if (w0 < total) {
    locate(w0, b1, i1);
    m1 = cand_meta[s_pbase[b1] + i1];
  }
  if (w0 + nw < total) {
    locate(w0 + nw, b2, i2);
    m2 = cand_meta[s_pbase[b2] + i2];
  }
  if (w0 < total) fetch_codes(m1);
  int nbuf = 0;
  for (int64_t w = w0; w < total; w += nw) {
    const uint4 m0 = m1;
    const int b0 = b1;
    const int64_t i0 = i1;
    const uint32_t creg0 = c1, creg1 = c1b;
    m1 = m2; b1 = b2; i1 = i2;
    if (w + 2 * nw < total) {
      locate(w + 2 * nw, b2, i2);
      m2 = cand_meta[s_pbase[b2] + i2];
    }
    if (w + nw < total) fetch_codes(m1);
    const int64_t off = (int64_t)m0.z | ((int64_t)(m0.w & 0xFF) << 32);
    const int len = (int)m0.y;
    const char* Tb = reinterpret_cast<const char*>(QCT + (int64_t)(rb + b0) * KP * LQP);
    toks += (unsigned long long)(m0.w >> 8);
    ucodes += (unsigned long long)len;
    float mx = NP_NEG_INF, my = NP_NEG_INF, mz = NP_NEG_INF, mw = NP_NEG_INF;
    {
      const int n1 = len - 64;               // codes in the second chunk
      const int ni1 = (n1 + RPI - 1) / RPI;
      int t0 = 0;
      if (RPI == 8 && len > 64 && ni1 <= 4) {
        // first chunk (8 instructions) and the short second chunk in flight together
        switch (ni1) {
          case 1: s4_gather<LPR, 8, 1>(Tb, row_bytes, col_bytes, creg0, creg1, 0, grp, mx, my, mz, mw); break;
          case 2: s4_gather<LPR, 8, 2>(Tb, row_bytes, col_bytes, creg0, creg1, 0, grp, mx, my, mz, mw); break;
          case 3: s4_gather<LPR, 8, 3>(Tb, row_bytes, col_bytes, creg0, creg1, 0, grp, mx, my, mz, mw); break;
          default: s4_gather<LPR, 8, 4>(Tb, row_bytes, col_bytes, creg0, creg1, 0, grp, mx, my, mz, mw); break;
        }
        t0 = 128;
      }
      for (; t0 < len; t0 += 64) {
        const uint32_t creg = (t0 == 0) ? creg0 : (t0 == 64 ? creg1 : codes[off + min(t0 + lane, len - 1)]);
        s4_chunk<LPR>(Tb, row_bytes, col_bytes, creg, grp, min(64, len - t0), mx, my, mz, mw);
      }
    }
    // combine the RPI row groups: lanes with equal jq
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      my = fmaxf(my, __shfl_xor(my, o));
      mz = fmaxf(mz, __shfl_xor(mz, o));
      mw = fmaxf(mw, __shfl_xor(mw, o));
    }
    if (lane < LPR) {
      float* r = rows + nbuf * RW + 4 * jq;
      r[0] = mx; r[1] = my; r[2] = mz; r[3] = mw;
    }
    if (lane == 0) {
      s_out[wave][nbuf] = s_pbase[b0] + i0;
      s_olq[wave][nbuf] = s_lq[b0];
    }
    if (++nbuf == DPF) {
      flush(nbuf);
      nbuf = 0;
    }
  }
  flush(nbuf);
  if (lane == 0 && toks && ctr) {
    atomicAdd(&ctr->n_cand_tokens, toks);
    atomicAdd(&ctr->n_cand_codes, ucodes);
  }
}

// ---------------------------------------------------------------------------------------------
// S4, L2-resident variant (same value, bit for bit, as approx_kernel).
// Measured on MI355X (tools/probes/gather_probe.hip): random 128-byte row gathers run at ~250 G rows/s when the
// table slice sits in the XCD's 4 MiB L2 and at ~57 G rows/s when they miss it, whatever the occupancy or the row
// size: the miss path is request-rate bound.  One query's table is 8.4 MB and every XCD touching it refills it,
// so approx_kernel (all XCDs on one query) runs at the miss rate.  Here ONE XCD owns a query (workgroup w lands
// on XCD w % 8; query b is walked by the workgroups with w % 8 == b % 8) and walks it in P phases: in phase p
// every wave of the XCD gathers only the codes inside the p-th 1/P of the centroid range (an 8.4/P MB slice of
// the table), for all of its documents, then moves on, so a row is fetched from the fabric once and reused
// ~19x out of L2.  Each LPR-lane group owns up to NP_S4X_MAXD documents of the query (no cross-lane fold per
// phase); their running maxima wait in LDS between phases (thread-private slots, nothing shared).  A group's
// three code registers hold its segment of the document's sorted distinct-code list (useg gives the
// boundaries); positions past the segment hold a duplicate of one of the SAME document's codes, which cannot
// change a max, so no load is predicated and the position counter is wave-uniform.
// SWZ: broadcast a code inside the 8-lane group with ds_swizzle (no address VALU) instead of ds_bpermute.
// ---------------------------------------------------------------------------------------------
#define NP_S4X_MAXD 6
template <int LPR, bool SWZ>
__global__ void __launch_bounds__(256) approx_xcd_kernel(const float* __restrict__ QCT, int64_t KP, int LQP,
                                                         const int32_t* __restrict__ qoff,
                                                         const uint4* __restrict__ cand_meta,
                                                         const int32_t* __restrict__ n_cand, RoundPlan rp, int round,
                                                         int max_rounds,
                                                         CodeArr codes, int64_t T /* entries of `codes` */,
                                                         const uint4* __restrict__ useg, float* __restrict__ approx,
                                                         int pshift, Counters* ctr) {
  static_assert(!SWZ || LPR == 8, "the swizzle pattern broadcasts inside 8-lane groups");
  constexpr int RPI = 64 / LPR;   // groups (documents) per wave
  constexpr int GPB = 4 * RPI;    // groups per workgroup
  constexpr int CAP = 3 * LPR;    // codes of one segment held in registers
  __shared__ float4 s_state[NP_S4X_MAXD][256];
  __shared__ int64_t s_off[NP_S4X_MAXD][GPB];
  __shared__ uint16_t s_end[NP_S4X_MAXD][GPB][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & (LPR - 1), grp = lane / LPR, g = wave * RPI + grp;
  const int x = blockIdx.x & 7, NBX = gridDim.x >> 3;
  const int64_t NG = (int64_t)NBX * GPB;
  const int64_t gid = (int64_t)(blockIdx.x >> 3) * GPB + g;
  const int P = 8 >> pshift;
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  const uint32_t col_bytes = (4 * jq < LQP) ? (uint32_t)jq * 16u : 0u;
  const int lbase = grp * LPR;
  unsigned long long toks = 0, ucnt = 0;

  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], re = rp.round_tab[2 * round + 1];
  for (int b = rb + x; b < re; b += 8) {
    const int64_t n = n_cand[b];
    const char* Tb = reinterpret_cast<const char*>(QCT + (int64_t)b * KP * LQP);
    const int lq = qoff[b + 1] - qoff[b];
    const int64_t pbase = rp.cand_base[b];
    const uint4* metab = cand_meta + pbase;
    for (int64_t tile0 = 0; tile0 < n; tile0 += NG * NP_S4X_MAXD) {
      const int d = (int)min((int64_t)NP_S4X_MAXD, (n - tile0 + NG - 1) / NG);
      __syncthreads();   // previous tile's LDS slots are free
      for (int k = 0; k < d; ++k) {
        const int64_t i = tile0 + (int64_t)k * NG + gid;
        s_state[k][tid] = make_float4(NP_NEG_INF, NP_NEG_INF, NP_NEG_INF, NP_NEG_INF);
        if (jq == 0) {
          int64_t off = 0;
          uint4 sg = make_uint4(0, 0, 0, 0);
          if (i < n) {
            const uint4 m = metab[i];
            off = (int64_t)m.z | ((int64_t)(m.w & 0xFF) << 32);
            sg = useg[m.x];
            toks += (unsigned long long)(m.w >> 8);
            ucnt += (unsigned long long)m.y;
          }
          s_off[k][g] = off;
          *reinterpret_cast<uint4*>(&s_end[k][g][0]) = sg;
        }
      }
      __syncthreads();
      // segment of (slot k, phase p): [s, e) inside the document's sorted distinct-code list
      uint32_t cn0 = 0, cn1 = 0, cn2 = 0;
      int ns = 0, ne = 0;
      int64_t noff = 0;
      auto seg_of = [&](int k, int p, int& so, int& eo, int64_t& oo) {
        const int hi = ((p + 1) << pshift) - 1;
        eo = (int)s_end[k][g][hi];
        so = p ? (int)s_end[k][g][(p << pshift) - 1] : 0;
        oo = s_off[k][g];
      };
      auto code_at = [&](int64_t o, int pos, int top) -> uint32_t {   // past the segment: any code of the same document
        const int64_t a = o + (int64_t)max(min(pos, top), 0);
        return codes[min(a, T - 1)];
      };
      seg_of(0, 0, ns, ne, noff);
      cn0 = code_at(noff, ns + jq, ne - 1);
      cn1 = code_at(noff, ns + LPR + jq, ne - 1);
      cn2 = code_at(noff, ns + 2 * LPR + jq, ne - 1);
      for (int p = 0; p < P; ++p) {
        for (int k = 0; k < d; ++k) {
          uint32_t c0 = cn0, c1 = cn1, c2 = cn2;
          const int s = ns, e = ne;
          const int64_t off = noff;
          {
            int kn = k + 1, pn = p;
            if (kn == d) { kn = 0; pn = p + 1; }
            if (pn < P) {
              seg_of(kn, pn, ns, ne, noff);
              cn0 = code_at(noff, ns + jq, ne - 1);
              cn1 = code_at(noff, ns + LPR + jq, ne - 1);
              cn2 = code_at(noff, ns + 2 * LPR + jq, ne - 1);
            }
          }
          int lmax = e - s;
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) lmax = max(lmax, __shfl_xor(lmax, o));
          lmax = __builtin_amdgcn_readfirstlane(lmax);
          if (lmax == 0) continue;
          float4 m = s_state[k][tid];
          int tbase = 0;
          for (int t0 = 0; t0 < lmax; t0 += 8) {
            if (t0 - tbase >= CAP) {   // a segment longer than the registers hold
              tbase = t0;
              c0 = code_at(off, s + tbase + jq, e - 1);
              c1 = code_at(off, s + tbase + LPR + jq, e - 1);
              c2 = code_at(off, s + tbase + 2 * LPR + jq, e - 1);
            }
            const int tr0 = t0 - tbase, r = tr0 / LPR;
            const uint32_t cr = r == 0 ? c0 : (r == 1 ? c1 : c2);
            uint32_t cc[8];
            if constexpr (SWZ) {
#define NP_SWZ(J) cc[J] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)cr, 0x18 | ((J) << 5))
              NP_SWZ(0); NP_SWZ(1); NP_SWZ(2); NP_SWZ(3); NP_SWZ(4); NP_SWZ(5); NP_SWZ(6); NP_SWZ(7);
#undef NP_SWZ
            } else {
              const int jb = lbase + (tr0 & (LPR - 1));
#pragma unroll
              for (int u = 0; u < 8; ++u) cc[u] = (uint32_t)__shfl((int)cr, jb + u);
            }
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(Tb + (cc[u] * row_bytes + col_bytes));
            if (t0 + 4 < lmax) {
#pragma unroll
              for (int u = 4; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(Tb + (cc[u] * row_bytes + col_bytes));
            } else {
#pragma unroll
              for (int u = 4; u < 8; ++u) v[u] = make_float4(NP_NEG_INF, NP_NEG_INF, NP_NEG_INF, NP_NEG_INF);
            }
            s4_fence<8>(v);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              m.x = fmaxf(m.x, v[u].x);
              m.y = fmaxf(m.y, v[u].y);
              m.z = fmaxf(m.z, v[u].z);
              m.w = fmaxf(m.w, v[u].w);
            }
          }
          s_state[k][tid] = m;
        }
      }
      // q-ordered sums (search.rs:308-321), every lane of the group redundantly
      for (int k = 0; k < d; ++k) {
        const int64_t i = tile0 + (int64_t)k * NG + gid;
        const float4 m = s_state[k][tid];
        float score = 0.f;
        for (int jj = 0; 4 * jj < lq; ++jj) {
          const float a0 = __shfl(m.x, lbase + jj), a1 = __shfl(m.y, lbase + jj);
          const float a2 = __shfl(m.z, lbase + jj), a3 = __shfl(m.w, lbase + jj);
          if (a0 > NP_NEG_INF) score += a0;
          if (4 * jj + 1 < lq && a1 > NP_NEG_INF) score += a1;
          if (4 * jj + 2 < lq && a2 > NP_NEG_INF) score += a2;
          if (4 * jj + 3 < lq && a3 > NP_NEG_INF) score += a3;
        }
        if (jq == 0 && i < n) approx[pbase + i] = s_end[k][g][7] ? score : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
  }
  if (lane == 0 && toks && ctr) {
    atomicAdd(&ctr->n_cand_tokens, toks);
    atomicAdd(&ctr->n_cand_codes, ucnt);
  }
}

// ---------------------------------------------------------------------------------------------
// S4, L2-resident + streamed variant (NP_S4_MODE 5..8; same value, bit for bit).
// approx_xcd_kernel advances the 8 groups of a wave in lockstep over ONE document each, so every step runs to the
// longest of 8 segments (about 1.4 rows issued per useful row) and pays one dependent code fetch per step.
// Here a group's segments of the current phase (one per owned document) form one stream: segment k occupies
// positions [pfx[k], pfx[k] + pad4(len_k)), padded with duplicates of its own last code, so every aligned
// 4-position sub-batch belongs to exactly one document.  Per phase the whole stream window is staged into LDS
// in one burst (16 independent code loads per lane, u16 code - slice base), then the wave walks
// max_g(stream length) positions: 4 row gathers per sub-batch, folded into the running maxima of that
// sub-batch's document by an LDS read-modify-write (thread-private).  Positions past a group's stream read
// row `slice base` into a trash slot, so no load is predicated.  Streams longer than the window take
// further rounds.
// ---------------------------------------------------------------------------------------------
#define NP_S4S_MAXD 5      // documents per group per tile; slot NP_S4S_MAXD is the trash slot
#define NP_S4S_CAP 128     // stream positions staged per round
template <int LPR>
__global__ void __launch_bounds__(256, 4) approx_stream_kernel(const float* __restrict__ QCT, int64_t KP, int LQP,
                                                            const int32_t* __restrict__ qoff,
                                                            const uint4* __restrict__ cand_meta,
                                                            const int32_t* __restrict__ n_cand, RoundPlan rp, int round,
                                                            int max_rounds,
                                                            CodeArr codes, int64_t T /* entries of `codes` */,
                                                            const uint4* __restrict__ useg, float* __restrict__ approx,
                                                            int pshift, uint32_t slice_w, Counters* ctr) {
  constexpr int RPI = 64 / LPR;   // groups per wave
  constexpr int GPB = 4 * RPI;    // groups per workgroup
  constexpr int NST = NP_S4S_CAP / LPR;   // staged positions per lane per round
  static_assert(LPR >= 8 && NP_S4S_MAXD <= 8, "one lane per slot builds the segment table");
  __shared__ float4 s_state[NP_S4S_MAXD + 1][256];
  __shared__ int64_t s_off[NP_S4S_MAXD][GPB];
  __shared__ uint16_t s_end[NP_S4S_MAXD][GPB][8];
  __shared__ int64_t s_abs[GPB][NP_S4S_MAXD];     // this phase: first code of each segment (absolute index)
  __shared__ int s_len[GPB][NP_S4S_MAXD];         //             its length
  __shared__ int s_pfx[GPB][NP_S4S_MAXD];         //             its first stream position
  __shared__ uint16_t s_codes[GPB][NP_S4S_CAP];   // staged window: code - slice base
  __shared__ uint8_t s_slot[GPB][NP_S4S_CAP / 4]; // document slot of every 4-position sub-batch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jq = lane & (LPR - 1), grp = lane / LPR, g = wave * RPI + grp;
  const int x = blockIdx.x & 7, NBX = gridDim.x >> 3;
  const int64_t NG = (int64_t)NBX * GPB;
  const int64_t gid = (int64_t)(blockIdx.x >> 3) * GPB + g;
  const int P = 8 >> pshift;
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  const uint32_t col_bytes = (4 * jq < LQP) ? (uint32_t)jq * 16u : 0u;
  const int lbase = grp * LPR;
  unsigned long long toks = 0, ucnt = 0;

  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], re = rp.round_tab[2 * round + 1];
  for (int b = rb + x; b < re; b += 8) {
    const int64_t n = n_cand[b];
    const char* Tb = reinterpret_cast<const char*>(QCT + (int64_t)b * KP * LQP);
    const int lq = qoff[b + 1] - qoff[b];
    const int64_t pbase = rp.cand_base[b];
    const uint4* metab = cand_meta + pbase;
    for (int64_t tile0 = 0; tile0 < n; tile0 += NG * NP_S4S_MAXD) {
      const int d = (int)min((int64_t)NP_S4S_MAXD, (n - tile0 + NG - 1) / NG);
      __syncthreads();   // previous tile's LDS slots are free
      for (int k = 0; k < d; ++k) {
        const int64_t i = tile0 + (int64_t)k * NG + gid;
        s_state[k][tid] = make_float4(NP_NEG_INF, NP_NEG_INF, NP_NEG_INF, NP_NEG_INF);
        if (jq == 0) {
          int64_t off = 0;
          uint4 sg = make_uint4(0, 0, 0, 0);
          if (i < n) {
            const uint4 m = metab[i];
            off = (int64_t)m.z | ((int64_t)(m.w & 0xFF) << 32);
            sg = useg[m.x];
            toks += (unsigned long long)(m.w >> 8);
            ucnt += (unsigned long long)m.y;
          }
          s_off[k][g] = off;
          *reinterpret_cast<uint4*>(&s_end[k][g][0]) = sg;
        }
      }
      __syncthreads();
      for (int p = 0; p < P; ++p) {
        // segment table of this phase: lane jq < d of every group describes slot jq
        int len = 0;
        int64_t abs0 = 0;
        if (jq < d) {
          const int e = (int)s_end[jq][g][((p + 1) << pshift) - 1];
          const int s = p ? (int)s_end[jq][g][(p << pshift) - 1] : 0;
          len = e - s;
          abs0 = s_off[jq][g] + s;
        }
        const int pad = (len + 3) & ~3;
        int incl = pad;
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) {
          const int v = __shfl_up(incl, o, LPR);
          if (jq >= o) incl += v;
        }
        if (jq < NP_S4S_MAXD) {
          s_pfx[g][jq] = incl - pad;
          s_abs[g][jq] = abs0;
          s_len[g][jq] = len;
        }
        const int Lg = __shfl(incl, lbase + LPR - 1);   // this group's stream length (multiple of 4)
        int maxL = Lg;
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) maxL = max(maxL, __shfl_xor(maxL, o));
        maxL = __builtin_amdgcn_readfirstlane(maxL);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int p1 = s_pfx[g][1], p2 = s_pfx[g][2], p3 = s_pfx[g][3], p4 = s_pfx[g][4];
        const uint32_t slice_lo = (uint32_t)(p << pshift) * slice_w;
        const char* Tp = Tb + (size_t)slice_lo * row_bytes;
        for (int base = 0; base < maxL; base += NP_S4S_CAP) {
          // ---- stage the window [base, base + CAP): two bursts of NST/2 independent code loads per lane
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint32_t cv[NST / 2];
#pragma unroll
            for (int i = 0; i < NST / 2; ++i) {
              const int pos = base + jq + LPR * (h * (NST / 2) + i);
              const int slot = (pos >= p1) + (pos >= p2) + (pos >= p3) + (pos >= p4);
              const int idx = min(pos - s_pfx[g][slot], max(s_len[g][slot] - 1, 0));
              int64_t a = s_abs[g][slot] + (int64_t)max(idx, 0);
              a = (pos < Lg) ? min(a, T - 1) : 0;
              cv[i] = codes[a];
            }
#pragma unroll
            for (int i = 0; i < NST / 2; ++i) {
              const int w = jq + LPR * (h * (NST / 2) + i);   // position inside the window
              const int pos = base + w;
              const bool valid = pos < Lg;
              const int slot = (pos >= p1) + (pos >= p2) + (pos >= p3) + (pos >= p4);
              s_codes[g][w] = valid ? (uint16_t)(cv[i] - slice_lo) : (uint16_t)0;
              if ((pos & 3) == 0) s_slot[g][w >> 2] = valid ? (uint8_t)slot : (uint8_t)NP_S4S_MAXD;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // ---- walk it: 4 positions = one document
          const int nwin = min(NP_S4S_CAP, maxL - base);
          for (int t0 = 0; t0 < nwin; t0 += 8) {
            const bool two = t0 + 4 < nwin;   // wave-uniform
            const uint2 ca = *reinterpret_cast<const uint2*>(&s_codes[g][t0]);
            const int sa = (int)s_slot[g][t0 >> 2];
            float4 v[8];
            v[0] = *reinterpret_cast<const float4*>(Tp + ((ca.x & 0xFFFFu) * row_bytes + col_bytes));
            v[1] = *reinterpret_cast<const float4*>(Tp + ((ca.x >> 16) * row_bytes + col_bytes));
            v[2] = *reinterpret_cast<const float4*>(Tp + ((ca.y & 0xFFFFu) * row_bytes + col_bytes));
            v[3] = *reinterpret_cast<const float4*>(Tp + ((ca.y >> 16) * row_bytes + col_bytes));
            int sb = NP_S4S_MAXD;
            if (two) {
              const uint2 cb = *reinterpret_cast<const uint2*>(&s_codes[g][t0 + 4]);
              sb = (int)s_slot[g][(t0 + 4) >> 2];
              v[4] = *reinterpret_cast<const float4*>(Tp + ((cb.x & 0xFFFFu) * row_bytes + col_bytes));
              v[5] = *reinterpret_cast<const float4*>(Tp + ((cb.x >> 16) * row_bytes + col_bytes));
              v[6] = *reinterpret_cast<const float4*>(Tp + ((cb.y & 0xFFFFu) * row_bytes + col_bytes));
              v[7] = *reinterpret_cast<const float4*>(Tp + ((cb.y >> 16) * row_bytes + col_bytes));
            } else {
#pragma unroll
              for (int u = 4; u < 8; ++u) v[u] = make_float4(NP_NEG_INF, NP_NEG_INF, NP_NEG_INF, NP_NEG_INF);
            }
            s4_fence<8>(v);
            float4 ma = s_state[sa][tid];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              ma.x = fmaxf(ma.x, v[u].x);
              ma.y = fmaxf(ma.y, v[u].y);
              ma.z = fmaxf(ma.z, v[u].z);
              ma.w = fmaxf(ma.w, v[u].w);
            }
            s_state[sa][tid] = ma;
            float4 mb = s_state[sb][tid];   // after the store above: sa may equal sb
#pragma unroll
            for (int u = 4; u < 8; ++u) {
              mb.x = fmaxf(mb.x, v[u].x);
              mb.y = fmaxf(mb.y, v[u].y);
              mb.z = fmaxf(mb.z, v[u].z);
              mb.w = fmaxf(mb.w, v[u].w);
            }
            s_state[sb][tid] = mb;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();   // the next round / phase overwrites the window and the tables
        }
      }
      // q-ordered sums (search.rs:308-321), every lane of the group redundantly
      for (int k = 0; k < d; ++k) {
        const int64_t i = tile0 + (int64_t)k * NG + gid;
        const float4 m = s_state[k][tid];
        float score = 0.f;
        for (int jj = 0; 4 * jj < lq; ++jj) {
          const float a0 = __shfl(m.x, lbase + jj), a1 = __shfl(m.y, lbase + jj);
          const float a2 = __shfl(m.z, lbase + jj), a3 = __shfl(m.w, lbase + jj);
          if (a0 > NP_NEG_INF) score += a0;
          if (4 * jj + 1 < lq && a1 > NP_NEG_INF) score += a1;
          if (4 * jj + 2 < lq && a2 > NP_NEG_INF) score += a2;
          if (4 * jj + 3 < lq && a3 > NP_NEG_INF) score += a3;
        }
        if (jq == 0 && i < n) approx[pbase + i] = s_end[k][g][7] ? score : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
  }
  if (lane == 0 && toks && ctr) {
    atomicAdd(&ctr->n_cand_tokens, toks);
    atomicAdd(&ctr->n_cand_codes, ucnt);
  }
}

// ---------------------------------------------------------------------------------------------
// S4, upper-bound filter (ahead of the exact kernels above; selection-preserving).
// The exact approximate score needs one 128-B f32 table row per (document, distinct code) and one query's f32
// table (8.4 MB) does not fit an XCD's 4 MB L2.  S1 also writes a u8 table (2 MB per query at Lq <= 32) whose
// entry is a monotone UPPER bound of the score: u = floor(max(x, 0) / s * 254) + 1 (round 5: positive scores only --
// half the step of rounds 2-4's floor((x / s + 1) * 127.5) + 1, i.e. half the slack in score units).  Monotone, so
//     U(d) = sum_q max_{c in codes(d)} u[q, c]     (integers, exact)
// is an upper bound of the f32 score in table units:  254 * approx(d) / s <= U + z  with z < 1 covering the rounding of u
// and of the reference's q-ordered f32 sum; and a LOWER bound needs the clipped entries: a token whose maximum is the
// bottom entry (u = 1: every code of the document scores below s / 254 against it, possibly negative) contributes at
// least -254, any other token at least u - 1, so
//     L(d) = U(d) - 254 * #{q : max_c u[q, c] = 1}   obeys   L - Lq - z <= 254 * approx(d) / s.
// The threshold of a cut rests on lower bounds and the cut itself compares upper bounds: with tau = (n_sel-th largest L)
// every document of the true top n_sel has U >= tau - (Lq + 2) -- if U(d) < tau - (Lq + 2), n_sel documents e have
// 254 approx(e) / s >= L(e) - Lq - z >= tau - Lq - z > U(d) + 2 - z > 254 approx(d) / s.  ub_cut_kernel keeps exactly
// those ("survivors", typically n_sel plus a few hundred) and only they get the exact f32 score, from which S5
// selects -- the same documents in the same order as without the filter.  (A document with a clipped token is one whose
// every code is nearly orthogonal or opposed to a query token: it merely stops counting towards the threshold.)  Random row gathers run at ~260 G rows/s
// out of L2 whatever the row size (tools/probes/gather_probe2.hip: the L2 serves one request per channel per
// clock), vs ~57 G rows/s once they miss it, so the win is the table fitting L2, not the smaller rows.
// One XCD owns a query (workgroup w -> XCD w % 8); a row is ROWB = LQP bytes = LPD lanes x 16 B, so a wave walks
// 64 / LPD documents in lockstep, 8 gathers in flight per lane, codes fetched 4 at a time (positions past a
// document's list repeat one of its own codes: no load is predicated).  Per-byte running maxima are SDWA
// v_max_u32 on byte lanes.  Queries with a non-finite value (qflag) and queries with <= n_sel candidates skip
// the filter: all their candidates survive.
// ---------------------------------------------------------------------------------------------
#define NP_UB_NBX 96       // workgroups per XCD: 3 per CU (48 KB of LDS each)

// CT = uint16_t when every code fits 16 bits (K <= 65536), else uint32_t.
// MODE 0: plain loads, every position of the lockstep walk gathers a row (positions past a list repeat its first code).
// MODE 1: as 0 with non-temporal loads of the records / code lists (read once).
// MODE 2: the table is read through a buffer descriptor and positions past a document's list carry an out-of-range
//         offset: the bounds check returns zeros (u >= 1, so 0 is the identity of the max) WITHOUT a memory request, so
//         the padding of the lockstep walk costs no L2 request slot.
// FLOOR (MODE 2): the rows of the centroids outside `warmbits` (M[c] <= Lambda2: no query token is close to
// them) are not requested, and every real token's maximum is floored at Lambda2 (u16 codes: the K-bit map is copied into
// LDS; u32 codes, K up to 2^19: the staging lanes look their codes up in the map where it lies, in the L2 -- all lookups of
// a staging batch in flight at once):
//     up(d) = sum_q max(Lambda2, max_{c kept} u[q, c]) >= U(d) >= lo(d) = sum_q max_{c kept} u[q, c].
// U[] receives `up` (the cut keeps a document on its UPPER bound) and the histogram counts `lo` (the cut's threshold may
// only rest on LOWER bounds of the n_sel-th largest U): the selection stays exact, for about half the row requests on the
// metric corpus and ~1.3x the survivors (tools/sim/s4_warm_sim.py, profiles/r04_sim_s4_warm_10m.txt).
template <int ROWB, typename CT, int MODE, int FLOOR = 0>
__global__ void __launch_bounds__(256) approx_ub_kernel(const uint8_t* __restrict__ QCU, int64_t KP,
                                                        const uint4* __restrict__ cand_meta /* records of the list to score */,
                                                        const int32_t* __restrict__ n_begin /* [B] first record (NULL = 0) */,
                                                        const int32_t* __restrict__ n_count /* [B] records to score */,
                                                        const int32_t* __restrict__ n_cand /* [B] ALL candidates of the query: skip rule */,
                                                        RoundPlan rp, int round,
                                                        int max_rounds, const CT* __restrict__ codes /* lists start 8-B aligned, array padded */,
                                                        const uint32_t* __restrict__ qflag, int n_sel,
                                                        uint16_t* __restrict__ U /* [pool] by record position */,
                                                        uint32_t* __restrict__ hist /* [B][NP_UB_BINS] or NULL */, int hshift,
                                                        uint32_t* __restrict__ cursor /* [B] zeroed: next unclaimed candidate */,
                                                        int32_t* __restrict__ slots /* [8][B+1] = -1 */,
                                                        int32_t* __restrict__ ticket /* [1] = -1 */, int B,
                                                        int steal_min /* unclaimed documents worth joining a query for */,
                                                        Counters* ctr, int count_tokens /* add the lists' token counts to ctr */,
                                                        int direct_wpq /* > 0: no hand-out, workgroups [j * wpq, (j+1) * wpq) take the
                                                                          round's j-th query (short lists: every query at once) */,
                                                        int static_claims /* waves take the claims round-robin, no cursor atomic (see
                                                                             approx_hot_kernel); queries are never shared between XCDs */,
                                                        const uint32_t* __restrict__ warmbits = nullptr /* FLOOR: [B][KP / 32] */,
                                                        const uint32_t* __restrict__ lam2_b = nullptr /* FLOOR: [B] Lambda2 */,
                                                        const int32_t* __restrict__ qoff = nullptr /* FLOOR: token offsets (Lq) */) {
  static_assert(FLOOR == 0 || (MODE == 2 && ROWB <= 64), "the floored level: bounds-checked loads");
  constexpr bool WLDS = FLOOR && sizeof(CT) == 2;   // the kept-centroid bitmap in LDS (K <= 65536) or read from memory
  constexpr int LPD = ROWB / 16;   // lanes per document (one 16-B piece of the row each)
  constexpr int DPW = 64 / LPD;    // documents per wave
  constexpr int CAP = sizeof(CT) == 2 ? 128 : 64;   // distinct codes of one document staged per pass (32 KB of LDS per
                                                    // workgroup either way; longer lists take further passes)
  constexpr int CPS = CAP / 32;                     // codes a staging lane loads (half a wave per document)
  constexpr bool NT = MODE == 1, OOB = MODE == 2;
  static_assert(ROWB == 32 || ROWB == 64 || ROWB == 128 || ROWB == 256, "row = power-of-two bytes >= LQP");
  // A document's distinct-code list is read from memory ONCE, coalesced, into LDS (half a wave per document, 16 B
  // per lane), and the gathers take their codes from there.  (Reading it 16 B at a time per lane as the walk
  // proceeds keeps ~3 lines per document live for the whole walk -- 6 MB per XCD next to the 2 MB table: measured
  // 30 M L2 misses and 8x over-fetch per launch.)
  __shared__ uint32_t s_hist[NP_UB_BINS];
  __shared__ CT s_codes[4][DPW][CAP];
  __shared__ uint32_t s_warm[WLDS ? 2048 : 4];    // FLOOR, u16 codes: the query's kept-centroid bitmap (K <= 65536)
  __shared__ int s_ndk[4][FLOOR ? DPW : 1];        // FLOOR: kept codes of each document in the staged window
  __shared__ int64_t s_cl[4][DPW];
  __shared__ int s_nd[4][DPW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & (LPD - 1), grp = lane / LPD;
  const int x = blockIdx.x & 7;
  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], re = rp.round_tab[2 * round + 1];
  unsigned long long toks = 0, ucnt = 0, ndoc = 0;
  const int half = lane >> 5, hl = lane & 31;   // staging: half-wave `half` loads 4 codes per lane of one document
  // The waves of an XCD claim the query's documents DPW at a time from a per-query cursor, so every workgroup of
  // the XCD finishes a query within one claim of the others and moves on together: with a fixed share per
  // workgroup the fast ones run ahead, two queries' tables (2 x 2 MB) are live in the 4 MB L2 and the gathers miss.
  __shared__ int s_q;
  for (int step = 0;; ++step) {
    __syncthreads();
    if (direct_wpq > 0) {
      if (step > 0) break;
      const int j = rb + (int)blockIdx.x / direct_wpq;
      if (j >= re) break;
      if (tid == 0) s_q = rp.order[j];
    } else if (tid == 0)
      s_q = xcd_next_query(slots, ticket, x, step, B, rp.order, rb, re, [&]() {
        // no query left to start: join the one with the most unclaimed documents, if that is worth pulling its table
        // into this XCD's L2 (the end of the launch otherwise waits for the XCDs that drew the last queries)
        int best = -3;
        if (static_claims) return best;
        int64_t most = steal_min;
        for (int j = rb; j < re; ++j) {
          const int b2 = rp.order[j];
          if (qflag[b2] || n_cand[b2] <= n_sel) continue;
          const int64_t n2 = n_count[b2];
          const int64_t left = n2 - (int64_t)__hip_atomic_load(&cursor[b2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (left >= most) {
            most = left;
            best = b2;
          }
        }
        return best;
      });
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(s_q);
    if (b < 0) break;
    if (qflag[b] || n_cand[b] <= n_sel) continue;   // the final cut keeps every candidate of this query
    const int64_t first = n_begin ? n_begin[b] : 0;
    const int64_t n = n_count[b];                    // records [first, first + n) of the query's list
    if (n <= 0) continue;
    const int64_t pbase = rp.cand_base[b] + first;
    const uint4* metab = cand_meta + pbase;
    const char* Tb = reinterpret_cast<const char*>(QCU + (int64_t)b * KP * ROWB) + jl * 16;
    // MODE 2: this query's table as a bounds-checked buffer (descriptor built from wave-uniform values only)
    const uint64_t tb64 = reinterpret_cast<uint64_t>(QCU + (int64_t)b * KP * ROWB);
    const uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb64);
    const uint32_t thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tb64 >> 32));
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)thi << 32) | tlo), 0, (int)(KP * ROWB), 0x00020000);
    uint32_t* hb = hist ? hist + (int64_t)b * NP_UB_BINS : nullptr;
    __syncthreads();
    if (hb)
      for (int i = tid; i < NP_UB_BINS; i += 256) s_hist[i] = 0;
    uint32_t lam2 = 0;
    int Lq = 0;
    const uint32_t* wb = nullptr;
    if constexpr (FLOOR) {
      wb = warmbits + (int64_t)b * (KP >> 5);
      if constexpr (WLDS)
        for (int i = tid; i < 2048; i += 256) s_warm[i] = i < (int)(KP >> 5) ? wb[i] : 0u;
      lam2 = lam2_b[b];
      Lq = qoff[b + 1] - qoff[b];
    }
    __syncthreads();
    // static_claims: wave g of the NWS waves working on this query takes claims g, g + NWS, ... (no cursor atomic)
    const int64_t NWS = direct_wpq > 0 ? (int64_t)direct_wpq * 4 : (int64_t)(gridDim.x >> 3) * 4;
    const int64_t gw = direct_wpq > 0 ? (int64_t)((int)blockIdx.x % direct_wpq) * 4 + wave : (int64_t)(blockIdx.x >> 3) * 4 + wave;
    int64_t istat = gw * DPW;
    uint32_t inext = 0;
    if (!static_claims && lane == 0) inext = atomicAdd(&cursor[b], (uint32_t)DPW);
    for (;;) {
      const int64_t i0 = static_claims ? istat : (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)inext);
      if (i0 >= n) break;
      istat += NWS * DPW;
      if (!static_claims && lane == 0) inext = atomicAdd(&cursor[b], (uint32_t)DPW);   // the next claim travels while this one is processed
      const int64_t i = i0 + grp;
      const bool valid = i < n;
      uint4 m;
      if constexpr (NT) {
        const uint32_t* mp = reinterpret_cast<const uint32_t*>(metab + (valid ? i : n - 1));
        m = make_uint4(__builtin_nontemporal_load(mp), __builtin_nontemporal_load(mp + 1), __builtin_nontemporal_load(mp + 2),
                       __builtin_nontemporal_load(mp + 3));
      } else {
        m = metab[valid ? i : n - 1];
      }
      const int nd = valid ? (int)m.y : 0;
      const int64_t cl = (int64_t)m.z | ((int64_t)(m.w & 0xFF) << 32);
      if (jl == 0) {
        s_cl[wave][grp] = cl;
        s_nd[wave][grp] = nd;
        if (valid) {
          toks += (unsigned long long)(m.w >> 8);
          if constexpr (!FLOOR) ucnt += (unsigned long long)nd;   // FLOOR: the rows actually requested, counted per window below
          ++ndoc;
        }
      }
      int nmax = nd;
#pragma unroll
      for (int o = LPD; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o));
      nmax = __builtin_amdgcn_readfirstlane(nmax);
      uint32_t st[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) st[k] = 0;
      for (int p0 = 0; p0 < nmax; p0 += CAP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // s_cl / s_nd written, the previous pass's codes consumed
        // ---- stage codes [p0, p0 + CAP) of every document of the wave; positions past a list hold the list's
        // first code (a duplicate cannot change a max), so the walk below is unpredicated
        // (all loads of a batch of documents are issued before the first LDS write: one memory round trip per batch,
        // not one per document)
        constexpr int SB = DPW / 2 < 8 ? DPW / 2 : 8;   // staging steps per batch (two documents per step)
#pragma unroll 1
        for (int sb = 0; sb < DPW / 2; sb += SB) {
          uint32_t cv[SB][4], c0[SB], wvg[(FLOOR && !WLDS) ? SB : 1][CPS];
          int nds[SB];
#pragma unroll
          for (int j = 0; j < SB; ++j) {
            const int sl = 2 * (sb + j) + half;
            nds[j] = s_nd[wave][sl];
            const CT* cp = codes + s_cl[wave][sl];
            // clamped to the list's last CPS-aligned group: one aligned 8-byte load (lists start 8-B aligned, the array is
            // padded), at most CPS - 1 entries past the list's end
            const int pos = min(p0 + CPS * hl, max(nds[j] - 1, 0) & ~(CPS - 1));
            if constexpr (NT) {
              c0[j] = (uint32_t)__builtin_nontemporal_load(cp);
#pragma unroll
              for (int k = 0; k < CPS; ++k) cv[j][k] = (uint32_t)__builtin_nontemporal_load(cp + pos + k);
            } else {
              c0[j] = (uint32_t)cp[0];
              const uint2 raw = *reinterpret_cast<const uint2*>(cp + pos);
              if constexpr (sizeof(CT) == 2) {
                cv[j][0] = raw.x & 0xFFFFu; cv[j][1] = raw.x >> 16; cv[j][2] = raw.y & 0xFFFFu; cv[j][3] = raw.y >> 16;
              } else {
                cv[j][0] = raw.x; cv[j][1] = raw.y;
              }
            }
          }
          if constexpr (FLOOR && !WLDS) {   // every bitmap word of the batch requested before the first one is used
            // (only for slots inside the list, and clamped to the map: what follows a list in memory -- padding, the next block's
            // header, another overflow list -- must never become an address)
            const uint32_t wlast = (uint32_t)(KP >> 5) - 1u;
#pragma unroll
            for (int j = 0; j < SB; ++j)
#pragma unroll
              for (int k = 0; k < CPS; ++k)
                wvg[j][k] = (p0 + CPS * hl + k < nds[j]) ? wb[min(cv[j][k] >> 5, wlast)] : 0u;
          }
#pragma unroll
          for (int j = 0; j < SB; ++j) {
            const int sl = 2 * (sb + j) + half;
            const int pos = p0 + CPS * hl;   // the position this LDS slot stands for (the load above may have been clamped)
            if constexpr (FLOOR) {
              // keep only the codes of centroids in the query's bitmap, compacted per document (order is irrelevant to a
              // max): one ballot per code slot, the half-wave of the document counts its own 32 bits
              uint32_t wv[CPS];
#pragma unroll
              for (int k = 0; k < CPS; ++k) {
                if constexpr (WLDS) wv[k] = s_warm[(cv[j][k] >> 5) & 0x7FFu];
                else wv[k] = wvg[j][k];
              }
              int base = 0;
              CT* dstc = &s_codes[wave][sl][0];
#pragma unroll
              for (int k = 0; k < CPS; ++k) {
                const bool kp = pos + k < nds[j] && ((wv[k] >> (cv[j][k] & 31u)) & 1u) != 0u;
                const unsigned long long bal = __ballot(kp);
                const uint32_t hm = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                if (kp) dstc[base + (int)__popc(hm & ((1u << hl) - 1u))] = (CT)cv[j][k];
                base += (int)__popc(hm);
              }
              if (hl == 0) s_ndk[wave][sl] = base;
              __builtin_amdgcn_sched_barrier(0);   // one document pair at a time: hoisting every lookup and ballot spills
              continue;
            }
#pragma unroll
            for (int k = 0; k < CPS; ++k)
              if (pos + k >= nds[j]) cv[j][k] = c0[j];
            CT* dst = &s_codes[wave][sl][CPS * hl];
            if constexpr (sizeof(CT) == 2) {
              *reinterpret_cast<uint2*>(dst) = make_uint2(cv[j][0] | (cv[j][1] << 16), cv[j][2] | (cv[j][3] << 16));
            } else {
              *reinterpret_cast<uint2*>(dst) = make_uint2(cv[j][0], cv[j][1]);
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- walk: two batches of 8 gathers in flight per lane: the per-byte maxima of batch A are folded while
        // batch B's rows are on their way (a wave alone waits ~3 us for a batch at this occupancy).  Documents are
        // walked in lockstep from position 0 of their SORTED lists, so the rows a wave wants at the same time sit
        // in a narrow band of the table (measured: starting every document at a different position costs 5 %).
        int np = (min(CAP, nmax - p0) + 7) & ~7;
        int ndk = 0;
        if constexpr (FLOOR) {
          ndk = s_ndk[wave][grp];
          if (jl == 0 && valid) ucnt += (unsigned long long)ndk;
          int km = ndk;
#pragma unroll
          for (int o = LPD; o < 64; o <<= 1) km = max(km, __shfl_xor(km, o));
          np = (__builtin_amdgcn_readfirstlane(km) + 7) & ~7;
        }
        const CT* mine = &s_codes[wave][grp][0];
        auto issue = [&](int t, uint4 (&v)[8]) {
          uint32_t c[8];
          if constexpr (sizeof(CT) == 2) {
            const uint4 cw = *reinterpret_cast<const uint4*>(mine + t);
            c[0] = cw.x & 0xFFFFu; c[1] = cw.x >> 16; c[2] = cw.y & 0xFFFFu; c[3] = cw.y >> 16;
            c[4] = cw.z & 0xFFFFu; c[5] = cw.z >> 16; c[6] = cw.w & 0xFFFFu; c[7] = cw.w >> 16;
          } else {
            const uint4 ca = *reinterpret_cast<const uint4*>(mine + t), cb = *reinterpret_cast<const uint4*>(mine + t + 4);
            c[0] = ca.x; c[1] = ca.y; c[2] = ca.z; c[3] = ca.w;
            c[4] = cb.x; c[5] = cb.y; c[6] = cb.z; c[7] = cb.w;
          }
          if constexpr (OOB) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              // FLOOR: the staged window holds only the ndk KEPT codes of this pass, compacted
              const bool live = FLOOR ? (t + k < ndk) : (p0 + t + k < nd);
              const uint32_t off = live ? c[k] * (uint32_t)ROWB + (uint32_t)(jl * 16) : 0x7FFFFFF0u;
              const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)off, 0, 0);
              v[k] = make_uint4(r.x, r.y, r.z, r.w);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint4*>(Tb + (size_t)c[k] * ROWB);
          }
        };
        auto fold = [&](uint4 (&v)[8]) {
          asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) st[4 * j + e] = max(st[4 * j + e], (w4[j] >> (8 * e)) & 0xFFu);
          }
        };
        uint4 va[8], vb[8];
        issue(0, va);
        int t = 0;
        for (; t + 16 < np; t += 16) {   // steady state, no branch between an issue and the fold before it
          issue(t + 8, vb);
          fold(va);
          issue(t + 16, va);
          fold(vb);
        }
        if (t + 8 < np) {                // va holds batch t; one more batch
          issue(t + 8, vb);
          fold(va);
          fold(vb);
        } else {
          fold(va);
        }
      }
      // sum = the upper bound; slo = the LOWER bound the histogram counts: a real token whose maximum is the table's bottom
      // entry (1; with FLOOR also 0: none of its rows was requested) may score anything down to -s, i.e. 254 units lower
      uint32_t sum = 0;
      int slo = 0;
      if constexpr (FLOOR) {
        // byte k of lane jl is query token 16 jl + k; padding tokens (>= Lq) hold 0 in every row and stay 0.  (The floor is
        // fenced: left visible, its 16 per-byte values are hoisted out of the claim loop and live across the walk.)
        uint32_t fl = lam2;
        int lq_here = Lq - 16 * jl;
        asm volatile("" : "+v"(fl), "+v"(lq_here));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          slo += (int)st[k] - ((k < lq_here && st[k] <= 1u) ? 254 : 0);
          sum += max(st[k], k < lq_here ? fl : 0u);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          sum += st[k];
          slo += (int)st[k] - (st[k] == 1u ? 254 : 0);   // real tokens hold >= 1 (the list is not empty), padding tokens 0
        }
      }
#pragma unroll
      for (int o = 1; o < LPD; o <<= 1) {
        sum += (uint32_t)__shfl_xor((int)sum, o);
        slo += __shfl_xor(slo, o);
      }
      const uint32_t sum_lo = (uint32_t)max(slo, 0);
      if (valid && jl == 0) {
        U[pbase + i] = (uint16_t)sum;
        if (hb) atomicAdd(&s_hist[min(sum_lo >> hshift, (uint32_t)(NP_UB_BINS - 1))], 1u);
      }
      __builtin_amdgcn_wave_barrier();   // s_cl / s_nd of this group are rewritten by the next one
    }
    __syncthreads();
    if (hb)
      for (int i = tid; i < NP_UB_BINS; i += 256) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&hb[i], v);
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
    ndoc += __shfl_xor(ndoc, o);
  }
  // one set of (same-address) memory atomics per workgroup, not per wave
  __shared__ unsigned long long s_cnt[3];
  __syncthreads();
  if (tid == 0) s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
  __syncthreads();
  if (lane == 0 && toks) {
    atomicAdd(&s_cnt[0], toks);
    atomicAdd(&s_cnt[1], ucnt);
    atomicAdd(&s_cnt[2], ndoc);
  }
  __syncthreads();
  if (tid == 0 && s_cnt[0] && ctr) {
    if (count_tokens) {   // the single-level filter: every candidate passes through here once
      atomicAdd(&ctr->n_cand_tokens, s_cnt[0]);
      atomicAdd(&ctr->n_cand_dcodes, s_cnt[1]);
    } else {              // a list of the two-level filter
      atomicAdd(&ctr->n_level2, s_cnt[2]);
    }
    atomicAdd(&ctr->n_cand_codes, s_cnt[1]);      // table rows gathered
  }
}

// Per query: the histogram bin of the n_sel-th largest bound minus the slack (see above) -> thr[b]; 0 = "the filter does
// not apply" (flagged queries, queries with <= n_sel candidates, or a cut that reaches bin 0).  One block per query.
__global__ void __launch_bounds__(256) ub_thr_kernel(const uint32_t* __restrict__ hist, int hshift, int slack, int n_sel,
                                                     const int32_t* __restrict__ n_cand, RoundPlan rp, int round,
                                                     const uint32_t* __restrict__ qflag, uint32_t* __restrict__ thr) {
  __shared__ uint32_t s_part[256];
  constexpr int BPT = NP_UB_BINS / 256;   // bins per thread
  const int b = blockIdx.x, tid = threadIdx.x;
  if (rp.round_of[b] != round) return;
  if (qflag[b] != 0 || n_cand[b] <= n_sel) {
    if (tid == 0) thr[b] = 0;
    return;
  }
  // bins from the top: thread t owns bins [(255 - t) * BPT, +BPT)
  const uint32_t* hb = hist + (int64_t)b * NP_UB_BINS;
  const int top = (255 - tid) * BPT;
  uint32_t mine = 0;
#pragma unroll
  for (int k4 = 0; k4 < BPT / 4; ++k4) {
    const uint4 v = *reinterpret_cast<const uint4*>(hb + top + 4 * k4);
    mine += v.x + v.y + v.z + v.w;
  }
  s_part[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t cum = 0;
    int t = 0;
    for (; t < 255; ++t) {
      if (cum + s_part[t] >= (uint32_t)n_sel) break;
      cum += s_part[t];
    }
    int bin = (255 - t) * BPT + BPT - 1;
    for (; bin > (255 - t) * BPT; --bin) {
      if (cum + hb[bin] >= (uint32_t)n_sel) break;
      cum += hb[bin];
    }
    // bins are U >> hshift: the slack in bins is rounded up
    thr[b] = (uint32_t)max(0, bin - ((slack + (1 << hshift) - 1) >> hshift));
  }
}

// Cut by bound: appends the records of `src` whose bin is in [lo[b], hi[b]) to dst.  grid (blocks per query, B).  Records
// are appended in arbitrary order (S5 orders by (score, doc id) itself).  Uses:
//   single-level filter   src = all candidates, U = exact bound, zero_mode 0                      -> survivors
//   two-level, list 1     src = all candidates, U = hot bound U', lo = thr1, zero_mode 1          -> S1 (the n_sel best by U')
//   two-level, list 2     the same with lo = thr2, hi = thr1, appended behind list 1             -> S2 (U' >= tau, not in S1)
//   two-level, final      src = list 1 + list 2, U = exact bound by list position, lo = thr2,
//                         zero_mode 0 with all_src = all candidates                              -> survivors
struct CutP {
  const uint4* src;          // records to test, at src[cand_base[b] + i]
  const int32_t* n_src_a;    // [B] number of them ...
  const int32_t* n_src_b;    // [B] ... plus this (NULL = 0)
  const uint16_t* U;         // bound of record i at U[cand_base[b] + i]
  const uint32_t* lo;        // [B] keep bin >= lo
  const uint32_t* hi;        // [B] and bin < hi (NULL: no upper limit)
  int zero_mode;             // lo[b] == 0 ("filter does not apply"): 0 = keep every record of all_src, 1 = keep none
  const uint4* all_src;      // zero_mode 0 source: every candidate of the query
  const int32_t* n_all;
  uint4* dst;                // appended at dst[cand_base[b] + dst_begin[b] + ...]
  const int32_t* dst_begin;  // NULL = 0
  int32_t* n_dst;            // [B] zeroed
  int hshift;
  Counters* ctr;             // work counters of the queries the filter skipped (zero_mode 0 only)
};

__global__ void __launch_bounds__(256) ub_cut_kernel(CutP p, RoundPlan rp, int round) {
  __shared__ int s_wcnt[4], s_base;
  __shared__ unsigned long long s_cnt[2];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (rp.round_of[b] != round) return;
  const uint32_t lo = p.lo[b], hi = p.hi ? p.hi[b] : 0xFFFFFFFFu;
  const bool all = lo == 0;
  if (all && p.zero_mode == 1) return;
  const uint4* src = all ? p.all_src : p.src;
  const int64_t n = all ? (int64_t)p.n_all[b] : (int64_t)p.n_src_a[b] + (p.n_src_b ? p.n_src_b[b] : 0);
  const int64_t pbase = rp.cand_base[b];
  const int64_t dbase = pbase + (p.dst_begin ? p.dst_begin[b] : 0);
  if (tid == 0) s_cnt[0] = s_cnt[1] = 0;
  unsigned long long toks = 0, ucnt = 0;
  // 8 records per thread and step: one append (same-address atomic) and two barriers per 2048 records
  constexpr int PT = 8;
  for (int64_t i0 = (int64_t)blockIdx.x * (256 * PT); i0 < n; i0 += (int64_t)gridDim.x * (256 * PT)) {   // block-uniform trip count
    bool keep[PT];
    unsigned long long bal[PT];
    int wtot = 0;
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int64_t i = i0 + k * 256 + tid;
      bool kp = i < n;
      if (kp && !all) {
        const uint32_t bin = min((uint32_t)p.U[pbase + i] >> p.hshift, (uint32_t)(NP_UB_BINS - 1));
        kp = bin >= lo && bin < hi;
      }
      keep[k] = kp;
      bal[k] = __ballot(kp);
      wtot += (int)__popcll(bal[k]);
    }
    if (lane == 0) s_wcnt[wave] = wtot;
    __syncthreads();
    if (tid == 0) {
      const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
      s_base = tot ? atomicAdd(&p.n_dst[b], tot) : 0;
    }
    __syncthreads();
    int pos = s_base;
    for (int k = 0; k < wave; ++k) pos += s_wcnt[k];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (keep[k]) {
        const uint4 m = src[pbase + i0 + k * 256 + tid];
        p.dst[dbase + pos + (int)__popcll(bal[k] & ((1ull << lane) - 1ull))] = m;
        if (all) {
          toks += (unsigned long long)(m.w >> 8);
          ucnt += (unsigned long long)m.y;
        }
      }
      pos += (int)__popcll(bal[k]);
    }
    __syncthreads();
  }
  if (all && p.ctr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      toks += __shfl_xor(toks, o);
      ucnt += __shfl_xor(ucnt, o);
    }
    if (lane == 0 && toks) {
      atomicAdd(&s_cnt[0], toks);
      atomicAdd(&s_cnt[1], ucnt);
    }
    __syncthreads();
    if (tid == 0 && s_cnt[0]) {
      atomicAdd(&p.ctr->n_cand_tokens, s_cnt[0]);
      atomicAdd(&p.ctr->n_cand_dcodes, s_cnt[1]);
      atomicAdd(&p.ctr->n_cand_codes, s_cnt[1]);   // these candidates all reach the f32 pass
    }
  }
}

// ---------------------------------------------------------------------------------------------
// S4, first filter level: the HOT bound (round 3).
// The exact bound U(d) costs one L2 row request per (document, distinct code): 813 M per batch at 10 M documents, and the L2
// serves ~200-266 G requests/s whatever the row size -- the stage's ceiling.  Most of those rows are noise: a random
// centroid scores N(0, 1/sqrt(dim)) against every query token.  Per query let M[c] = max_q u[q, c] (hot_prep_kernel) and
// call the `hot_permille` centroids with the largest M HOT (M[c] > Lambda).  For a cold centroid u[q, c] <= Lambda for
// every token, so
//     U'(d) = sum_q max( Lambda, max_{c in codes(d), c hot} u[q, c] )  >=  U(d)
// and it needs table rows for the document's hot codes only (~5-10 of ~68 at 8-12 % hot).  Whether a code is hot is one
// bit of a K-bit map held in LDS (8 KB at K = 65536), so the scan of a document's code list issues no L2 row request at
// all.  U' feeds the three-step cut of np_search.hip: S1 = the n_sel documents with the largest U' get the exact bound,
// tau = (their n_sel-th largest exact U) - slack is a valid cut (any n_sel documents give a lower bound of the n_sel-th
// best score), S2 = the other documents with U' >= tau get the exact bound too, and the survivors are the documents of
// S1 + S2 with exact U >= tau -- a superset of the true top n_sel, like the single-level filter's.  CPU simulation on the
// metric corpus (tools/sim/s4_hot_sim.py): 80-90 % of the candidates drop out at the first level, none of the exact
// filter's survivors is lost.
// Work split as in approx_ub_kernel (one XCD per query, documents claimed DPW at a time).  Per claim a wave (1) stages
// the claim's code lists into LDS exactly like the exact kernel (one coalesced read per list, a batch of lists in
// flight), (2) tests every staged code against the bitmap -- the whole wave on one document, ballot / prefix-count
// compaction of the hot codes to the front of the same LDS row -- and (3) walks the hot codes in lockstep like the
// exact kernel walks full lists: LPD lanes per 16-B piece of a row, 8 gathers in flight per lane through the
// bounds-checked buffer descriptor (positions past a hot list issue no request).  Lists longer than the staged window
// take further passes; the per-byte running maxima carry over.  (First version, measured at 10 M documents: scanning
// straight from memory four documents at a time was latency-bound, 5.5 ms per batch for 2.9x fewer table rows.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hot_prep_kernel(const uint8_t* __restrict__ QCU, int64_t K, int64_t KP, int RB,
                                                       uint8_t* __restrict__ cmaxu /* [B][KP] */,
                                                       uint32_t* __restrict__ chist /* [B][256] zeroed */) {
  __shared__ uint32_t s_h[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  s_h[tid] = 0;
  __syncthreads();
  for (int64_t c = (int64_t)blockIdx.x * 256 + tid; c < KP; c += (int64_t)gridDim.x * 256) {
    const uint4* row = reinterpret_cast<const uint4*>(QCU + ((int64_t)b * KP + c) * RB);
    uint32_t m = 0;
    for (int j = 0; j < RB / 16; ++j) {
      const uint4 v = row[j];
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(m, (w4[e] >> (8 * k)) & 0xFFu);
    }
    cmaxu[(int64_t)b * KP + c] = (uint8_t)m;
    if (c < K) atomicAdd(&s_h[m], 1u);
  }
  __syncthreads();
  if (s_h[tid]) atomicAdd(&chist[(int64_t)b * 256 + tid], s_h[tid]);
}

// Lambda of every query: the smallest level with at most hot_permille of the centroids above it.  {v in [1, 255] :
// #(M >= v) <= limit} is an upper interval [v0, 255], Lambda = v0 - 1 = 255 - its size.  One block per query.
__global__ void __launch_bounds__(256) hot_lam_kernel(const uint32_t* __restrict__ chist, int64_t K, int hot_permille,
                                                      uint32_t* __restrict__ lam) {
  __shared__ uint32_t s_h[256];
  __shared__ uint32_t s_wsum[4];
  __shared__ int s_ok;
  const int b = blockIdx.x, v = threadIdx.x, lane = v & 63, wave = v >> 6;
  if (v == 0) s_ok = 0;
  uint32_t suf = chist[(int64_t)b * 256 + v];   // -> #(M >= v): suffix sum over the block
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_down((int)suf, o);
    if (lane + o < 64) suf += t;
  }
  if (lane == 0) s_wsum[wave] = suf;
  __syncthreads();
  for (int k = wave + 1; k < 4; ++k) suf += s_wsum[k];
  (void)s_h;
  const uint64_t limit = (uint64_t)K * (uint64_t)hot_permille / 1000u;
  const bool ok = v >= 1 && (uint64_t)suf <= limit;
  const int cnt = (int)__popcll(__ballot(ok));
  if (lane == 0 && cnt) atomicAdd(&s_ok, cnt);
  __syncthreads();
  if (v == 0) lam[b] = (uint32_t)(255 - s_ok);
}

template <int ROWB, typename CT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) approx_hot_kernel(
    const uint8_t* __restrict__ QCU, int64_t K, int64_t KP, const uint8_t* __restrict__ cmaxu, const uint32_t* __restrict__ lam_b,
    const uint32_t* __restrict__ cand_ids /* [pool] shard-local document ids (compact_kernel) */,
    uint4* __restrict__ cand_meta /* [pool] OUT: the candidates' 16-B records, for the cuts and the exact level */,
    int ublock_stride /* entries per document block of `codes` */, int64_t ovf_base /* first entry of the overflow region */,
    const int32_t* __restrict__ n_cand, RoundPlan rp, int round, int max_rounds,
    const CT* __restrict__ codes, const uint32_t* __restrict__ qflag, const int32_t* __restrict__ qoff, int n_sel,
    uint16_t* __restrict__ U, uint32_t* __restrict__ hist, int hshift, uint32_t* __restrict__ cursor, int32_t* __restrict__ slots,
    int32_t* __restrict__ ticket, int B, int steal_min, Counters* ctr,
    int probe /* DIAGNOSTIC ONLY (results invalid when != 0): 1 skip the walk, 2 skip the scan, 4 skip the staging */,
    int static_claims /* 1: round-robin claims, 0: claims from the per-query cursor */) {
  constexpr int LPD = ROWB / 16;   // lanes per document in the walk (one 16-B piece of the row each)
  constexpr int DPW = 64 / LPD;    // documents per claim
  constexpr int CAP = sizeof(CT) == 2 ? 128 : 64;   // codes of one document staged per pass
  constexpr int CPS = CAP / 32;                     // codes a staging lane loads (half a wave per document)
  constexpr int HDR = 16 / (int)sizeof(CT);         // entries of a block's 16-byte header {#distinct, doc length, overflow index, 0}
  constexpr int RS = HDR + CAP;                     // LDS row = a whole list block: header + up to CAP codes (rows 16-B aligned,
                                                    // 272 B apart: off each other's banks)
  extern __shared__ uint32_t s_bits[];   // KP / 32 words: hot centroids of the current query
  __shared__ uint32_t s_hist[NP_UB_BINS];
  __shared__ __attribute__((aligned(16))) CT s_codes[4][DPW][RS];
  __shared__ int64_t s_cl[4][DPW];
  __shared__ int s_nd[4][DPW];
  __shared__ uint32_t s_did[4][DPW];
  __shared__ int s_q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & (LPD - 1), grp = lane / LPD;
  const int half = lane >> 5, hl = lane & 31;
  const int x = blockIdx.x & 7;
  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], re = rp.round_tab[2 * round + 1];
  uint32_t toks32 = 0, ucnt32 = 0, rows32 = 0;   // per-lane work counters (a lane sees one document per claim: 32 bits are plenty)
  // the scan reads whole 4-code groups and looks every code up in the bitmap, also past a list's end: the staging area
  // must never hold anything but codes (< K), so it starts zeroed (afterwards it only ever receives list entries)
  for (int i = tid; i < (int)(sizeof(s_codes) / 4); i += 256) reinterpret_cast<uint32_t*>(&s_codes[0][0][0])[i] = 0u;
  for (int step = 0;; ++step) {
    __syncthreads();
    if (tid == 0)
      s_q = xcd_next_query(slots, ticket, x, step, B, rp.order, rb, re, [&]() { return -3; });   // round-robin claims: no sharing
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(s_q);
    if (b < 0) break;
    const int64_t n = n_cand[b];
    const int64_t pbase = rp.cand_base[b];
    const uint32_t* idb = cand_ids + pbase;
    uint4* metab = cand_meta + pbase;
    if (qflag[b] || n <= (int64_t)n_sel) {
      // the cuts keep every candidate of this query: no bound to compute, but the later stages still want the records
      for (int64_t i = ((int64_t)(blockIdx.x >> 3) * 256 + tid); i < n; i += (int64_t)(gridDim.x >> 3) * 256) {
        const uint32_t d = idb[i];
        const uint4 hd = *reinterpret_cast<const uint4*>(codes + (int64_t)d * ublock_stride);
        const int64_t cl = (int)hd.x > ublock_stride - HDR ? ovf_base + (int64_t)hd.z * 4 : (int64_t)d * ublock_stride + HDR;
        metab[i] = make_uint4(d, hd.x, (uint32_t)(cl & 0xFFFFFFFFll), (uint32_t)((cl >> 32) & 0xFF) | (hd.y << 8));
      }
      continue;
    }
    const int Lq = qoff[b + 1] - qoff[b];
    const int fit = ublock_stride - HDR;             // codes a block holds
    const uint32_t lam = lam_b[b];   // hot_lam_kernel
    // ---- hot bitmap: bit c = M[c] > Lambda
    {
      const uint8_t* cm = cmaxu + (int64_t)b * KP;
      for (int64_t w = tid; w < (KP >> 5); w += 256) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(cm + w * 32), v1 = *reinterpret_cast<const uint4*>(cm + w * 32 + 16);
        const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k) bits |= (((w8[e] >> (8 * k)) & 0xFFu) > lam ? 1u : 0u) << (4 * e + k);
        s_bits[w] = bits;
      }
      for (int i = tid; i < NP_UB_BINS; i += 256) s_hist[i] = 0;
    }
    __syncthreads();
    const uint64_t tb64 = reinterpret_cast<uint64_t>(QCU + (int64_t)b * KP * ROWB);
    const uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb64);
    const uint32_t thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tb64 >> 32));
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)thi << 32) | tlo), 0, (int)(KP * ROWB), 0x00020000);
    uint32_t* hb = hist + (int64_t)b * NP_UB_BINS;
    // Claims.  A claim from a per-query cursor costs one device-scope atomic on a line every XCD's cursor shares (~50 ns
    // each, serialised: with every other phase switched off, NP_S4_PROBE=7, the 372 k claims of a launch at 10 M documents
    // take 2.3 ms).  A document costs the same work to within a few per cent, so by default the waves of the XCD take the
    // query's claims ROUND-ROBIN: wave g of NW takes claims g, g + NW, ... -- no atomic, the next claim's records prefetched
    // a whole iteration ahead (hot kernel 2.06 -> 1.68 ms).  static_claims = 0 keeps the cursor (claims two ahead).
    const int64_t NW = (int64_t)(gridDim.x >> 3) * 4;
    uint32_t inext = 0;
    int64_t i0, i1;
    if (static_claims) {
      i0 = ((int64_t)(blockIdx.x >> 3) * 4 + wave) * DPW;
      i1 = i0 + NW * DPW;
    } else {
      if (lane == 0) inext = atomicAdd(&cursor[b], (uint32_t)(2 * DPW));
      i0 = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)inext);
      i1 = i0 + DPW;
    }
    uint32_t did = idb[min(i0 + grp, n - 1)];
    constexpr int EPL = 8 / (int)sizeof(CT);   // entries per lane of a block load
    for (;;) {
      if (i0 >= n) break;
      if (!static_claims && lane == 0) inext = atomicAdd(&cursor[b], (uint32_t)DPW);
      const uint32_t did_next = idb[min(i1 + grp, n - 1)];   // prefetch (clamped: a claim past the end is never used)
      const int64_t i = i0 + grp;
      const bool valid = i < n;
      // ---- (0) the claim's list BLOCKS -> LDS rows, header included, in one burst: half a wave per document, 8 bytes per
      // lane (a block is at most 256 bytes).  The block address comes from the document id alone: no per-candidate record
      // was gathered for this.  (16 bytes per lane: 1.9 instead of 1.3 ms for the whole kernel.  Issuing the next claim's
      // loads during this claim's scan and walk -- the loads held in registers across them -- measured no gain, 1.8 ms
      // with the extra register pressure, and was dropped.)
      __builtin_amdgcn_wave_barrier();       // the previous claim's rows are consumed
      if (jl == 0) s_did[wave][grp] = valid ? did : idb[n - 1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!(probe & 4)) {
        uint2 raw[DPW / 2];
#pragma unroll
        for (int j = 0; j < DPW / 2; ++j) {
          const int sl = 2 * j + half;
          const CT* bp = codes + (int64_t)s_did[wave][sl] * ublock_stride;
          raw[j] = *reinterpret_cast<const uint2*>(bp + min(EPL * hl, ublock_stride - EPL));   // lanes past the block repeat its last piece
        }
#pragma unroll
        for (int j = 0; j < DPW / 2; ++j) {
          const int sl = 2 * j + half;
          *reinterpret_cast<uint2*>(&s_codes[wave][sl][EPL * hl]) = raw[j];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint4 hd = *reinterpret_cast<const uint4*>(&s_codes[wave][grp][0]);   // {#distinct, doc length, overflow index, 0}
      const int nd = valid ? (int)hd.x : 0;
      const bool ovf = nd > fit;
      const int64_t cl = ovf ? ovf_base + (int64_t)hd.z * 4 : (int64_t)did * ublock_stride + HDR;
      if (jl == 0) {
        s_cl[wave][grp] = cl;
        s_nd[wave][grp] = nd;
        if (valid) {
          toks32 += hd.y;
          ucnt32 += (uint32_t)nd;
          metab[i] = make_uint4(did, (uint32_t)nd, (uint32_t)(cl & 0xFFFFFFFFll), (uint32_t)((cl >> 32) & 0xFF) | (hd.y << 8));
        }
      }
      const int nmax = wave_max_nonneg(nd);
      uint32_t st[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) st[k] = 0;
      for (int p0 = 0; p0 < nmax; p0 += CAP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // s_cl / s_nd written, the previous pass's lists consumed
        // lists that fit their block are already staged; only a claim with an overflow list (0.1 % of the documents) or a
        // second window of a long list goes back to memory
        const bool restage = p0 > 0 || nmax > fit;   // wave-uniform
        // ---- (1) stage codes [p0, p0 + CAP) of every document of the claim: one coalesced read per list, half a wave per
        // document, all loads of a batch of documents in flight before the first LDS write (one memory round trip per
        // batch).  Positions past a list's end hold whatever follows it in memory: the scan below knows the lengths.
        constexpr int SB = DPW / 2;   // staging steps, two documents each: the whole claim in ONE burst of loads
#pragma unroll 1
        for (int sb = 0; sb < ((probe & 4) || !restage ? 0 : DPW / 2); sb += SB) {
          uint2 raw[SB];
#pragma unroll
          for (int j = 0; j < SB; ++j) {
            const int sl = 2 * (sb + j) + half;
            const int nds = s_nd[wave][sl];
            const CT* cp = codes + s_cl[wave][sl];
            // clamped to the list's last CPS-aligned group: an aligned 8-byte load, at most CPS - 1 entries past the end
            const int pos = min(p0 + CPS * hl, max(nds - 1, 0) & ~(CPS - 1));
            raw[j] = make_uint2(0u, 0u);
            if (nds > (p0 > 0 ? p0 : fit)) raw[j] = *reinterpret_cast<const uint2*>(cp + pos);   // only the lists not staged yet
          }
#pragma unroll
          for (int j = 0; j < SB; ++j) {
            const int sl = 2 * (sb + j) + half;
            if (p0 > 0 || s_nd[wave][sl] > fit) *reinterpret_cast<uint2*>(&s_codes[wave][sl][HDR + CPS * hl]) = raw[j];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- (2) scan: lanes 0 and 1 of a document's group each take one half of its staged codes [0, cnt), test them
        // against the bitmap and compact the hot ones to the front of their own half (a write never passes the lane's read
        // position).  All documents of the claim advance together, four codes per lane and step: ONE LDS read for the codes,
        // four independent bitmap reads, then the writes -- two LDS round trips per four codes.  The loop bound comes from
        // the claim's longest list (no reduction); the two halves stay where they are, the walk takes them one after the other.
        const int cnt = min(max(nd - p0, 0), CAP);
        const int part = (((cnt + 1) >> 1) + 3) & ~3;          // halves start 8-B (u16) / 16-B (u32) aligned
        CT* row = &s_codes[wave][grp][HDR];   // the codes of the row (behind the block header)
        int hmine = 0;
        if (!(probe & 2)) {
          const int start = jl == 0 ? 0 : part, end = jl == 0 ? min(part, cnt) : cnt;
          const int itmax = (((min(nmax - p0, CAP) + 1) >> 1) + 3) & ~3;   // >= every document's half
          // Every staged position holds a valid code (< K: the staging reads only list entries or the zero padding of the
          // array), so the bitmap read needs no clamp; a cold code is written too, at a position the next hot code (or
          // nobody) overwrites -- wpos never passes the codes already in registers -- so the loop body has no branch.
          int wpos = start;
          int rem = jl < 2 ? end - start : 0;     // codes this lane still has to test
          for (int it = 0; it < itmax; it += 4) {
            const int pos = min(start + it, CAP - 4);
            uint32_t c[4];
            if constexpr (sizeof(CT) == 2) {
              const uint2 w2 = *reinterpret_cast<const uint2*>(row + pos);
              c[0] = w2.x & 0xFFFFu; c[1] = w2.x >> 16; c[2] = w2.y & 0xFFFFu; c[3] = w2.y >> 16;
            } else {
              const uint4 w4 = *reinterpret_cast<const uint4*>(row + pos);
              c[0] = w4.x; c[1] = w4.y; c[2] = w4.z; c[3] = w4.w;
            }
            uint32_t bw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) bw[k] = s_bits[c[k] >> 5];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t h = (k < rem) ? ((bw[k] >> (c[k] & 31)) & 1u) : 0u;
              if (rem > 0) row[wpos] = (CT)c[k];
              wpos += (int)h;
            }
            rem -= 4;
          }
          hmine = wpos - start;
        }
        const int h0 = group_lane<LPD, 0>(hmine, lane), h1 = group_lane<LPD, 1>(hmine, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- (3) walk the hot codes in lockstep (LPD lanes per row, 8 gathers in flight per lane; positions past a
        // document's hot codes carry an out-of-range offset: the bounds check answers without a memory request).
        // Segment A = lane 0's hot codes at row[0 ..), segment B = lane 1's at row[part ..).
        if (jl == 0) rows32 += (uint32_t)(h0 + h1);
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
          const int hn = seg == 0 ? h0 : h1;
          const int hmax = (probe & 1) ? 0 : wave_max_nonneg(hn);
          const CT* mine = row + (seg == 0 ? 0 : part);
          for (int t = 0; t < hmax; t += 8) {
            uint32_t c[8];
            if constexpr (sizeof(CT) == 2) {
              const uint4 cw = *reinterpret_cast<const uint4*>(mine + t);
              c[0] = cw.x & 0xFFFFu; c[1] = cw.x >> 16; c[2] = cw.y & 0xFFFFu; c[3] = cw.y >> 16;
              c[4] = cw.z & 0xFFFFu; c[5] = cw.z >> 16; c[6] = cw.w & 0xFFFFu; c[7] = cw.w >> 16;
            } else {
              const uint4 ca = *reinterpret_cast<const uint4*>(mine + t), cb = *reinterpret_cast<const uint4*>(mine + t + 4);
              c[0] = ca.x; c[1] = ca.y; c[2] = ca.z; c[3] = ca.w;
              c[4] = cb.x; c[5] = cb.y; c[6] = cb.z; c[7] = cb.w;
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t off = (t + k < hn) ? c[k] * (uint32_t)ROWB + (uint32_t)(jl * 16) : 0x7FFFFFF0u;
              const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)off, 0, 0);
              v[k] = make_uint4(r.x, r.y, r.z, r.w);
            }
            asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[4 * j + e] = max(st[4 * j + e], (w4[j] >> (8 * e)) & 0xFFu);
            }
          }
        }
      }
      uint32_t sum = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += (16 * jl + k < Lq) ? max(st[k], lam) : 0u;   // padding tokens carry no bound
      if constexpr (LPD == 2) {
        sum += (uint32_t)__builtin_amdgcn_mov_dpp((int)sum, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]: the group's other lane
      } else {
#pragma unroll
        for (int o = 1; o < LPD; o <<= 1) sum += (uint32_t)__shfl_xor((int)sum, o);
      }
      if (valid && jl == 0) {
        U[pbase + i] = (uint16_t)sum;
        atomicAdd(&s_hist[min(sum >> hshift, (uint32_t)(NP_UB_BINS - 1))], 1u);
      }
      __builtin_amdgcn_wave_barrier();   // s_cl / s_nd of this group are rewritten by the next claim
      i0 = i1;
      did = did_next;
      i1 = static_claims ? i1 + NW * DPW : (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)inext);
    }
    __syncthreads();
    for (int i = tid; i < NP_UB_BINS; i += 256) {
      const uint32_t v = s_hist[i];
      if (v) atomicAdd(&hb[i], v);
    }
  }
  unsigned long long toks = toks32, ucnt = ucnt32, rows = rows32;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
    rows += __shfl_xor(rows, o);
  }
  __shared__ unsigned long long s_cnt[3];
  __syncthreads();
  if (tid == 0) s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
  __syncthreads();
  if (lane == 0 && toks) {
    atomicAdd(&s_cnt[0], toks);
    atomicAdd(&s_cnt[1], ucnt);
    atomicAdd(&s_cnt[2], rows);
  }
  __syncthreads();
  if (tid == 0 && s_cnt[0]) {
    atomicAdd(&ctr->n_cand_tokens, s_cnt[0]);
    atomicAdd(&ctr->n_cand_dcodes, s_cnt[1]);
    atomicAdd(&ctr->n_cand_codes, s_cnt[2]);     // table rows gathered
  }
}

// ---------------------------------------------------------------------------------------------
// S4, first filter level in BIT-PLANE form (round 4).
// approx_hot_kernel is VALU-bound (50 wave instructions per document at 10 M documents: scan 16, walk 12-24, staging 7):
// the walk folds 32 byte maxima per table row and pads every document of a claim to the longest hot list.  Rounding the
// values above Lambda UP to one of P = 8 levels  Lambda = t_0 < t_1 < ... < t_8  turns a hot centroid's row into 8 bit
// planes -- plane j = the query tokens with u[q, c] > t_j -- the max over a document's hot codes into a bitwise OR and
// the bound into a weighted popcount:
//     U''(d) = Lq * Lambda + sum_j (t_{j+1} - t_j) * popcount( OR_{c in codes(d), c hot} plane_j[c] )  >=  U'(d)  >=  U(d).
// Any upper bound keeps the three-step cut of np_search.hip selection-preserving (S1 = the n_sel largest bounds, tau from
// their exact U, S2 = the rest with bound >= tau); what the rounding costs is a larger S2.  CPU simulation on the metric
// corpus (tools/sim/s4_planes_sim.py, profiles/r04_sim_s4_planes_10m.txt): uniform levels up to the table's maximum keep
// |S2| within ~10 % of the exact hot bound's (16 planes: within 1 %; levels crowded next to Lambda are much worse -- the
// documents near the cut are told apart by their STRONG matches).
// Kernel structure (one XCD per query, claims round-robin as in approx_hot_kernel), per claim of DPW = 64 / LPD documents:
//   stage  the claim's list blocks -> LDS rows, 16 aligned 8-byte loads per lane in one burst (LPD = 2: half a wave per
//          block of <= 256 B; LPD = 4: a whole wave per block of <= 512 B, for corpora with long distinct-code lists);
//   scan   the LPD lanes of a document each test one share of its codes (<= 64) against the hot bitmap, 4 codes per step,
//          and keep the hot POSITIONS as a 64-bit mask in registers: no compaction, no LDS write;
//   walk   every lane pops its own hot positions (G per step), fetches the code from the staged row and the 8 planes of
//          that centroid through a bounds-checked buffer (an exhausted lane issues no request) and ORs them in: 8 v_or per
//          row instead of 32 byte maxima, and all 64 lanes walk lists of their own (lockstep over <= 4 positions per lane);
//   bound  OR across the document's lanes (DPP), weighted popcount, u16 bound + LDS histogram.
// ---------------------------------------------------------------------------------------------
#define NP_PLANES 8

// Per query: the plane thresholds and the hot bitmap.  lev[b][0..7] = t_0..t_7 (t_0 = Lambda, hot_lam_kernel), lev[b][8..15] =
// the weights t_{j+1} - t_j, with t_8 = the largest per-centroid maximum of the query's table and the levels spaced by a
// power law, t_j = Lambda + span * (j / 8)^e: e = 1 is uniform, the default e = 1.5 is finer next to Lambda -- on the metric
// corpus 8 planes at e = 1.5 keep |S2| within ~10 % of the exact hot bound's, uniform ones within ~35 %, and levels crowded
// next to Lambda (geometric, quantiles of the values) are far worse (profiles/r04_sim_s4_planes*_10m.txt).
// hotbits[b][w] bit i = (M[32 w + i] > Lambda): the filter's workgroups copy it into LDS instead of each rebuilding it from
// the 64 KB of per-centroid maxima.  One block per query.
__global__ void __launch_bounds__(256) hot_levels_kernel(const uint32_t* __restrict__ chist, int64_t K, int hot_permille,
                                                         const uint8_t* __restrict__ cmaxu, int64_t KP, int pexp10,
                                                         uint32_t* __restrict__ lam_out, uint32_t* __restrict__ lev,
                                                         uint32_t* __restrict__ hotbits,
                                                         int warm_permille /* >= 1000: no second threshold */,
                                                         uint32_t* __restrict__ lam2_out /* [B] Lambda2 (floor of the exact level) */,
                                                         uint32_t* __restrict__ warmbits /* [B][KP / 32] bit c = M[c] > Lambda2 */,
                                                         const int32_t* __restrict__ n_cand = nullptr /* [B]: scale the hot share by the
                                                                                                         query's candidate count */,
                                                         int hot_ref = 0 /* candidates up to which hot_permille applies as given */) {
  // grid (blocks, B): every block derives Lambda from the 256-bin histogram itself (as hot_lam_kernel: the smallest level
  // with at most hot_permille of the centroids above it) and builds its slice of the bitmap; block 0 publishes Lambda and
  // the thresholds.  (One block per query took 37 us at K = 2^16, serial in the 64 KB of maxima.)
  __shared__ uint32_t s_wsum[4];
  __shared__ int s_ok, s_top;
  __shared__ int s_t[NP_PLANES + 1];
  const int b = blockIdx.y, v = threadIdx.x, lane = v & 63, wave = v >> 6;
  if (v == 0) {
    s_ok = 0;
    s_top = 0;
  }
  const uint32_t hv = chist[(int64_t)b * 256 + v];
  uint32_t suf = hv;   // -> #(M >= v): suffix sum over the block
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_down((int)suf, o);
    if (lane + o < 64) suf += t;
  }
  if (lane == 0) s_wsum[wave] = suf;
  __syncthreads();
  for (int k = wave + 1; k < 4; ++k) suf += s_wsum[k];
  // Round 5: the hot share follows the query's candidate count.  Every candidate pays the first level in proportion to the
  // share h (plane rows and walk steps), while what a larger h buys -- fewer documents at the exact level -- does not grow with
  // the candidates, so the best share falls as they grow.  Measured at 10 M documents: 186 k candidates per query (t_cs = 0.4)
  // are best served by 40-80 per mille; 2.06 M (the REST API's default, t_cs = None with nprobe 8) by ~30: 3.5 / 4.5 / 4.9 /
  // 4.4 / 3.2 k queries/s at 8 / 15 / 30 / 60 / 120.  h = h0 (n_ref / n)^(1/3) passes through both.
  if (n_cand != nullptr && hot_ref > 0) {
    const float nq = (float)max(n_cand[blockIdx.y], 1);
    if (nq > (float)hot_ref)
      hot_permille = max(min(hot_permille, 8), (int)((float)hot_permille * __powf((float)hot_ref / nq, 0.3333333f)));
  }
  const uint64_t limit = (uint64_t)K * (uint64_t)hot_permille / 1000u;
  const bool ok = v >= 1 && (uint64_t)suf <= limit;
  const int cnt = (int)__popcll(__ballot(ok));
  if (lane == 0 && cnt) atomicAdd(&s_ok, cnt);
  if (hv) atomicMax(&s_top, v);
  // Lambda2 <= Lambda by the same rule at the (larger) warm share: the exact level skips the rows of the centroids with
  // M <= Lambda2 and floors every token's maximum there (approx_ub_kernel, FLOOR)
  __shared__ int s_ok2;
  if (v == 0) s_ok2 = 0;
  __syncthreads();
  const bool warm_on = warm_permille < 1000 && warmbits != nullptr;
  if (warm_on) {
    const uint64_t limit2 = (uint64_t)K * (uint64_t)max(warm_permille, hot_permille) / 1000u;
    const int cnt2 = (int)__popcll(__ballot(v >= 1 && (uint64_t)suf <= limit2));
    if (lane == 0 && cnt2) atomicAdd(&s_ok2, cnt2);
  }
  __syncthreads();
  const int lam = 255 - s_ok;
  const int lam2 = 255 - s_ok2;
  if (v <= NP_PLANES) {
    const int top = max(s_top, lam + 1), span = top - lam;
    const float w = __powf((float)v / (float)NP_PLANES, 0.1f * (float)pexp10);
    int t = v == 0 ? lam : (v == NP_PLANES ? top : lam + (int)ceilf((float)span * w));
    t = min(max(t, min(lam + v, top)), top);          // strictly increasing while there is room, never past the top
    s_t[v] = t;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (v == 0) lam_out[b] = (uint32_t)lam;
    if (v == 0 && warm_on) lam2_out[b] = (uint32_t)lam2;
    if (v < NP_PLANES) {
      const int t0 = s_t[v], t1 = max(s_t[v + 1], t0);
      lev[b * 16 + v] = (uint32_t)t0;
      lev[b * 16 + 8 + v] = (uint32_t)(t1 - t0);
    }
  }
  const uint8_t* cm = cmaxu + (int64_t)b * KP;
  for (int64_t w = (int64_t)blockIdx.x * 256 + v; w < (KP >> 5); w += (int64_t)gridDim.x * 256) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(cm + w * 32), v1 = *reinterpret_cast<const uint4*>(cm + w * 32 + 16);
    const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint32_t bits = 0, bits2 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t m = (w8[e] >> (8 * k)) & 0xFFu;
        bits |= (m > (uint32_t)lam ? 1u : 0u) << (4 * e + k);
        bits2 |= (m > (uint32_t)lam2 ? 1u : 0u) << (4 * e + k);
      }
    hotbits[(int64_t)b * (KP >> 5) + w] = bits;
    if (warm_on) warmbits[(int64_t)b * (KP >> 5) + w] = bits2;
  }
}

// planes[b][c][j] for every hot centroid c of query b: bit q of plane j = (u[q, c] > t_j).  RB = bytes of a u8 table row =
// bits of a plane x 8 planes / 8: plane rows have the u8 rows' size and addressing.  A wave takes 64 centroids and builds
// the rows of the hot ones one after the other: lane q holds u[q, c], a plane is one ballot.  Rows of cold centroids are
// never read and stay unwritten.
template <int RB>
__global__ void __launch_bounds__(256) hot_planes_kernel(const uint8_t* __restrict__ QCU, int64_t KP,
                                                         const uint8_t* __restrict__ cmaxu, const uint32_t* __restrict__ lam_b,
                                                         const uint32_t* __restrict__ lev, uint32_t* __restrict__ planes) {
  static_assert(RB == 32 || RB == 64, "plane rows of 32 or 64 query tokens");
  // A block takes 2048 centroids at a time: their hot ones are compacted into LDS (one ballot per wave and 64 centroids),
  // then every LANE builds the row of one hot centroid -- its RB bytes against the 8 thresholds, two instructions per
  // (byte, threshold) -- so the work is spread over all lanes whatever the hot share (a wave building one row at a time
  // by ballots took 0.33 ms per batch at K = 2^19).
  constexpr int CH = 2048;
  __shared__ uint32_t s_list[CH];
  __shared__ int s_n;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const uint32_t lam = lam_b[b];
  uint32_t t[NP_PLANES];
#pragma unroll
  for (int j = 0; j < NP_PLANES; ++j) t[j] = lev[b * 16 + j];
  const uint8_t* cm = cmaxu + (int64_t)b * KP;
  for (int64_t c0 = (int64_t)blockIdx.x * CH; c0 < KP; c0 += (int64_t)gridDim.x * CH) {
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < CH; i += 256) {   // CH is a multiple of 256: whole waves
      const int64_t c = c0 + i;
      const bool hot = c < KP && (uint32_t)cm[c] > lam;
      const unsigned long long bal = __ballot(hot);
      int base = 0;
      if (lane == 0 && bal) base = atomicAdd(&s_n, (int)__popcll(bal));
      base = __shfl(base, 0);
      if (hot) s_list[base + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
    __syncthreads();
    const int n = s_n;
    for (int i = tid; i < n; i += 256) {
      const int64_t row = (int64_t)b * KP + c0 + s_list[i];
      const uint4* src = reinterpret_cast<const uint4*>(QCU + row * RB);
      uint32_t pl[NP_PLANES][RB / 32];
#pragma unroll
      for (int j = 0; j < NP_PLANES; ++j)
#pragma unroll
        for (int h = 0; h < RB / 32; ++h) pl[j][h] = 0;
#pragma unroll
      for (int q4 = 0; q4 < RB / 16; ++q4) {
        const uint4 v = src[q4];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t u = (w4[e] >> (8 * k)) & 0xFFu;
            const int q = 16 * q4 + 4 * e + k;
#pragma unroll
            for (int j = 0; j < NP_PLANES; ++j) pl[j][q >> 5] |= (u > t[j] ? 1u : 0u) << (q & 31);
          }
      }
      uint32_t* dst = planes + row * (RB / 4);
      if constexpr (RB == 32) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(pl[0][0], pl[1][0], pl[2][0], pl[3][0]);
        *reinterpret_cast<uint4*>(dst + 4) = make_uint4(pl[4][0], pl[5][0], pl[6][0], pl[7][0]);
      } else {
#pragma unroll
        for (int j = 0; j < NP_PLANES; j += 2)
          *reinterpret_cast<uint4*>(dst + 2 * j) = make_uint4(pl[j][0], pl[j][1], pl[j + 1][0], pl[j + 1][1]);
      }
    }
  }
}

template <int RB, typename CT, int LPD, int PF /* walk steps in flight ahead of the prefetch: 1 or 2 */,
          int DPI /* documents per staging instruction: 4 (blocks <= 240 B), 2 (<= 496 B), 1 */,
          int QM /* hot codes of a lane: 0 = compacted in place over its share (LDS writes), 1 = a 64-bit position mask in registers */,
          int WPB = 4 /* waves per workgroup.  The hot bitmap is one copy per WORKGROUP: at K = 2^19 its 64 KB left room for ONE
                         4-wave workgroup per CU (round 6: 3.5 ms of hot level per batch in the crate-natural regime); 12 waves
                         sharing one copy fill the CU like three small workgroups do at K = 2^16 */>
// (Round 5, measured and removed -- commit 122dafe has the code: the NEXT claim's blocks requested into REGISTERS the moment this
// claim's rows are in LDS, so that they travel during its whole scan, and written to the rows after the walk.  The ISA did what
// was asked -- 8 buffer loads at the top of the claim, no vector-memory wait until the walk's first fold -- and the kernel did
// not move: 2.19 vs 2.18 ms of S4 at 10 M documents, and 5.07 vs 5.34 k queries/s in the dense regime (t_cs = None), where the
// 141 VGPRs cost more than the overlap gained.  A claim's time is one HBM round trip under load plus its compute whichever way the
// two are arranged; what would help is more claims in flight per CU, which LDS rows or registers both cap.)
__global__ void __launch_bounds__(64 * WPB) approx_hotp_kernel(
    const uint32_t* __restrict__ planes /* [B][KP][RB / 4] */, int64_t K, int64_t KP, const uint32_t* __restrict__ hotbits /* [B][KP / 32] */,
    const uint32_t* __restrict__ lam_b, const uint32_t* __restrict__ lev /* [B][16] */,
    const uint32_t* __restrict__ cand_ids /* [pool] shard-local document ids (compact_kernel) */,
    uint4* __restrict__ cand_meta /* [pool] OUT: the candidates' 16-B records, for the cuts and the exact level */,
    int ublock_stride /* entries per document block of `codes` */, int64_t ovf_base /* first entry of the overflow region */,
    const int32_t* __restrict__ n_cand, RoundPlan rp, int round, int max_rounds, const CT* __restrict__ codes,
    const uint32_t* __restrict__ qflag, const int32_t* __restrict__ qoff, int n_sel, uint16_t* __restrict__ U,
    uint32_t* __restrict__ hist, int hshift, int32_t* __restrict__ slots, int32_t* __restrict__ ticket, int B, Counters* ctr,
    int slack /* spare LDS bytes behind a wave's rows: what the idle lanes of the last staging instruction overrun */,
    int probe_arg /* NP_DIAGNOSTICS builds only (results invalid when != 0): 1 no row loads in the walk, 2 no scan, 4 no staging */,
    int unordered = 0 /* the candidate ids come in blocks of ascending ids in any order (gain_sweep_kernel): a claim's block
                         offsets are taken from its smallest id instead of its first */) {
  static_assert(RB == 32 || RB == 64, "plane rows of 32 or 64 query tokens");
  static_assert(LPD == 2 || LPD == 4, "lanes per document");
#ifdef NP_DIAGNOSTICS
  const int probe = probe_arg;
#else
  constexpr int probe = 0;
#endif
  constexpr int DPW = 64 / LPD;                      // documents per claim
  constexpr int HDR = 16 / (int)sizeof(CT);          // entries of a block's 16-byte header {#distinct, doc length, overflow index, 0}
  constexpr int NS = RB / 4;                         // dwords of plane state (and of a plane row)
  constexpr int PW = RB / 32;                        // dwords per plane
  constexpr int G = 128 / RB;                        // hot positions popped per walk step (32 dwords of rows in flight)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // LDS: the hot bitmap -- STATIC (8 KB) with u16 codes (K <= 65536), so that its address folds into the ds_read offset
  // field of the scan's four lookups per step, else the first KP / 32 words of the dynamic region -- and [4][DPW] rows
  constexpr bool SBM = sizeof(CT) == 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  __shared__ __attribute__((aligned(16))) uint32_t s_bits_static[SBM ? 2048 : 4];
  // histogram of the bounds at HALF the global resolution (4 KB instead of 8: a fourth workgroup per CU).  A document of bin
  // t is counted at bin t & ~1; ub_thr_kernel's threshold can only come out lower (more documents in S1), which any cut on
  // an upper bound tolerates, and ub_cut_kernel tests the true bins.
  __shared__ uint32_t s_hist[NP_UB_BINS / 2];
  __shared__ int s_q;
#define BITS(i) (SBM ? s_bits_static[(i)] : s_dyn[(i)])
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & (LPD - 1), grp = lane / LPD;
  const int x = blockIdx.x & 7;
  if (round >= rp.round_tab[2 * max_rounds]) return;
  const int rb = rp.round_tab[2 * round], re = rp.round_tab[2 * round + 1];
  const int bmw = SBM ? 0 : (int)(((KP >> 5) + 3) & ~3ll);   // dynamic bitmap words; the rows behind stay 16-byte aligned
  const int stride_b = ublock_stride * (int)sizeof(CT);              // block bytes (a multiple of 64)
  const int row_b = stride_b + 16;                   // LDS row: the block + 16 bytes (rows off each other's banks)
  char* s_rows = reinterpret_cast<char*>(s_dyn + bmw) + (size_t)wave * (DPW * row_b + slack);
  const int fit = ublock_stride - HDR;               // codes a block holds = codes staged per window
  uint32_t toks32 = 0, ucnt32 = 0, rows32 = 0;
  // the scan looks every staged position up in the bitmap, also past a list's end: the rows must never hold anything but
  // codes (< K) behind the header, so they start zeroed (afterwards they only ever receive list entries or zeros)
  for (int i = tid; i < WPB * (DPW * row_b + slack) / 4; i += 64 * WPB) (s_dyn + bmw)[i] = 0u;
  for (int step = 0;; ++step) {
    __syncthreads();
    if (tid == 0) s_q = xcd_next_query(slots, ticket, x, step, B, rp.order, rb, re, [&]() { return -3; });
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(s_q);
    if (b < 0) break;
    const int64_t n = n_cand[b];
    const int64_t pbase = rp.cand_base[b];
    const uint32_t* idb = cand_ids + pbase;
    uint4* metab = cand_meta + pbase;
    if (qflag[b] || n <= (int64_t)n_sel) {
      // the cuts keep every candidate of this query: no bound to compute, but the later stages still want the records
      for (int64_t i = ((int64_t)(blockIdx.x >> 3) * (64 * WPB) + tid); i < n; i += (int64_t)(gridDim.x >> 3) * (64 * WPB)) {
        const uint32_t d = idb[i];
        const uint4 hd = *reinterpret_cast<const uint4*>(codes + (int64_t)d * ublock_stride);
        const int64_t cl = (int)hd.x > fit ? ovf_base + (int64_t)hd.z * 4 : (int64_t)d * ublock_stride + HDR;
        metab[i] = make_uint4(d, hd.x, (uint32_t)(cl & 0xFFFFFFFFll), (uint32_t)((cl >> 32) & 0xFF) | (hd.y << 8));
      }
      continue;
    }
    const int Lq = qoff[b + 1] - qoff[b];
    const uint32_t lam = lam_b[b];
    uint32_t wj[NP_PLANES];                          // plane weights (wave-uniform: scalar registers)
#pragma unroll
    for (int j = 0; j < NP_PLANES; ++j) wj[j] = lev[b * 16 + 8 + j];
    // ---- hot bitmap (hot_levels_kernel): bit c = M[c] > Lambda
    {
      const uint32_t* hbits = hotbits + (int64_t)b * (KP >> 5);
      for (int w = tid; w < (int)(KP >> 5); w += 64 * WPB) BITS(w) = hbits[w];
      for (int i = tid; i < NP_UB_BINS / 2; i += 64 * WPB) s_hist[i] = 0;
    }
    __syncthreads();
    const uint64_t tb64 = reinterpret_cast<uint64_t>(planes + (int64_t)b * KP * NS);
    const uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb64);
    const uint32_t thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tb64 >> 32));
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)thi << 32) | tlo), 0, (int)(KP * RB), 0x00020000);
    uint32_t* hb = hist + (int64_t)b * NP_UB_BINS;
    // claims round-robin: wave g of the NW waves of this XCD takes claims g, g + NW, ... (approx_hot_kernel: a cursor
    // atomic per claim costs more than the imbalance it removes)
    const int64_t NW = (int64_t)(gridDim.x >> 3) * WPB;
    int64_t i0 = ((int64_t)(blockIdx.x >> 3) * WPB + wave) * DPW, i1 = i0 + NW * DPW;
    const uint32_t id_last = idb[n - 1];
    char* row = s_rows + (size_t)grp * row_b;        // this lane's document
    const CT* rowc = reinterpret_cast<const CT*>(row) + HDR;   // its codes (behind the block header)
    // ---- stage: a claim's list BLOCKS -> LDS rows, header included, by LDS-direct loads (buffer_load ... lds: the data goes
    // from memory straight into the rows at M0 + lane x size, no staging VGPRs).  The stage is bound by the CU's vector-memory
    // pipe, which charges a 64-lane instruction by the lanes and cache lines it touches, not by its bytes (a dword per lane
    // moved 256 bytes per instruction: 675 pipe cycles per claim of 32 documents, more than scan and walk together).  So
    // the blocks travel PACKED, 16 bytes per lane: a row = the block + 16 bytes of padding = LPR lanes, DPI documents per
    // instruction (a power of two, DPI x LPR <= 64), per-lane offsets from the claim's FIRST block (candidate ids ascend,
    // so the offsets are non-negative; a claim whose ids span 2 GiB of blocks takes the one-block-per-instruction path).
    // NO lane is masked off: idle lanes are out of the buffer's range, get zero, and their 16 bytes land on the head of
    // the rows the NEXT instruction fills, issued later (loads return in order; slack spare bytes follow a wave's
    // last row).  Masking would put the loads behind an EXEC branch, and the compiler would then have to assume at every
    // later wait that they may not have been issued: the wait for a walk step's rows would wait for these blocks too.
    const int LPR = stride_b / 16 + 1;
    const int sslot = lane / LPR, spiece = lane - sslot * LPR;
    auto stage = [&](uint32_t dv) {
      if (probe & 4) return;
      uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)dv, 0), d1 = (uint32_t)__builtin_amdgcn_readlane((int)dv, 63);
      if (unordered) {   // shard-local ids are below 2^31
        d1 = (uint32_t)wave_max_nonneg((int)dv);
        d0 = 0x7FFFFFFFu - (uint32_t)wave_max_nonneg((int)(0x7FFFFFFFu - dv));
      }
      if ((uint64_t)(d1 - d0) * (uint64_t)stride_b < 0x7FFF0000ull && d1 >= d0) {
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<CT*>(codes) + (int64_t)d0 * ublock_stride, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int k = 0; k < DPW / DPI; ++k) {
          const uint32_t dj = (uint32_t)__shfl((int)dv, LPD * (k * DPI + sslot));   // the id of this lane's document (any lane of its group)
          const uint32_t voff = sslot < DPI ? (dj - d0) * (uint32_t)stride_b + 16u * (uint32_t)spiece : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(s_rows + (size_t)k * DPI * row_b), 16, (int)voff, 0, 0, 0);
        }
      } else {
#pragma unroll 1
        for (int j = 0; j < DPW; ++j) {
          const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)dv, LPD * j);
          const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<CT*>(codes) + (int64_t)dj * ublock_stride, 0, stride_b, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(s_rows + (size_t)j * row_b), 4, 4 * lane, 0, 0, 0);
          if (stride_b > 256)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(s_rows + (size_t)j * row_b + 256), 4, 256 + 4 * lane, 0, 0, 0);
        }
      }
    };
    uint32_t did = i0 + grp < n ? idb[i0 + grp] : id_last;
    if (i0 < n) stage(did);
    for (;;) {
      if (i0 >= n) break;
      const int64_t i = i0 + grp;
      const bool valid = i < n;
      const bool more = i1 < n;                      // wave-uniform: this wave has another claim
      const uint32_t did_next = more && i1 + grp < n ? idb[i1 + grp] : id_last;
      __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): the claim's rows are in LDS
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint4 hd = *reinterpret_cast<const uint4*>(row);   // {#distinct, doc length, overflow index, 0}
      const int nd = valid ? (int)hd.x : 0;
      const bool ovf = nd > fit;
      const int64_t cl = ovf ? ovf_base + (int64_t)hd.z * 4 : (int64_t)did * ublock_stride + HDR;
      if (jl == 0 && valid) {
        toks32 += hd.y;
        ucnt32 += (uint32_t)nd;
        metab[i] = make_uint4(did, (uint32_t)nd, (uint32_t)(cl & 0xFFFFFFFFll), (uint32_t)((cl >> 32) & 0xFF) | (hd.y << 8));
      }
      const int nmax = wave_max_nonneg(nd);
      uint32_t st[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) st[k] = 0;
      int hq = 0;                                    // entries in this lane's hot queue
      // ---- walk: G queue entries per lane and step; a lane without one issues out-of-range offsets (no request).  With
      // `prefetch` the next claim's blocks are requested right behind the first step's rows: the rest of the walk and the
      // bound overlap their way from memory (vmcnt waits are in order, so the first step's rows arrive first).
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const char* hotq = nullptr;                    // the lane's hot queue (set by the scan of each window)
      unsigned long long hm = 0ull;                  // QM = 1: the lane's hot positions inside its share
      auto issue = [&](int t, u32x4 (&v)[G][RB / 16]) {
        uint32_t c[G];
        if constexpr (QM == 1) {
          // pop G positions from the bottom of the mask: straight-line, the LDS reads unconditional (position 0 for an
          // exhausted lane) and fenced so that the compiler does not sink each into a branch of its own
          const CT* mine_c = reinterpret_cast<const CT*>(hotq);
          bool has[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            has[g] = hm != 0ull;
            const int p = max(__ffsll((long long)hm) - 1, 0);
            hm &= hm - 1ull;
            c[g] = (uint32_t)mine_c[p];
          }
          if constexpr (G == 4) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
          else asm volatile("" : "+v"(c[0]), "+v"(c[1]));
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const uint32_t off = (has[g] && !(probe & 1)) ? c[g] * (uint32_t)RB : 0x7FFFFF00u;
#pragma unroll
            for (int k = 0; k < RB / 16; ++k) v[g][k] = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(off + 16u * k), 0, 0);
          }
          return;
        }
        if constexpr (sizeof(CT) == 2 && G == 4) {
          const uint2 q2 = *reinterpret_cast<const uint2*>(hotq + 2 * t);
          c[0] = q2.x & 0xFFFFu; c[1] = q2.x >> 16; c[2] = q2.y & 0xFFFFu; c[3] = q2.y >> 16;
        } else if constexpr (sizeof(CT) == 2) {
          const uint32_t q1 = *reinterpret_cast<const uint32_t*>(hotq + 2 * t);
          c[0] = q1 & 0xFFFFu; c[1] = q1 >> 16;
        } else {
#pragma unroll
          for (int g = 0; g < G; g += 2) {
            const uint2 q2 = *reinterpret_cast<const uint2*>(hotq + 4 * (t + g));
            c[g] = q2.x; c[g + 1] = q2.y;
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const uint32_t off = (t + g < hq && !(probe & 1)) ? c[g] * (uint32_t)RB : 0x7FFFFF00u;
#pragma unroll
          for (int k = 0; k < RB / 16; ++k) v[g][k] = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(off + 16u * k), 0, 0);
        }
      };
      auto fold = [&](const u32x4 (&v)[G][RB / 16]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int k = 0; k < RB / 16; ++k) {
            st[4 * k + 0] |= v[g][k].x;
            st[4 * k + 1] |= v[g][k].y;
            st[4 * k + 2] |= v[g][k].z;
            st[4 * k + 3] |= v[g][k].w;
          }
      };
      // The queue lives IN the row: a lane writes its hot codes back over the start of its own share (the write position
      // never passes the codes already read).  `prefetch`: the queue's first PF steps are read and their rows requested,
      // then the next claim's blocks right behind them (they overwrite the rows: every later queue entry must have been
      // read before, so steps beyond PF run first), and only then are the rows folded -- straight-line code, so that the
      // wait before the fold is vmcnt(#block loads): the rows arrive first (vmcnt is in order) and the blocks travel during
      // the fold, the bound and the claim bookkeeping.
      auto walk = [&](bool prefetch) {
        if constexpr (QM == 1) {
          // positions beyond the PF * G the prefetch steps take are popped from the TOP of the mask first
          const CT* mine_c = reinterpret_cast<const CT*>(hotq);
          int cnt = (int)__popcll(hm);
          rows32 += (uint32_t)cnt;
          const int keep = prefetch ? PF * G : 0;
          while (__ballot(cnt > keep) != 0ull) {
            u32x4 v[G][RB / 16];
            uint32_t c[G];
            bool has[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
              has[g] = cnt > keep;
              const int p = has[g] ? 63 - (int)__builtin_clzll(hm | 1ull) : 0;
              hm &= ~((has[g] ? 1ull : 0ull) << p);
              cnt -= has[g] ? 1 : 0;
              c[g] = (uint32_t)mine_c[p];
            }
            if constexpr (G == 4) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
            else asm volatile("" : "+v"(c[0]), "+v"(c[1]));
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const uint32_t off = (has[g] && !(probe & 1)) ? c[g] * (uint32_t)RB : 0x7FFFFF00u;
#pragma unroll
              for (int k = 0; k < RB / 16; ++k) v[g][k] = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(off + 16u * k), 0, 0);
            }
            fold(v);
          }
          if (prefetch) {
            u32x4 v0[G][RB / 16], v1[PF == 2 ? G : 1][RB / 16];
            issue(0, v0);
            if constexpr (PF == 2) issue(G, v1);
            stage(did_next);
            fold(v0);
            if constexpr (PF == 2) fold(v1);
          }
          hm = 0ull;
          return;
        }
        const int hmax = wave_max_nonneg(hq);
        for (int t = prefetch ? PF * G : 0; t < hmax; t += G) {
          u32x4 v[G][RB / 16];
          issue(t, v);
          fold(v);
        }
        if (prefetch) {
          u32x4 v0[G][RB / 16], v1[PF == 2 ? G : 1][RB / 16];
          issue(0, v0);
          if constexpr (PF == 2) issue(G, v1);
          stage(did_next);
          fold(v0);
          if constexpr (PF == 2) fold(v1);
        }
        rows32 += (uint32_t)hq;
        hq = 0;
      };
      for (int p0 = 0; p0 < nmax; p0 += fit) {
        // lists that fit their block are already staged; only a claim with an overflow list or a further window of a long
        // list goes back to memory: the same LDS-direct loads, document by document, from the list's own address (reads
        // past the list's end return zeros: the buffer ends with the list)
        if (p0 > 0 || nmax > fit) {                  // wave-uniform
          __builtin_amdgcn_wave_barrier();           // the previous window is consumed
#pragma unroll 1
          for (int j = 0; j < DPW; ++j) {
            const int ndj = __builtin_amdgcn_readlane(nd, LPD * j);
            if (ndj <= (p0 > 0 ? p0 : fit)) continue;   // staged with its block / already finished
            const uint32_t clo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cl, LPD * j);
            const uint32_t chi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cl >> 32), LPD * j);
            const int64_t clj = (int64_t)(((uint64_t)chi << 32) | clo) + p0;
            const int left_b = (min(ndj - p0, fit) * (int)sizeof(CT) + 3) & ~3;   // whole dwords: an odd u16 count reads one entry of the list's padding
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<CT*>(codes) + clj, 0, left_b, 0x00020000);
            for (int h = 0; 256 * h < fit * (int)sizeof(CT); ++h)
              if (256 * h + 4 * lane < fit * (int)sizeof(CT))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(s_rows + (size_t)j * row_b + 16 + 256 * h), 4, 256 * h + 4 * lane, 0, 0, 0);
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        // ---- scan: lane jl of the document's LPD lanes tests codes [start, start + len) of the window, CPI per step (one
        // 16-byte LDS read: 8 u16 / 4 u32 codes).  Every code is WRITTEN to the lane's hot queue at the queue's current
        // length and the length grows by the code's hot flag (one v_bfe_u32 on the bitmap word: the bit offset is the
        // code's low 5 bits): a cold code is overwritten by the next one -- no branch, no compaction pass, no mask.  A
        // lane's last step may run up to CPI - 1 positions past its list: those hold ZEROS (block padding / out-of-range
        // reads), i.e. code 0, which at worst ORs centroid 0's planes into the bound -- still an upper bound.  The queue is
        // all the walk reads, so after the last window's scan the rows are free for the next claim's blocks.
        constexpr int CPI = 16 / (int)sizeof(CT);
        const int cnt = min(max(nd - p0, 0), fit);
        const int share = (((cnt + LPD - 1) / LPD) + CPI - 1) & ~(CPI - 1);
        const int start = jl * share;
        const int len8 = (max(min(share, cnt - start), 0) + CPI - 1) & ~(CPI - 1);      // this lane's positions, whole steps
        const int itmax = (((min(nmax - p0, fit) + LPD - 1) / LPD) + CPI - 1) & ~(CPI - 1);   // wave-uniform, >= every len8, <= 64
        CT* mine = const_cast<CT*>(rowc) + start;    // the lane's share: read CPI codes ahead, written back as the hot queue
        hotq = reinterpret_cast<const char*>(mine);
        uint4 w = *reinterpret_cast<const uint4*>(mine);
        if constexpr (QM == 1) {
          // position mask: the hot flags enter from the top (v_alignbit: acc = acc >> 1 | flag << 31, one instruction per
          // code, only bit 0 of `flag` counts, so the bitmap word is just shifted right by the code); no LDS write at all
          const int len = max(min(share, cnt - start), 0);
          uint32_t mlo = 0, mhi = 0;
          auto look = [&](const uint4& x, uint32_t& acc) {
            const uint32_t ww[4] = {x.x, x.y, x.z, x.w};
            if constexpr (sizeof(CT) == 2) {
              uint32_t bl[4], bh[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                bl[k] = BITS((ww[k] >> 5) & 0x7FFu);
                bh[k] = BITS(ww[k] >> 21);
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                acc = __builtin_amdgcn_alignbit(bl[k] >> (ww[k] & 31u), acc, 1);
                acc = __builtin_amdgcn_alignbit(bh[k] >> ((ww[k] >> 16) & 31u), acc, 1);
              }
            } else {
              uint32_t bw[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) bw[k] = BITS(ww[k] >> 5);
#pragma unroll
              for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_alignbit(bw[k] >> (ww[k] & 31u), acc, 1);
            }
          };
          const int n_lo = min(itmax, 32), n_hi = itmax - n_lo;
          const int itm = (probe & 2) ? 0 : itmax;
          for (int it = 0; it < min(itm, 32); it += CPI) {
            const uint4 wn = *reinterpret_cast<const uint4*>(mine + min(it + CPI, fit - CPI - start));
            look(w, mlo);
            w = wn;
          }
          for (int it = 32; it < itm; it += CPI) {
            const uint4 wn = *reinterpret_cast<const uint4*>(mine + min(it + CPI, fit - CPI - start));
            look(w, mhi);
            w = wn;
          }
          mlo = (n_lo && itm) ? mlo >> (32 - n_lo) : 0u;
          mhi = (n_hi && itm) ? mhi >> (32 - n_hi) : 0u;
          hm = (((unsigned long long)mhi << 32) | mlo) & (len >= 64 ? ~0ull : ((1ull << len) - 1ull));
        } else
        for (int it = 0; it < ((probe & 2) ? 0 : itmax); it += CPI) {
          const uint4 wn = *reinterpret_cast<const uint4*>(mine + min(it + CPI, fit - CPI - start));   // next step's codes, on their way
          if (it < len8) {
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            if constexpr (sizeof(CT) == 2) {
              uint32_t bl[4], bh[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                bl[k] = BITS((ww[k] >> 5) & 0x7FFu);
                bh[k] = BITS(ww[k] >> 21);
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                mine[hq] = (CT)ww[k];
                hq += (int)__builtin_amdgcn_ubfe(bl[k], ww[k], 1u);
                mine[hq] = (CT)(ww[k] >> 16);
                hq += (int)__builtin_amdgcn_ubfe(bh[k], ww[k] >> 16, 1u);
              }
            } else {
              uint32_t bw[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) bw[k] = BITS(ww[k] >> 5);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                mine[hq] = (CT)ww[k];
                hq += (int)__builtin_amdgcn_ubfe(bw[k], ww[k], 1u);
              }
            }
          }
          w = wn;
        }
        const bool last_window = p0 + fit >= nmax;   // wave-uniform
        if (!last_window) walk(false);
      }
      walk(more);                                    // the last window's queue; the next claim's blocks ride behind its first step
      // ---- bound: OR across the document's lanes, weighted popcount
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        st[k] |= (uint32_t)__builtin_amdgcn_mov_dpp((int)st[k], 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
        if constexpr (LPD == 4) st[k] |= (uint32_t)__builtin_amdgcn_mov_dpp((int)st[k], 0x4E, 0xf, 0xf, true);   // [2,3,0,1]
      }
      uint32_t sum = (uint32_t)Lq * lam;
#pragma unroll
      for (int j = 0; j < NP_PLANES; ++j) {
        uint32_t pc = (uint32_t)__popc(st[j * PW]);
        if constexpr (PW == 2) pc += (uint32_t)__popc(st[j * PW + 1]);
        sum += __umul24(wj[j], pc);
      }
      if (valid && jl == 0) {
        U[pbase + i] = (uint16_t)sum;
        atomicAdd(&s_hist[min(sum >> (hshift + 1), (uint32_t)(NP_UB_BINS / 2 - 1))], 1u);
      }
      i0 = i1;
      did = did_next;
      i1 += NW * DPW;
    }
    __syncthreads();
    for (int i = tid; i < NP_UB_BINS / 2; i += 64 * WPB) {
      const uint32_t v = s_hist[i];
      if (v) atomicAdd(&hb[2 * i], v);
    }
  }
  unsigned long long toks = toks32, ucnt = ucnt32, rows = rows32;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
    rows += __shfl_xor(rows, o);
  }
  __shared__ unsigned long long s_cnt[3];
  __syncthreads();
  if (tid == 0) s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
  __syncthreads();
  if (lane == 0 && toks) {
    atomicAdd(&s_cnt[0], toks);
    atomicAdd(&s_cnt[1], ucnt);
    atomicAdd(&s_cnt[2], rows);
  }
  __syncthreads();
  if (tid == 0 && s_cnt[0]) {
    atomicAdd(&ctr->n_cand_tokens, s_cnt[0]);
    atomicAdd(&ctr->n_cand_dcodes, s_cnt[1]);
    atomicAdd(&ctr->n_cand_codes, s_cnt[2]);     // table rows gathered
  }
}

#undef BITS

// ---------------------------------------------------------------------------------------------
// Batched path (K > centroid_batch_size): approximate scores in the reference's arithmetic.
// search.rs:259-272 does not reuse the probe's GEMM: it recomputes Q.c for every distinct centroid of the candidates
// as an ndarray mat-vec, i.e. per (q, c) numeric_util::unrolled_dot -- eight partial sums p_j += x[8i+j] * y[8i+j]
// (multiply, then add: no FMA), then ((p0+p4) + (p1+p5)) + (p2+p6) + (p3+p7) folded into 0 one pair at a time.  That
// rounds differently from S1's k-ordered FMA chain (~1e-7), so the selection by approximate score can differ from the
// dense path's at near-ties.  The values are order-exact here too:
//   * ub_cut's survivors get the GEMM-valued score G as on the dense path (the u8 bound brackets both G and R);
//   * gcut_kernel keeps the documents with G >= (n_sel-th largest G) - 2e, where e >= |G - R| per document
//     (rounding analysis of both dot-product orders: (152 Lq + 2 Lq^2) 2^-24 ||q|| max||c||, x1.5): every document of
//     the R-ordered top n_sel is among them, typically n_sel plus a handful;
//   * approx_matvec_kernel recomputes R = sum_q max_c unrolled_dot(Q[q], C[c]) for those only, S5 selects on R.
// debug_trace skips the bounds and recomputes R for every candidate.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Approximate scores of the listed documents in unrolled_dot's arithmetic, as a register-blocked product on the vector ALUs.
//
// Work unit: a wave takes NP_MV_DOCS consecutive listed documents and walks the concatenation of their distinct-code lists
// 16 x NP_MV_PQ (document, code) pairs at a time.  A QUAD of lanes owns NP_MV_PQ pairs and splits unrolled_dot's eight partial
// sums between its lanes: lane kp = lane & 3 keeps (p[2kp], p[2kp+1]) of each of the quad's pairs, i.e. it multiplies
// x[8i + 2kp + {0,1}] by y[8i + 2kp + {0,1}] for i = 0, 1, ... in order -- one v_pk_mul_f32 and one v_pk_add_f32 per (pair, i),
// multiply THEN add, no FMA, exactly numeric_util::unrolled_dot's chains.  The lane's quarter of the quad's centroid rows sits in
// registers for the whole chunk (NP_MV_PQ x DIM / 4 floats); the query token's quarter row is DIM/4 floats read from LDS (DIM/16
// ds_read_b128 per token, a token ahead of its use).  The fold ((p0+p4) + (p1+p5)) + (p2+p6) + (p3+p7), added to 0 one pair at a
// time, crosses the quad with four DPP operands per pair.  Nothing else touches LDS on the arithmetic path.
//
// Why this shape -- kernel time per batch of 64 queries, K = 2^19, t_cs None, nprobe 8, ~1 030 listed documents x ~72 distinct
// codes per query (profiles/r06_ab_runs.md):
//   round 5: one document per wave, lanes = tokens x two codes, the centroid row read from LDS by every lane     1.42 ms
//   one lane per pair, the query row in SGPRs (scalar loads): 64 queries x 16 KB thrash the scalar cache         2.70 ms
//   one lane per pair, the query row broadcast from LDS to every lane (512 B per lane and token)                 1.99 ms
//   one lane per pair, a quad-replicated query row, v_mul_f32_dpp quad broadcasts                                1.54 ms
//   the partial sums split over the quad (this kernel), 4 pairs per quad, 8 documents per wave (2 waves/SIMD)    1.43 ms
//   ... the same with unpacked v_mul_f32 / v_add_f32: packed f32 is not the slower form here                     1.76 ms
//   2 pairs per quad (161 VGPRs, 3 waves/SIMD) 1.31 ms; 1 pair per quad 1.49 ms; 2 pairs, 4 documents per wave   1.20 ms
// The arithmetic alone (DIM packed multiplies + DIM packed adds per 16 x NP_MV_PQ pairs and token) is ~0.6 ms of that, the row
// gather (2.4 GB of 512-byte rows out of a 268 MB table) ~0.4 ms when nothing else runs; a wave does them one after the other, so
// occupancy -- registers -- decides how much of the gather hides.
// The per-token maxima over a document's codes are taken over a 32-token tile of sums in LDS (`if s > max`, search.rs:286-291 --
// the maximum's VALUE does not depend on the order the codes are visited in), and lane d adds document d's maxima in token order
// (search.rs:294-297), continuing from the previous 32-token tile's sum for longer queries.
// TAIL: the index files' dim (ldim <= DIM, rows zero-padded) is not a multiple of 8 -- unrolled_dot adds the last ldim % 8
// products one by one AFTER the eight partial sums.  That (rare) geometry takes the plain form: one lane per pair, the query
// row through scalar loads.
#ifndef NP_MV_DOCS
#define NP_MV_DOCS 4
#endif
#ifndef NP_MV_PQ
#define NP_MV_PQ 2
#endif
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int DIM, bool TAIL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3)))
approx_matvec_kernel(const float* __restrict__ qrows, const int32_t* __restrict__ qoff, const float* __restrict__ centroids,
                     const uint4* __restrict__ meta, const int32_t* __restrict__ n_list, RoundPlan rp, int round, CodeArr codes,
                     float* __restrict__ approx, int ldim) {
#pragma clang fp contract(off)
  constexpr int D = NP_MV_DOCS;
  constexpr int PQ = TAIL ? 4 : NP_MV_PQ;   // pairs of a quad
  constexpr int CH = 16 * PQ;               // pairs of a chunk
  constexpr int SS = CH + 4;                // row stride of the tile of sums (floats; 16-byte rows)
  constexpr int XK = DIM / 4 + 4;           // a k-part's quarter row (+16 B: the four parts start on distinct banks)
  constexpr int XT = 4 * XK;                // a token
  __shared__ __attribute__((aligned(16))) float sS[4][32 * SS];
  __shared__ float sM[4][D][33];
  __shared__ __attribute__((aligned(16))) float sX[TAIL ? 4 : 32 * XT];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (rp.round_of[b] != round) return;
  const int n = n_list[b];
  const int64_t pbase = rp.cand_base[b];
  const int q0 = qoff[b], Lq = qoff[b + 1] - q0;
  const int ngroups = (n + D - 1) / D;
  const int kfull = TAIL ? (ldim & ~7) : DIM;
  const int ql = lane & 31, half = lane >> 5, kp = lane & 3;
  for (int qt = 0; qt < Lq; qt += 32) {              // block-uniform (barriers inside)
    const int nq = min(32, Lq - qt);
    if constexpr (!TAIL) {
      __syncthreads();
      for (int w = tid; w < 32 * DIM; w += 256) {    // x[t][k] -> part (k % 8) / 2, slot 2 (k / 8) + (k & 1)
        const int t = w / DIM, k = w - t * DIM;
        sX[t * XT + ((k & 7) >> 1) * XK + 2 * (k >> 3) + (k & 1)] = t < nq ? qrows[(int64_t)(q0 + qt + t) * DIM + k] : 0.f;
      }
      __syncthreads();
    }
    for (int g = (int)blockIdx.x * 4 + wave; g < ngroups; g += (int)gridDim.x * 4) {
      const int i0 = g * D;
      const bool dl = lane < D && i0 + lane < n;
      const uint4 m = meta[pbase + (dl ? i0 + lane : 0)];
      const int nd = dl ? (int)m.y : 0;
      const uint32_t cl_lo = m.z, cl_hi = m.w & 0xFFu;
      int incl = nd;                          // lanes 0..D-1: where document `lane`'s codes start in the group's pair sequence
#pragma unroll
      for (int o = 1; o < D; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int excl = incl - nd;
      int st[D + 1];                          // wave-uniform
#pragma unroll
      for (int k = 0; k < D; ++k) st[k] = __builtin_amdgcn_readlane(excl, k);
      st[D] = __builtin_amdgcn_readlane(incl, D - 1);
      const int total = st[D];
      for (int w = lane; w < D * 33; w += 64) (&sM[wave][0][0])[w] = NP_NEG_INF;
      for (int c0 = 0; c0 < total; c0 += CH) {
        const int p = min(c0 + lane, total - 1);        // lanes past the end repeat the last pair (their sums are never read)
        int d = 0;
#pragma unroll
        for (int k = 1; k < D; ++k) d += (p >= st[k]) ? 1 : 0;
        const int j = p - __shfl(excl, d);
        const int64_t cl = (int64_t)(uint32_t)__shfl((int)cl_lo, d) | ((int64_t)(uint32_t)__shfl((int)cl_hi, d) << 32);
        const uint32_t code = codes[cl + j];
        if constexpr (TAIL) {
          const float4* src = reinterpret_cast<const float4*>(centroids + (int64_t)code * DIM);
          float4 cr[DIM / 4];
#pragma unroll
          for (int k4 = 0; k4 < DIM / 4; ++k4) cr[k4] = src[k4];
          for (int qq = 0; qq < nq; ++qq) {
            const float* xq = qrows + (int64_t)(q0 + qt + qq) * DIM;      // wave-uniform: scalar loads
            f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f}, p45 = {0.f, 0.f}, p67 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < DIM; k += 8) {
              if (k < kfull) {
                const float4 y0 = cr[k / 4], y1 = cr[k / 4 + 1];
                p01 = p01 + (f32x2){xq[k], xq[k + 1]} * (f32x2){y0.x, y0.y};
                p23 = p23 + (f32x2){xq[k + 2], xq[k + 3]} * (f32x2){y0.z, y0.w};
                p45 = p45 + (f32x2){xq[k + 4], xq[k + 5]} * (f32x2){y1.x, y1.y};
                p67 = p67 + (f32x2){xq[k + 6], xq[k + 7]} * (f32x2){y1.z, y1.w};
              }
            }
            float sum = 0.f;
            sum = sum + (p01.x + p45.x);
            sum = sum + (p01.y + p45.y);
            sum = sum + (p23.x + p67.x);
            sum = sum + (p23.y + p67.y);
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float4 y = cr[k / 4];
              const float yk = (k & 3) == 0 ? y.x : (k & 3) == 1 ? y.y : (k & 3) == 2 ? y.z : y.w;
              if (k >= kfull && k < ldim) sum = sum + xq[k] * yk;
            }
            sS[wave][qq * SS + lane] = sum;
          }
        } else {
          // the quad's PQ pairs: this lane's quarter (k-part kp) of their centroid rows
          f32x2 yr[PQ][DIM / 8];
#pragma unroll
          for (int c = 0; c < PQ; ++c) {
            const uint32_t cq = (uint32_t)__shfl((int)code, (lane >> 2) * PQ + c);
            const float* r = centroids + (int64_t)cq * DIM + 2 * kp;
#pragma unroll
            for (int i = 0; i < DIM / 8; ++i) yr[c][i] = *reinterpret_cast<const f32x2*>(r + 8 * i);
          }
          const float* xb = &sX[kp * XK];
          float4 x[DIM / 16];
#pragma unroll
          for (int jj = 0; jj < DIM / 16; ++jj) x[jj] = *reinterpret_cast<const float4*>(xb + 4 * jj);
          for (int qq = 0; qq < nq; ++qq) {
            const float* xnext = xb + min(qq + 1, nq - 1) * XT;
            f32x2 pp[PQ];
#pragma unroll
            for (int c = 0; c < PQ; ++c) pp[c] = (f32x2){0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < DIM / 16; ++jj) {
              const float4 xa = x[jj];
              const f32x2 xe = {xa.x, xa.y}, xo = {xa.z, xa.w};      // i = 2 jj, 2 jj + 1
#pragma unroll
              for (int c = 0; c < PQ; ++c) pp[c] = pp[c] + xe * yr[c][2 * jj];
#pragma unroll
              for (int c = 0; c < PQ; ++c) pp[c] = pp[c] + xo * yr[c][2 * jj + 1];
              x[jj] = *reinterpret_cast<const float4*>(xnext + 4 * jj);      // the next token's piece, a token ahead of its use
            }
            // lane kp holds (p[2kp], p[2kp+1]): kp 0 and 1 fold with kp 2 and 3, then kp 0 takes kp 1's two terms
            float sums[PQ];
#pragma unroll
            for (int c = 0; c < PQ; ++c) {
              const float a = pp[c].x + quad_perm<0x4E>(pp[c].x);      // kp 0: p0 + p4   kp 1: p2 + p6
              const float e = pp[c].y + quad_perm<0x4E>(pp[c].y);      // kp 0: p1 + p5   kp 1: p3 + p7
              float sum = 0.f;
              sum = sum + a;
              sum = sum + e;
              sum = sum + quad_perm<0xB1>(a);
              sum = sum + quad_perm<0xB1>(e);
              sums[c] = sum;
            }
            if (kp == 0) {
              float* dst = &sS[wave][qq * SS + (lane >> 2) * PQ];
              if constexpr (PQ == 4) *reinterpret_cast<float4*>(dst) = make_float4(sums[0], sums[1], sums[2], sums[3]);
              else if constexpr (PQ == 2) *reinterpret_cast<float2*>(dst) = make_float2(sums[0], sums[1]);
              else dst[0] = sums[0];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // per-token maxima of the documents this chunk touches: lanes = (token, even / odd pairs of the segment)
        const int cend = min(c0 + CH, total);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const int s0 = max(st[dd], c0) - c0, s1 = min(st[dd + 1], cend) - c0;
          if (s1 <= s0) continue;
          float mm = NP_NEG_INF;
          for (int e = s0 + half; e < s1; e += 2) {
            const float v = sS[wave][ql * SS + e];
            if (v > mm) mm = v;
          }
          const float other = __shfl_xor(mm, 32);
          if (other > mm) mm = other;
          if (half == 0 && ql < nq) {
            const float cur = sM[wave][dd][ql];
            if (mm > cur) sM[wave][dd][ql] = mm;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                // sS is rewritten by the next chunk
      }
      if (dl) {
        float score = qt ? approx[pbase + i0 + lane] : 0.f;      // the q-ordered sum continues over the token tiles
        for (int qq = 0; qq < nq; ++qq) {
          const float mm = sM[wave][lane][qq];
          if (mm > NP_NEG_INF) score = score + mm;
        }
        approx[pbase + i0 + lane] = score;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                  // sM is reset by the next group
    }
  }
}

// work counters (candidate tokens / distinct codes) of a record list, for the paths whose scoring kernel carries none
__global__ void __launch_bounds__(256) count_work_kernel(const uint4* __restrict__ meta, const int32_t* __restrict__ n_list,
                                                         RoundPlan rp, int round, Counters* ctr) {
  __shared__ unsigned long long s_cnt[2];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  if (rp.round_of[b] != round) return;
  const int n = n_list[b];
  const int64_t pbase = rp.cand_base[b];
  if (tid == 0) s_cnt[0] = s_cnt[1] = 0;
  __syncthreads();
  unsigned long long toks = 0, ucnt = 0;
  for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
    const uint4 m = meta[pbase + i];
    toks += (unsigned long long)(m.w >> 8);
    ucnt += (unsigned long long)m.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    toks += __shfl_xor(toks, o);
    ucnt += __shfl_xor(ucnt, o);
  }
  if (lane == 0 && toks) {
    atomicAdd(&s_cnt[0], toks);
    atomicAdd(&s_cnt[1], ucnt);
  }
  __syncthreads();
  if (tid == 0 && s_cnt[0]) {
    atomicAdd(&ctr->n_cand_tokens, s_cnt[0]);
    atomicAdd(&ctr->n_cand_codes, s_cnt[1]);
  }
}

// per query: threshold on the GEMM-valued scores, list2 = survivors with G >= tau_G - 2 eps (see above)
__global__ void __launch_bounds__(1024) gcut_kernel(const float* __restrict__ approx, const uint4* __restrict__ in_meta,
                                                    const int32_t* __restrict__ n_in, RoundPlan rp, int round, int n_sel,
                                                    const float* __restrict__ qinv, const uint32_t* __restrict__ qflag,
                                                    const int32_t* __restrict__ qoff, uint4* __restrict__ out_meta,
                                                    int32_t* __restrict__ n_out) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_rem, s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (rp.round_of[b] != round) return;
  const int n = n_in[b];
  const int64_t pbase = rp.cand_base[b];
  const float* ap = approx + pbase;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  bool all = n <= n_sel || qflag[b] != 0 || n_sel <= 0;
  float thr = NP_NEG_INF;
  if (!all) {
    if (tid == 0) {
      s_prefix = 0;
      s_rem = (uint32_t)n_sel;
    }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const uint32_t pre = s_prefix;
      for (int i = tid; i < n; i += 1024) {
        const uint32_t key = okey(ap[i]);
        if (pass == 0 || (key >> (shift + 8)) == pre) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t rem = s_rem, cum = 0;
        int bin = 255;
        for (; bin > 0; --bin) {
          if (cum + hist[bin] >= rem) break;
          cum += hist[bin];
        }
        s_prefix = (pre << 8) | (uint32_t)bin;
        s_rem = rem - cum;
      }
      __syncthreads();
    }
    const uint32_t tau = s_prefix;
    if (tau == 0) {
      all = true;   // the cut sits among non-finite scores: keep everything
    } else {
      const int Lq = qoff[b + 1] - qoff[b];
      const float s = 1.0f / qinv[b];
      const float eps = 1.5f * (152.0f * (float)Lq + 2.0f * (float)Lq * (float)Lq) * 5.9604645e-8f * s;
      thr = unkey(tau) - 2.0f * eps;
    }
  }
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool keep = i < n && (all || !(ap[i] < thr));   // NaN scores are kept (never dropped by a bound)
    if (keep) out_meta[pbase + atomicAdd(&s_cnt, 1u)] = in_meta[pbase + i];
  }
  __syncthreads();
  if (tid == 0) n_out[b] = (int32_t)s_cnt;
}

// ---------------------------------------------------------------------------------------------
// block bitonic sort, descending, n = power of two, in LDS
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* s, int n, int tid, int nthreads) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < n; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = s[i], c = s[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < c) : (a > c)) {
            s[i] = c;
            s[ixj] = a;
          }
        }
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// S5  per query: the n_sel best candidates by (approx desc [finite first], doc id asc), sorted.
// key64 = okey(approx) << 32 | (0xFFFFFFFF - global_doc_id)
// ---------------------------------------------------------------------------------------------
struct SelectP {
  const float* approx;      // [pool]
  const uint32_t* cand;     // [pool] shard-local doc ids
  int cand_step;            // u32 words between consecutive doc ids (1 = id array, 4 = the x field of 16-B records)
  RoundPlan rp;
  int round;
  const int32_t* n_cand;
  int64_t doc_begin;
  int n_sel;    // min(n_full_scores, max(n_full_scores/4, top_k))
  int NSELP;    // pow2 >= n_sel
  uint64_t* sel_keys;   // [B][n_sel]
  uint32_t* sel_doc;    // [B][n_sel] shard-local doc
  int32_t* nsel_out;    // [B]
  Counters* ctr;        // n_survivors += n when non-NULL (the filter's survivor lists)
};

__global__ void __launch_bounds__(1024) select_kernel(SelectP p) {
  extern __shared__ uint64_t s_sel[];
  __shared__ uint32_t hist[256];
  __shared__ uint64_t s_prefix;
  __shared__ uint32_t s_rem, s_ngt, s_done;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (p.rp.round_of[b] != p.round) return;
  const int n = p.n_cand[b];
  const float* ap = p.approx + p.rp.cand_base[b];
  const uint32_t* cd = p.cand + p.rp.cand_base[b] * p.cand_step;
  const int cstep = p.cand_step;
  const int nsel = min(p.n_sel, n);
  // rank key of search.rs:460's stable sort over ascending doc ids: (approx desc [finite first], doc id asc);
  // doc ids are unique, so the keys are too and the candidates may arrive in any order
  auto comp_of = [&](int i) {
    return ((uint64_t)okey(ap[i]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(p.doc_begin + cd[(int64_t)i * cstep]));
  };
  for (int i = tid; i < p.NSELP; i += 1024) s_sel[i] = 0;
  if (tid == 0) s_ngt = 0;
  __syncthreads();
  if (nsel > 0) {
    if (n <= nsel) {
      for (int i = tid; i < n; i += 1024) s_sel[i] = comp_of(i);
    } else {
      // radix select of the nsel-th largest 64-bit key, 8 bits per pass from the top; the low word (doc id) is only
      // walked when the score bits alone leave a tie at the cut
      if (tid == 0) {
        s_prefix = 0;
        s_rem = (uint32_t)nsel;
        s_done = 0;
      }
      __syncthreads();
      for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if (s_done) break;
        const uint64_t pre = s_prefix;
        for (int i = tid; i < n; i += 1024) {
          const uint64_t key = comp_of(i);
          if (pass == 0 || (key >> (shift + 8)) == pre) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          uint32_t rem = s_rem, cum = 0;
          int bin = 255;
          for (; bin > 0; --bin) {
            if (cum + hist[bin] >= rem) break;
            cum += hist[bin];
          }
          s_prefix = (pre << 8) | (uint64_t)bin;
          s_rem = rem - cum;
          // every key of the cut bin is taken: the remaining (lower) bits cannot matter
          if (hist[bin] == rem - cum) s_done = (uint32_t)(pass + 1);
        }
        __syncthreads();
      }
      const int npass = s_done ? (int)s_done : 8;
      const uint64_t tau = s_prefix;                 // top 8 * npass bits of the cut
      const int sh = 64 - 8 * npass;
      for (int i = tid; i < n; i += 1024) {
        const uint64_t key = comp_of(i);
        if ((sh == 0 ? key : (key >> sh)) >= tau) s_sel[atomicAdd(&s_ngt, 1u)] = key;
      }
    }
  }
  bitonic_sort_desc(s_sel, p.NSELP, tid, 1024);
  for (int j = tid; j < p.n_sel; j += 1024) {
    const uint64_t c = (j < nsel) ? s_sel[j] : 0ull;
    p.sel_keys[(int64_t)b * p.n_sel + j] = c;
    p.sel_doc[(int64_t)b * p.n_sel + j] =
        (j < nsel) ? (uint32_t)((int64_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull)) - p.doc_begin) : 0u;
  }
  if (tid == 0) {
    p.nsel_out[b] = nsel;
    if (p.ctr) atomicAdd(&p.ctr->n_survivors, (unsigned long long)n);
  }
}

// ---------------------------------------------------------------------------------------------
// S6  exact MaxSim.  One wave walks a document in 32-token tiles.  Lane (tok = lane&31,
// half = lane>>5) unpacks its half of the token's residual bytes, adds the centroid row, the two
// halves exchange their sum of squares with one shuffle, and the normalised values ARE the MFMA
// A fragment (no LDS staging of D).  S = D.Q^T accumulates on the matrix cores with tokens as
// MFMA rows and query tokens as columns, so the row-max over document tokens is an in-lane max.
//   decompress: out = centroid + wlut[segment]; row /= max(||row||, 1e-12)   codec.rs:443-467
//   maxsim:     sum_q max_t S[q,t], non-finite entries ignored                maxsim.rs:281-291
// ---------------------------------------------------------------------------------------------
struct ExactP {
  const float* Qt;          // [B][DIM][LQP] f32
  const __bf16* Qb;         // [B][LQP][DIM]
  const __bf16* Qb_lo;      // [B][LQP][DIM] bf16(q - Qb)
  const float* QCT;         // [B][KP][LQP] from S1
  int64_t KP;
  const float* inv_norm;    // [T] 1 / max(||centroid + residual||, 1e-12), derived at index open
  const int32_t* qoff;
  int LQP;
  const float* centroids;
  const float* wlut;
  CodeArr codes;
  const uint8_t* residuals;
  const int64_t* doc_off;
  const uint64_t* sel_keys; // [B][n_sel]
  const uint32_t* sel_doc;
  const int32_t* nsel;      // [B]
  const uint64_t* cut;      // [B] or NULL
  int n_sel;
  float* exact;             // [B][n_sel]
  Counters* ctr;
  int xcd_B;                // > 0: 1-D grid, workgroup w serves query (w/8/gx)*8 + w%8 so a query stays on one XCD
  int gx;                   // workgroups per query
  const uint32_t* qflag;    // [B] query has a non-finite or huge value (prep_queries_kernel)
  int fast_ok;              // index values finite and bounded: with an unflagged query every S6 product is finite
  int qt0;                  // exact_qct_kernel<.., NQT = 1, ..>: the 32-token query tile this launch scores (queries longer than
  int acc;                  // 32 tokens take one launch per tile); acc = continue the q-ordered sum from exact[] (tiles > 0)
  float pad_ss;             // (DIM - file dim) * wlut[0]^2: what the padding of a stored row adds to an inline sum of squares
};                          // (pad centroid values are 0 and pad residual bytes are 0, so every pad dim reads exactly wlut[0]; the
                            //  products themselves vanish against the zero-padded query).  0 for an unpadded index.

#define NP_EXACT_DPW 4   // documents per wave

template <int NBITS>
__device__ __forceinline__ float seg_weight(const float* sW, uint32_t byte, int e) {
  constexpr uint32_t MASK = (1u << NBITS) - 1u;
  return sW[(byte >> (8 - NBITS * (e + 1))) & MASK];  // segment e: 0 = highest bits = first dim
}

template <int DIM, int NBITS, int NQT>
__global__ void __launch_bounds__(256) exact_f32_kernel(ExactP p) {
  constexpr int H = DIM / 2;              // dims per lane
  constexpr int PD = DIM * NBITS / 8;     // bytes per token
  constexpr int PH = PD / 2;              // bytes per lane
  constexpr int PER = 8 / NBITS;          // dims per byte
  static_assert(PH % 4 == 0 && H % 4 == 0, "unsupported DIM/NBITS");
  extern __shared__ float smem[];
  const int LQP = p.LQP;
  float* sQ = smem;                       // [DIM][LQP]
  float* sW = smem + DIM * LQP;           // [1<<NBITS]
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < DIM * LQP; i += 256) sQ[i] = p.Qt[(int64_t)b * DIM * LQP + i];
  if (tid < (1 << NBITS)) sW[tid] = p.wlut[tid];
  __syncthreads();
  const int lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int Lq = p.qoff[b + 1] - p.qoff[b];
  const int nqt = (Lq + 31) >> 5;
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  unsigned long long toks = 0, ndocs = 0;
  for (int dd = 0; dd < NP_EXACT_DPW; ++dd) {
    const int j = (blockIdx.x * 4 + wave) * NP_EXACT_DPW + dd;
    if (j >= nsel) break;
    const int64_t oj = (int64_t)b * p.n_sel + j;
    if (p.sel_keys[oj] < cut) {
      if (lane == 0) p.exact[oj] = 0.f;
      continue;
    }
    const uint32_t doc = p.sel_doc[oj];
    const int64_t off = p.doc_off[doc];
    const int len = (int)(p.doc_off[doc + 1] - off);
    toks += (unsigned long long)len;
    ++ndocs;
    float m[NQT];
#pragma unroll
    for (int x = 0; x < NQT; ++x) m[x] = NP_NEG_INF;
    for (int t0 = 0; t0 < len; t0 += 32) {
      const int tt = t0 + li;
      const bool valid = tt < len;
      const int64_t tok = off + (valid ? tt : len - 1);
      const uint32_t code = p.codes[tok];
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(p.residuals + tok * PD + kk * PH);
      const float4* cp = reinterpret_cast<const float4*>(p.centroids + (int64_t)code * DIM + kk * H);
      float v[H];
      float ss = 0.f;
#pragma unroll
      for (int w = 0; w < PH / 4; ++w) {
        const uint32_t word = rp[w];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t byte = (word >> (8 * i)) & 0xFFu;
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int jdim = (w * 4 + i) * PER + e;
            const float c = reinterpret_cast<const float*>(cp)[jdim];
            const float x = c + seg_weight<NBITS>(sW, byte, e);
            v[jdim] = x;
            ss = fmaf(x, x, ss);
          }
        }
      }
      // 1/||row|| is applied to the MFMA output rows (S[t][q] = rn[t] * <raw_t, q>) instead of to
      // the 64 fragment values; lane li holds rn of token t0+li, row r of this lane needs token
      // t0 + mfma_row(r, kk).
      const float tot = ss + __shfl_xor(ss, 32) - p.pad_ss;
      const float rn = valid ? 1.0f / fmaxf(sqrtf(tot), 1e-12f) : 0.f;
      float rrow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rrow[r] = __shfl(rn, mfma_row(r, kk));
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        if (qt < nqt) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const float* qb = sQ + (kk * H) * LQP + qt * 32 + li;
#pragma unroll
          for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s], qb[s * LQP], acc, 0, 0, 0);
          float mm = m[qt];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = t0 + mfma_row(r, kk);
            const float x = acc[r] * rrow[r];
            if (trow < len && finitef(x)) mm = fmaxf(mm, x);
          }
          m[qt] = mm;
        }
      }
    }
    float total = 0.f;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt < nqt) {
        const float mm = fmaxf(m[qt], __shfl_xor(m[qt], 32));
        const int nq = min(32, Lq - qt * 32);
        for (int qi = 0; qi < nq; ++qi) {
          const float x = readlane_f(mm, qi);
          if (x > NP_NEG_INF) total += x;
        }
      }
    }
    if (lane == 0) p.exact[oj] = total;
  }
  (void)toks;
  (void)ndocs;
}

// bf16 MFMA variant (precision = 1): A fragment s of lane (tok, kk) = dims [16s + 8kk, +8).
template <int DIM, int NBITS, int NQT>
__global__ void __launch_bounds__(256) exact_bf16_kernel(ExactP p) {
  constexpr int NS = DIM / 16;            // MFMA k-steps
  constexpr int PD = DIM * NBITS / 8;
  constexpr int PER = 8 / NBITS;
  static_assert(DIM % 16 == 0 && (NBITS == 2 || NBITS == 4), "unsupported DIM/NBITS");
  __shared__ float sW[1 << NBITS];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < (1 << NBITS)) sW[tid] = p.wlut[tid];
  __syncthreads();
  const int LQP = p.LQP;
  const int lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int Lq = p.qoff[b + 1] - p.qoff[b];
  const int nqt = (Lq + 31) >> 5;
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  const __bf16* Qb = p.Qb + (int64_t)b * LQP * DIM;
  bf16x8 bq0[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) bq0[s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)li * DIM + 16 * s + 8 * kk);
  unsigned long long toks = 0, ndocs = 0;
  for (int dd = 0; dd < NP_EXACT_DPW; ++dd) {
    const int j = (blockIdx.x * 4 + wave) * NP_EXACT_DPW + dd;
    if (j >= nsel) break;
    const int64_t oj = (int64_t)b * p.n_sel + j;
    if (p.sel_keys[oj] < cut) {
      if (lane == 0) p.exact[oj] = 0.f;
      continue;
    }
    const uint32_t doc = p.sel_doc[oj];
    const int64_t off = p.doc_off[doc];
    const int len = (int)(p.doc_off[doc + 1] - off);
    toks += (unsigned long long)len;
    ++ndocs;
    float m[NQT];
#pragma unroll
    for (int x = 0; x < NQT; ++x) m[x] = NP_NEG_INF;
    for (int t0 = 0; t0 < len; t0 += 32) {
      const int tt = t0 + li;
      const bool valid = tt < len;
      const int64_t tok = off + (valid ? tt : len - 1);
      const uint32_t code = p.codes[tok];
      const uint8_t* rp = p.residuals + tok * PD;
      const float* cp = p.centroids + (int64_t)code * DIM;
      bf16x8 a[NS];
      float ss = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int d0 = 16 * s + 8 * kk;
        const float4 c0 = *reinterpret_cast<const float4*>(cp + d0);
        const float4 c1 = *reinterpret_cast<const float4*>(cp + d0 + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t word;
        if (NBITS == 4) word = *reinterpret_cast<const uint32_t*>(rp + d0 / 2);
        else word = *reinterpret_cast<const uint16_t*>(rp + d0 / 4);
#pragma unroll
        for (int i = 0; i < 8 / PER; ++i) {
          const uint32_t byte = (word >> (8 * i)) & 0xFFu;
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const float x = cc[i * PER + e] + seg_weight<NBITS>(sW, byte, e);
            a[s][i * PER + e] = (__bf16)x;   // un-normalised; rows are scaled after the MFMA
            ss = fmaf(x, x, ss);
          }
        }
      }
      const float tot = ss + __shfl_xor(ss, 32) - p.pad_ss;
      const float rn = valid ? 1.0f / fmaxf(sqrtf(tot), 1e-12f) : 0.f;
      float rrow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rrow[r] = __shfl(rn, mfma_row(r, kk));
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        if (qt < nqt) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            bf16x8 bq;
            if (qt == 0) bq = bq0[s];
            else bq = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)(qt * 32 + li) * DIM + 16 * s + 8 * kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq, acc, 0, 0, 0);
          }
          float mm = m[qt];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int trow = t0 + mfma_row(r, kk);
            const float x = acc[r] * rrow[r];
            if (trow < len && finitef(x)) mm = fmaxf(mm, x);
          }
          m[qt] = mm;
        }
      }
    }
    float total = 0.f;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt < nqt) {
        const float mm = fmaxf(m[qt], __shfl_xor(m[qt], 32));
        const int nq = min(32, Lq - qt * 32);
        for (int qi = 0; qi < nq; ++qi) {
          const float x = readlane_f(mm, qi);
          if (x > NP_NEG_INF) total += x;
        }
      }
    }
    if (lane == 0) p.exact[oj] = total;
  }
  (void)toks;
  (void)ndocs;
}

// ---------------------------------------------------------------------------------------------
// S6, QC-reuse form (precision 1 and 2).  With D_t = (C[code_t] + R_t) / n_t (codec.rs:443-467):
//     Q . D_t  =  ( QC[q, code_t]  +  Q . R_t ) / n_t
// QC[q, code_t] is S1's exact-f32 output (one 128-B QCT line per token) and becomes the MFMA
// accumulator's initial value; only the residual part goes through the matrix cores.  R_t takes 2^nbits
// distinct values, so a 256-entry byte -> packed-bf16 LUT in LDS turns the packed residual bytes
// directly into MFMA A fragments (no per-value VALU unpack, no centroid gather, no norm reduction):
// per token the kernel reads pd residual bytes + a code + 1/n_t (derived at index open).
// SPLIT = 1: plain bf16.  SPLIT = 3: R and Q are split hi+lo in bf16 and hi.hi + lo.hi + hi.lo are
// accumulated in f32 -- the residual term is then f32-accurate (~2^-17 relative).
// ---------------------------------------------------------------------------------------------
template <int DIM, int NBITS, int NQT, int SPLIT>
__global__ void __launch_bounds__(256) exact_qc_kernel(ExactP p) {
  constexpr int NS = DIM / 16;            // MFMA k-steps
  constexpr int PD = DIM * NBITS / 8;     // bytes per token
  constexpr int PH = PD / 2;              // bytes per lane: lane (tok, kk) owns dims [kk*DIM/2, +DIM/2)
  constexpr int NW = PH / 4;              // residual dwords per lane
  constexpr int WPB = (NBITS == 4) ? 1 : 2;  // u32 words of packed bf16 per residual byte (2 or 4 values)
  static_assert(DIM % 32 == 0 && (NBITS == 2 || NBITS == 4) && PH % 4 == 0, "unsupported DIM/NBITS");
  // byte -> {hi words, lo words}: one LDS read per residual byte returns both halves of the split
  __shared__ uint32_t lut[256 * WPB * 2];
  const int b = blockIdx.y, tid = threadIdx.x;
  {
    constexpr int PER = 8 / NBITS;
    constexpr uint32_t MASK = (1u << NBITS) - 1u;
    uint16_t hh[4] = {0, 0, 0, 0}, ll[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < PER; ++e) {   // first dim of the byte = highest segment -> low half-word
      const float w = p.wlut[((uint32_t)tid >> (8 - NBITS * (e + 1))) & MASK];
      const __bf16 h = (__bf16)w;
      const __bf16 l = (__bf16)(w - (float)h);
      hh[e] = __builtin_bit_cast(uint16_t, h);
      ll[e] = __builtin_bit_cast(uint16_t, l);
    }
#pragma unroll
    for (int w2 = 0; w2 < WPB; ++w2) {
      lut[(tid * WPB + w2) * 2 + 0] = (uint32_t)hh[2 * w2] | ((uint32_t)hh[2 * w2 + 1] << 16);
      lut[(tid * WPB + w2) * 2 + 1] = (uint32_t)ll[2 * w2] | ((uint32_t)ll[2 * w2 + 1] << 16);
    }
  }
  __syncthreads();
  const uint2* lut2 = reinterpret_cast<const uint2*>(lut);
  const int LQP = p.LQP;
  const int lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int Lq = p.qoff[b + 1] - p.qoff[b];
  const int nqt = (Lq + 31) >> 5;
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  // slot (s, kk, e) of the MFMA k dimension <-> dim kk*DIM/2 + 8s + e, for A (tokens) and B (query) alike
  const __bf16* Qb = p.Qb + (int64_t)b * LQP * DIM + kk * (DIM / 2);
  const __bf16* Ql = p.Qb_lo + (int64_t)b * LQP * DIM + kk * (DIM / 2);
  const char* QCb = reinterpret_cast<const char*>(p.QCT + (int64_t)b * p.KP * LQP) + li * 4;
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  bf16x8 bh0[NS], bl0[SPLIT == 3 ? NS : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    bh0[s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)li * DIM + 8 * s);
    if (SPLIT == 3) bl0[s] = *reinterpret_cast<const bf16x8*>(Ql + (int64_t)li * DIM + 8 * s);
  }
  unsigned long long toks = 0, ndocs = 0;
  for (int dd = 0; dd < NP_EXACT_DPW; ++dd) {
    const int j = (blockIdx.x * 4 + wave) * NP_EXACT_DPW + dd;
    if (j >= nsel) break;
    const int64_t oj = (int64_t)b * p.n_sel + j;
    if (p.sel_keys[oj] < cut) {
      if (lane == 0) p.exact[oj] = 0.f;
      continue;
    }
    const uint32_t doc = p.sel_doc[oj];
    const int64_t off = p.doc_off[doc];
    const int len = (int)(p.doc_off[doc + 1] - off);
    toks += (unsigned long long)len;
    ++ndocs;
    float m[NQT];
#pragma unroll
    for (int x = 0; x < NQT; ++x) m[x] = NP_NEG_INF;
    // software pipeline: the NEXT tile's code / 1/n / residual words are in flight during this tile
    uint32_t code_n = 0, rw_n[NW];
    float rn_n = 0.f;
    auto fetch = [&](int t0) {
      const int tt = t0 + li;
      const bool valid = tt < len;
      const int64_t tok = off + (valid ? tt : len - 1);
      code_n = p.codes[tok];
      rn_n = valid ? p.inv_norm[tok] : __builtin_nanf("");   // rows past the end become NaN and drop out of fmaxf
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(p.residuals + tok * PD + kk * PH);
      if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int w4 = 0; w4 < NW / 4; ++w4) {
          const uint4 v = reinterpret_cast<const uint4*>(rp)[w4];
          rw_n[4 * w4] = v.x; rw_n[4 * w4 + 1] = v.y; rw_n[4 * w4 + 2] = v.z; rw_n[4 * w4 + 3] = v.w;
        }
      } else if constexpr (NW % 2 == 0) {
#pragma unroll
        for (int w2 = 0; w2 < NW / 2; ++w2) {
          const uint2 v = reinterpret_cast<const uint2*>(rp)[w2];
          rw_n[2 * w2] = v.x; rw_n[2 * w2 + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int w1 = 0; w1 < NW; ++w1) rw_n[w1] = rp[w1];
      }
    };
    if (len > 0) fetch(0);
    for (int t0 = 0; t0 < len; t0 += 32) {
      const uint32_t code = code_n;
      const float rn = rn_n;
      uint32_t rw[NW];
#pragma unroll
      for (int w1 = 0; w1 < NW; ++w1) rw[w1] = rw_n[w1];
      if (t0 + 32 < len) fetch(t0 + 32);
      // per output row (token t0 + mfma_row(r, kk)): its code (for the QC line) and 1/n_t
      uint32_t crow[16];
      float rrow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        crow[r] = (uint32_t)__shfl((int)code, mfma_row(r, kk));
        rrow[r] = __shfl(rn, mfma_row(r, kk));
      }
      f32x16 acc0;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = *reinterpret_cast<const float*>(QCb + crow[r] * row_bytes);   // C-in = Q.C[code], q-tile 0
      // residual bytes -> bf16 A fragments (8 dims = 8*NBITS/8 bytes per k-step)
      bf16x8 ah[NS], al[SPLIT == 3 ? NS : 1];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        uint32_t wh[4], wl[4];
        if constexpr (NBITS == 4) {
          const uint32_t word = rw[s];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint2 e = lut2[(word >> (8 * i)) & 0xFFu];
            wh[i] = e.x;
            wl[i] = e.y;
          }
        } else {
          const uint32_t word = rw[s >> 1] >> (16 * (s & 1));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint32_t byte = (word >> (8 * i)) & 0xFFu;
            const uint2 e0 = lut2[byte * 2], e1 = lut2[byte * 2 + 1];
            wh[2 * i] = e0.x; wl[2 * i] = e0.y;
            wh[2 * i + 1] = e1.x; wl[2 * i + 1] = e1.y;
          }
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 vh = {wh[0], wh[1], wh[2], wh[3]};
        ah[s] = __builtin_bit_cast(bf16x8, vh);
        if constexpr (SPLIT == 3) {
          u32x4 vl = {wl[0], wl[1], wl[2], wl[3]};
          al[s] = __builtin_bit_cast(bf16x8, vl);
        }
      }
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        if (qt < nqt) {
          f32x16 acc;
          if (qt == 0) {
            acc = acc0;
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = *reinterpret_cast<const float*>(QCb + (crow[r] * row_bytes + qt * 128));
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            bf16x8 bh, bl;
            if (qt == 0) {
              bh = bh0[s];
              if constexpr (SPLIT == 3) bl = bl0[s];
            } else {
              bh = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)(qt * 32 + li) * DIM + 8 * s);
              if constexpr (SPLIT == 3) bl = *reinterpret_cast<const bf16x8*>(Ql + (int64_t)(qt * 32 + li) * DIM + 8 * s);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc, 0, 0, 0);
            if constexpr (SPLIT == 3) {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc, 0, 0, 0);
            }
          }
          float mm = m[qt];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = acc[r] * rrow[r];
            // maxsim.rs:284-291 ignores non-finite entries: x + (x - x) is x when finite and NaN for
            // +-inf / NaN, and fmaxf never returns a NaN operand
            mm = fmaxf(mm, x + (x - x));
          }
          m[qt] = mm;
        }
      }
    }
    float total = 0.f;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt < nqt) {
        const float mm = fmaxf(m[qt], __shfl_xor(m[qt], 32));
        const int nq = min(32, Lq - qt * 32);
        for (int qi = 0; qi < nq; ++qi) {
          const float x = readlane_f(mm, qi);
          if (x > NP_NEG_INF) total += x;
        }
      }
    }
    if (lane == 0) p.exact[oj] = total;
  }
  (void)toks;
  (void)ndocs;
}

// The byte -> packed-bf16 LUT holds {hi word, lo word} per byte: one ds_read_b64 per residual byte.  The MFMA operands want
// four hi words (and four lo words) in consecutive registers, so each k-step pays 6 v_mov to un-interleave the pairs;
// separate hi / lo arrays do not help (the compiler fuses the two reads into ds_read2st64_b32, same pairs).  Replicating
// the LUT to avoid bank conflicts (16 or 32 bank-aligned copies) was measured and does not pay: 0.64 -> 0.65 / 0.91 ms.
template <int DIM, int NBITS, int NQT, int SPLIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NQT == 1 ? 3 : 2))) exact_qct_kernel(ExactP p) {
  constexpr int NS = DIM / 16;            // MFMA k-steps
  constexpr int PD = DIM * NBITS / 8;     // bytes per token
  constexpr int PH = PD / 2;              // bytes per lane: lane (tok, kk) owns dims [kk*DIM/2, +DIM/2)
  constexpr int NW = PH / 4;              // residual dwords per lane
  constexpr int WPB = (NBITS == 4) ? 1 : 2;  // u32 words of packed bf16 per residual byte (2 or 4 values)
  static_assert(DIM % 32 == 0 && (NBITS == 2 || NBITS == 4) && PH % 4 == 0, "unsupported DIM/NBITS");
  // byte -> {hi words, lo words}: one LDS read per residual byte returns both halves of the split
  __shared__ uint32_t lut[256 * WPB * 2];
  // S6 gathers one 128-B row of the query's score table per token (the MFMA C-in): the same L2-miss-bound
  // access as S4.  With xcd_B set, all workgroups of a query run on ONE XCD (workgroup w -> XCD w % 8), so rows
  // are reused out of that XCD's L2 instead of being refetched by eight.
  int b = blockIdx.y, bx = blockIdx.x;
  if (p.xcd_B > 0) {
    const int slot = blockIdx.x >> 3;
    b = (slot / p.gx) * 8 + (blockIdx.x & 7);
    bx = slot % p.gx;
    if (b >= p.xcd_B) return;
  }
  const int tid = threadIdx.x;
  {
    constexpr int PER = 8 / NBITS;
    constexpr uint32_t MASK = (1u << NBITS) - 1u;
    uint16_t hh[4] = {0, 0, 0, 0}, ll[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < PER; ++e) {   // first dim of the byte = highest segment -> low half-word
      const float w = p.wlut[((uint32_t)tid >> (8 - NBITS * (e + 1))) & MASK];
      const __bf16 h = (__bf16)w;
      const __bf16 l = (__bf16)(w - (float)h);
      hh[e] = __builtin_bit_cast(uint16_t, h);
      ll[e] = __builtin_bit_cast(uint16_t, l);
    }
#pragma unroll
    for (int w2 = 0; w2 < WPB; ++w2) {
      lut[(tid * WPB + w2) * 2 + 0] = (uint32_t)hh[2 * w2] | ((uint32_t)hh[2 * w2 + 1] << 16);
      lut[(tid * WPB + w2) * 2 + 1] = (uint32_t)ll[2 * w2] | ((uint32_t)ll[2 * w2 + 1] << 16);
    }
  }
  __syncthreads();
  const uint2* lut2 = reinterpret_cast<const uint2*>(lut);
  const int LQP = p.LQP;
  const int lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  // One launch per 32-token query tile when NQT == 1 (p.qt0 = the tile): the two-tile instantiation needs 242 VGPRs (two
  // waves per SIMD) and ran a 48-token batch at 2.46 ms against 0.60 ms for 32 tokens; two one-tile launches re-read the
  // residuals but keep three waves per SIMD.  Tiles > 0 continue the q-ordered sum from exact[] (same order as one pass).
  const int qt0 = NQT == 1 ? p.qt0 : 0;
  const int Lq = max(0, min(32 * NQT, p.qoff[b + 1] - p.qoff[b] - 32 * qt0));   // query tokens of this launch's tile(s)
  const int nqt = (Lq + 31) >> 5;
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  const bool fast = p.fast_ok != 0 && p.qflag[b] == 0;   // workgroup-uniform
  if (qt0 > 0 && Lq == 0) return;                         // this query has no token in the tile: nothing to add
  // slot (s, kk, e) of the MFMA k dimension <-> dim kk*DIM/2 + 8s + e, for A (tokens) and B (query) alike
  const __bf16* Qb = p.Qb + ((int64_t)b * LQP + 32 * qt0) * DIM + kk * (DIM / 2);
  const __bf16* Ql = p.Qb_lo + ((int64_t)b * LQP + 32 * qt0) * DIM + kk * (DIM / 2);
  const char* QCb = reinterpret_cast<const char*>(p.QCT + (int64_t)b * p.KP * LQP) + 128 * qt0;   // row base; columns added per load
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  bf16x8 bh0[NS], bl0[SPLIT == 3 ? NS : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    bh0[s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)li * DIM + 8 * s);
    if (SPLIT == 3) bl0[s] = *reinterpret_cast<const bf16x8*>(Ql + (int64_t)li * DIM + 8 * s);
  }
  unsigned long long toks = 0, ndocs = 0;
  for (int dd = 0; dd < NP_EXACT_DPW; ++dd) {
    const int j = (bx * 4 + wave) * NP_EXACT_DPW + dd;
    if (j >= nsel) break;
    const int64_t oj = (int64_t)b * p.n_sel + j;
    if (p.sel_keys[oj] < cut) {
      if (lane == 0) p.exact[oj] = 0.f;
      continue;
    }
    const float prior = (NQT == 1 && p.acc) ? p.exact[oj] : 0.f;
    const uint32_t doc = p.sel_doc[oj];
    const int64_t off = p.doc_off[doc];
    const int len = (int)(p.doc_off[doc + 1] - off);
    toks += (unsigned long long)len;
    ++ndocs;
    // TRANSPOSED product S^T = Q . D^T: MFMA rows = query tokens, columns = document tokens, so lane
    // (tok = li, kk) owns COLUMN tok: its accumulator rows are q = mfma_row(r, kk), the C-in values
    // QC[code_tok][q] are four float4 pieces of the lane's OWN table row, 1/n_tok is the lane's own scalar,
    // and the max over tokens is an element-wise running max (cross-lane only once per document).
    float m[NQT][16];
#pragma unroll
    for (int x = 0; x < NQT; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) m[x][r] = NP_NEG_INF;
    // software pipeline: the NEXT tile's code / 1/n / residual words are in flight during this tile
    uint32_t code_n = 0, rw_n[NW];
    float rn_n = 0.f;
    auto fetch = [&](int t0) {
      const int tt = t0 + li;
      const bool valid = tt < len;
      const int64_t tok = off + (valid ? tt : len - 1);
      code_n = p.codes[tok];
      rn_n = valid ? p.inv_norm[tok] : __builtin_nanf("");   // rows past the end become NaN and drop out of fmaxf
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(p.residuals + tok * PD + kk * PH);
      if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int w4 = 0; w4 < NW / 4; ++w4) {
          const uint4 v = reinterpret_cast<const uint4*>(rp)[w4];
          rw_n[4 * w4] = v.x; rw_n[4 * w4 + 1] = v.y; rw_n[4 * w4 + 2] = v.z; rw_n[4 * w4 + 3] = v.w;
        }
      } else if constexpr (NW % 2 == 0) {
#pragma unroll
        for (int w2 = 0; w2 < NW / 2; ++w2) {
          const uint2 v = reinterpret_cast<const uint2*>(rp)[w2];
          rw_n[2 * w2] = v.x; rw_n[2 * w2 + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int w1 = 0; w1 < NW; ++w1) rw_n[w1] = rp[w1];
      }
    };
    if (len > 0) fetch(0);
    for (int t0 = 0; t0 < len; t0 += 32) {
      const uint32_t code = code_n;
      const float rn = rn_n;
      uint32_t rw[NW];
#pragma unroll
      for (int w1 = 0; w1 < NW; ++w1) rw[w1] = rw_n[w1];
      if (t0 + 32 < len) fetch(t0 + 32);
      f32x16 acc0;   // C-in = Q.C[code]: rows q = 8g + 4kk + (0..3) are one float4 of this token's QCT row
      {
        const char* qrow = QCb + code * row_bytes;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(qrow + (8 * g + 4 * kk) * 4);
          acc0[4 * g] = v.x; acc0[4 * g + 1] = v.y; acc0[4 * g + 2] = v.z; acc0[4 * g + 3] = v.w;
        }
      }
      // residual bytes -> bf16 A fragments (8 dims = 8*NBITS/8 bytes per k-step)
      bf16x8 ah[NS], al[SPLIT == 3 ? NS : 1];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        uint32_t wh[4], wl[4];
        if constexpr (NBITS == 4) {
          const uint32_t word = rw[s];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint2 e = lut2[(word >> (8 * i)) & 0xFFu];
            wh[i] = e.x;
            wl[i] = e.y;
          }
        } else {
          const uint32_t word = rw[s >> 1] >> (16 * (s & 1));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint32_t byte = (word >> (8 * i)) & 0xFFu;
            const uint2 e0 = lut2[byte * 2], e1 = lut2[byte * 2 + 1];
            wh[2 * i] = e0.x; wl[2 * i] = e0.y;
            wh[2 * i + 1] = e1.x; wl[2 * i + 1] = e1.y;
          }
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 vh = {wh[0], wh[1], wh[2], wh[3]};
        ah[s] = __builtin_bit_cast(bf16x8, vh);
        if constexpr (SPLIT == 3) {
          u32x4 vl = {wl[0], wl[1], wl[2], wl[3]};
          al[s] = __builtin_bit_cast(bf16x8, vl);
        }
      }
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        if (qt < nqt) {
          f32x16 acc;
          if (qt == 0) {
            acc = acc0;
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = *reinterpret_cast<const float4*>(QCb + (code * row_bytes + qt * 128 + (8 * g + 4 * kk) * 4));
              acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
            }
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            bf16x8 bh, bl;
            if (qt == 0) {
              bh = bh0[s];
              if constexpr (SPLIT == 3) bl = bl0[s];
            } else {
              bh = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)(qt * 32 + li) * DIM + 8 * s);
              if constexpr (SPLIT == 3) bl = *reinterpret_cast<const bf16x8*>(Ql + (int64_t)(qt * 32 + li) * DIM + 8 * s);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[s], acc, 0, 0, 0);   // rows = q, cols = tokens
            if constexpr (SPLIT == 3) {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[s], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[s], acc, 0, 0, 0);
            }
          }
          if (fast) {
            // every value is finite by construction (finite, bounded index and query: p.fast_ok / qflag), only the rows
            // past the end are NaN (rn): one multiply and one v_max_f32 per element -- the raw instruction returns the
            // non-NaN operand, and fmaxf() would add a canonicalising v_max of the loop-carried maximum
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float x = acc[r] * rn;
              asm("v_max_f32 %0, %1, %2" : "=v"(m[qt][r]) : "v"(m[qt][r]), "v"(x));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float x = acc[r] * rn;
              // maxsim.rs:284-291 ignores non-finite entries: x + (x - x) is x when finite and NaN for
              // +-inf / NaN (rn is NaN for tokens past the end), and fmaxf never returns a NaN operand
              m[qt][r] = fmaxf(m[qt][r], x + (x - x));
            }
          }
        }
      }
    }
    float total = prior;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt < nqt) {
        // max over the document's tokens: across the 32 lanes of each half, once per document
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = m[qt][r];
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
          m[qt][r] = v;
        }
        // q-ordered sum (maxsim.rs:284-291): q = 8g + 4kk + e lives in register 4g + e of half kk
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = qt * 32 + 8 * g + 4 * hk + e;
              const float x = readlane_f(m[qt][4 * g + e], hk * 32);
              if (q < Lq && x > NP_NEG_INF) total += x;
            }
      }
    }
    if (lane == 0) p.exact[oj] = total;
  }
  (void)toks;
  (void)ndocs;
}

// S6, QC-reuse form with the QUERY FRAGMENTS IN LDS and the C-in rows gathered one tile ahead (round 3; NQT = 1 per launch).
// exact_qct_kernel holds the query's 8 hi + 8 lo B fragments in 64 VGPRs for the whole kernel: 155 VGPRs, three waves per
// SIMD, no room to prefetch the C-in rows -- and its PMC profile is a stalled one (MFMA pipe 38 % busy, VALU + MFMA issue
// 0.25 ms of 0.59 ms, 15 M of 20 M C-in row requests miss the L2).  Here the fragments are read from LDS per k-step (rows
// padded by 16 B: the 1 KiB a wave reads per step spreads over all banks), which frees the registers for (a) the codes
// two tiles ahead and the C-in rows ONE TILE AHEAD of their MFMAs, so the L2-miss latency of the row gather overlaps a
// whole tile of matrix work, and (b) four waves per SIMD.  Arithmetic and summation order identical to exact_qct_kernel.
// (Round 5, measured and removed -- commit 122dafe: the hi query fragments held in 4 * NS VGPRs for the whole kernel, only the lo
// ones read from LDS per k-step: a quarter less LDS traffic, 143 VGPRs, three waves per SIMD instead of four -- S6 0.575 -> 0.616 ms.
// Per k-step a wave reads 2 KiB of LUT entries and 2 KiB of fragments, 32 LDS cycles against the 96 cycles of its 3 MFMAs x 4 SIMDs:
// LDS is the busiest unit (PMC: 65 % + 29 % conflict stalls) but trading a wave per SIMD for LDS bytes loses.)
template <int DIM, int NBITS, int SPLIT, int WPE /* waves per SIMD the register budget is cut for */>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) exact_qcl_kernel(ExactP p) {
  constexpr int NS = DIM / 16;
  constexpr int PD = DIM * NBITS / 8;
  constexpr int PH = PD / 2;
  constexpr int NW = PH / 4;
  constexpr int WPB = (NBITS == 4) ? 1 : 2;
  constexpr int QS = DIM + 8;   // LDS row stride of the query fragments in bf16 (16 B of padding)
  static_assert(DIM % 32 == 0 && (NBITS == 2 || NBITS == 4) && PH % 4 == 0, "unsupported DIM/NBITS");
  __shared__ uint32_t lut[256 * WPB * 2];
  __shared__ __attribute__((aligned(16))) __bf16 sQh[32 * QS];
  __shared__ __attribute__((aligned(16))) __bf16 sQl[SPLIT == 3 ? 32 * QS : 8];
  int b = blockIdx.y, bx = blockIdx.x;
  if (p.xcd_B > 0) {
    const int slot = blockIdx.x >> 3;
    b = (slot / p.gx) * 8 + (blockIdx.x & 7);
    bx = slot % p.gx;
    if (b >= p.xcd_B) return;
  }
  const int tid = threadIdx.x;
  {
    constexpr int PER = 8 / NBITS;
    constexpr uint32_t MASK = (1u << NBITS) - 1u;
    uint16_t hh[4] = {0, 0, 0, 0}, ll[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const float w = p.wlut[((uint32_t)tid >> (8 - NBITS * (e + 1))) & MASK];
      const __bf16 h = (__bf16)w;
      const __bf16 l = (__bf16)(w - (float)h);
      hh[e] = __builtin_bit_cast(uint16_t, h);
      ll[e] = __builtin_bit_cast(uint16_t, l);
    }
#pragma unroll
    for (int w2 = 0; w2 < WPB; ++w2) {
      lut[(tid * WPB + w2) * 2 + 0] = (uint32_t)hh[2 * w2] | ((uint32_t)hh[2 * w2 + 1] << 16);
      lut[(tid * WPB + w2) * 2 + 1] = (uint32_t)ll[2 * w2] | ((uint32_t)ll[2 * w2 + 1] << 16);
    }
  }
  const int LQP = p.LQP;
  const int qt0 = p.qt0;
  {  // this launch's 32-token query tile -> LDS (16 B per thread and step)
    const __bf16* Qb = p.Qb + ((int64_t)b * LQP + 32 * qt0) * DIM;
    const __bf16* Ql = p.Qb_lo + ((int64_t)b * LQP + 32 * qt0) * DIM;
    for (int i = tid; i < 32 * DIM / 8; i += 256) {
      const int r = i / (DIM / 8), c8 = i - r * (DIM / 8);
      *reinterpret_cast<uint4*>(&sQh[r * QS + 8 * c8]) = *reinterpret_cast<const uint4*>(Qb + (int64_t)r * DIM + 8 * c8);
      if constexpr (SPLIT == 3)
        *reinterpret_cast<uint4*>(&sQl[r * QS + 8 * c8]) = *reinterpret_cast<const uint4*>(Ql + (int64_t)r * DIM + 8 * c8);
    }
  }
  __syncthreads();
  const uint2* lut2 = reinterpret_cast<const uint2*>(lut);
  const int lane = tid & 63, li = lane & 31, kk = lane >> 5, wave = tid >> 6;
  const int Lq = max(0, min(32, p.qoff[b + 1] - p.qoff[b] - 32 * qt0));
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  const bool fast = p.fast_ok != 0 && p.qflag[b] == 0;
  if (qt0 > 0 && Lq == 0) return;
  const __bf16* qh = &sQh[li * QS + kk * (DIM / 2)];
  const __bf16* ql = &sQl[SPLIT == 3 ? li * QS + kk * (DIM / 2) : 0];
  const char* QCb = reinterpret_cast<const char*>(p.QCT + (int64_t)b * p.KP * LQP) + 128 * qt0 + 16 * kk;
  const uint32_t row_bytes = (uint32_t)LQP * 4u;
  for (int dd = 0; dd < NP_EXACT_DPW; ++dd) {
    const int j = (bx * 4 + wave) * NP_EXACT_DPW + dd;
    if (j >= nsel) break;
    const int64_t oj = (int64_t)b * p.n_sel + j;
    if (p.sel_keys[oj] < cut) {
      if (lane == 0) p.exact[oj] = 0.f;
      continue;
    }
    const float prior = p.acc ? p.exact[oj] : 0.f;
    const uint32_t doc = p.sel_doc[oj];
    const int64_t off = p.doc_off[doc];
    const int len = (int)(p.doc_off[doc + 1] - off);
    float m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] = NP_NEG_INF;
    // pipeline: codes two tiles ahead, C-in rows + residual words + 1/n one tile ahead
    auto tok_of = [&](int t0) { return off + min(t0 + li, len - 1); };
    uint32_t code1 = 0, code2 = 0;      // codes of tiles t+1, t+2
    uint32_t rw_n[NW];
    float rn_n = 0.f;
    float4 cin_n[4];
    auto load_rest = [&](int t0) {      // residual words and 1/n of tile t0
      const bool valid = t0 + li < len;
      const int64_t tok = tok_of(t0);
      rn_n = valid ? p.inv_norm[tok] : __builtin_nanf("");
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(p.residuals + tok * PD + kk * PH);
      if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int w4 = 0; w4 < NW / 4; ++w4) {
          const uint4 v = reinterpret_cast<const uint4*>(rp)[w4];
          rw_n[4 * w4] = v.x; rw_n[4 * w4 + 1] = v.y; rw_n[4 * w4 + 2] = v.z; rw_n[4 * w4 + 3] = v.w;
        }
      } else if constexpr (NW % 2 == 0) {
#pragma unroll
        for (int w2 = 0; w2 < NW / 2; ++w2) {
          const uint2 v = reinterpret_cast<const uint2*>(rp)[w2];
          rw_n[2 * w2] = v.x; rw_n[2 * w2 + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int w1 = 0; w1 < NW; ++w1) rw_n[w1] = rp[w1];
      }
    };
    auto load_cin = [&](uint32_t code) {   // rows q = 8g + 4kk + (0..3) of the token's QCT row: four float4
      const char* qrow = QCb + code * row_bytes;
#pragma unroll
      for (int g = 0; g < 4; ++g) cin_n[g] = *reinterpret_cast<const float4*>(qrow + 32 * g);
    };
    if (len > 0) {
      const uint32_t code0 = p.codes[tok_of(0)];
      code1 = p.codes[tok_of(32)];
      load_rest(0);
      load_cin(code0);
    }
    for (int t0 = 0; t0 < len; t0 += 32) {
      const float rn = rn_n;
      uint32_t rw[NW];
#pragma unroll
      for (int w1 = 0; w1 < NW; ++w1) rw[w1] = rw_n[w1];
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[4 * g] = cin_n[g].x; acc[4 * g + 1] = cin_n[g].y; acc[4 * g + 2] = cin_n[g].z; acc[4 * g + 3] = cin_n[g].w;
      }
      if (t0 + 32 < len) {        // next tile: its code arrived a tile ago
        code2 = p.codes[tok_of(t0 + 64)];
        load_rest(t0 + 32);
        load_cin(code1);
        code1 = code2;
      }
      // k-steps: the query fragments of step s+1 are read from LDS while step s runs; a scheduling barrier per step keeps
      // the compiler from hoisting every step's LDS reads (fragments + LUT words) to the top of the tile (that is the 64
      // VGPRs this kernel exists to give back)
      bf16x8 bh_c = *reinterpret_cast<const bf16x8*>(qh), bl_c = bh_c;
      if constexpr (SPLIT == 3) bl_c = *reinterpret_cast<const bf16x8*>(ql);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8 bh_n = bh_c, bl_n = bl_c;
        if (s + 1 < NS) {
          bh_n = *reinterpret_cast<const bf16x8*>(qh + 8 * (s + 1));
          if constexpr (SPLIT == 3) bl_n = *reinterpret_cast<const bf16x8*>(ql + 8 * (s + 1));
        }
        uint32_t wh[4], wl[4];
        if constexpr (NBITS == 4) {
          const uint32_t word = rw[s];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint2 e = lut2[(word >> (8 * i)) & 0xFFu];
            wh[i] = e.x;
            wl[i] = e.y;
          }
        } else {
          const uint32_t word = rw[s >> 1] >> (16 * (s & 1));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint32_t byte = (word >> (8 * i)) & 0xFFu;
            const uint2 e0 = lut2[byte * 2], e1 = lut2[byte * 2 + 1];
            wh[2 * i] = e0.x; wl[2 * i] = e0.y;
            wh[2 * i + 1] = e1.x; wl[2 * i + 1] = e1.y;
          }
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 vh = {wh[0], wh[1], wh[2], wh[3]};
        const bf16x8 ah = __builtin_bit_cast(bf16x8, vh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh_c, ah, acc, 0, 0, 0);   // rows = q, cols = tokens
        if constexpr (SPLIT == 3) {
          const u32x4 vl = {wl[0], wl[1], wl[2], wl[3]};
          const bf16x8 al = __builtin_bit_cast(bf16x8, vl);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh_c, al, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl_c, ah, acc, 0, 0, 0);
        }
        // (round 5, measured and removed: s_setprio 1 around these MFMAs, or around the LUT expansion instead: 0.60 / 0.61 ms
        // against 0.575 without the scalar branches that selected them)
        __builtin_amdgcn_sched_barrier(0);
        bh_c = bh_n;
        bl_c = bl_n;
      }
      if (fast) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = acc[r] * rn;
          asm("v_max_f32 %0, %1, %2" : "=v"(m[r]) : "v"(m[r]), "v"(x));
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = acc[r] * rn;
          m[r] = fmaxf(m[r], x + (x - x));
        }
      }
    }
    float total = prior;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = m[r];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
      m[r] = v;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int hk = 0; hk < 2; ++hk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = 8 * g + 4 * hk + e;
          const float x = readlane_f(m[4 * g + e], hk * 32);
          if (q < Lq && x > NP_NEG_INF) total += x;
        }
    if (lane == 0) p.exact[oj] = total;
  }
}

// ---------------------------------------------------------------------------------------------
// S7  stable top-k by (exact desc [finite first], approx rank asc)   (search.rs:496-515)
// ---------------------------------------------------------------------------------------------
struct TopkP {
  const float* exact;
  const uint64_t* sel_keys;
  const uint32_t* sel_doc;
  const int32_t* nsel;
  const uint64_t* cut;
  int n_sel, NSELP, top_k;
  int64_t doc_begin;
  const int64_t* doc_off;   // document token offsets (work counters)
  Counters* ctr;
  int64_t* out_ids;     // [B][top_k]
  float* out_scores;
  uint64_t* out_keys;   // may be NULL
  int32_t* out_counts;
};

__global__ void __launch_bounds__(1024) topk_kernel(TopkP p) {
  extern __shared__ uint64_t s_sel[];
  __shared__ int s_valid;
  __shared__ unsigned long long s_ntok;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nsel = p.nsel[b];
  const uint64_t cut = p.cut ? p.cut[b] : 0ull;
  if (tid == 0) {
    s_valid = 0;
    s_ntok = 0;
  }
  __syncthreads();
  int nv = 0;
  unsigned long long ntok = 0;
  for (int j = tid; j < p.NSELP; j += 1024) {
    uint64_t c = 0;
    if (j < nsel && p.sel_keys[(int64_t)b * p.n_sel + j] >= cut) {
      c = ((uint64_t)okey(p.exact[(int64_t)b * p.n_sel + j]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)j);
      ++nv;
      const uint32_t d = p.sel_doc[(int64_t)b * p.n_sel + j];
      ntok += (unsigned long long)(p.doc_off[d + 1] - p.doc_off[d]);
    }
    s_sel[j] = c;
  }
  if (nv) atomicAdd(&s_valid, nv);
  if (p.ctr) {   // the exact-scored work of this query (S6 itself carries no counters)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ntok += __shfl_xor(ntok, o);
    if ((tid & 63) == 0 && ntok) atomicAdd(&s_ntok, ntok);
  }
  bitonic_sort_desc(s_sel, p.NSELP, tid, 1024);
  const int cnt = min(p.top_k, s_valid);
  for (int i = tid; i < cnt; i += 1024) {
    const int j = (int)(0xFFFFFFFFu - (uint32_t)(s_sel[i] & 0xFFFFFFFFull));
    const int64_t oj = (int64_t)b * p.n_sel + j;
    p.out_ids[(int64_t)b * p.top_k + i] = p.doc_begin + (int64_t)p.sel_doc[oj];
    p.out_scores[(int64_t)b * p.top_k + i] = p.exact[oj];
    if (p.out_keys) p.out_keys[(int64_t)b * p.top_k + i] = p.sel_keys[oj];
  }
  if (tid == 0) {
    p.out_counts[b] = cnt;
    if (p.ctr && s_valid) {
      atomicAdd(&p.ctr->n_exact_docs, (unsigned long long)s_valid);
      atomicAdd(&p.ctr->n_exact_tokens, s_ntok);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// document-sharded exchange
// ---------------------------------------------------------------------------------------------
// cut[b] = the n_sel-th largest key over the G shards' lists (1 if fewer exist: keep everything).  Rank g's list starts
// rs u64 words after rank g-1's (rs = B * n_sel for contiguous [G][B][n_sel] lists; np_dist.hip appends a status trailer to
// every rank's record: status_off >= 0 is its word offset inside a record, and a non-zero status of ANY rank makes the
// cut ~0 -- no local candidate survives, the batch comes back empty on every rank instead of silently missing a shard).
__global__ void __launch_bounds__(1024) select_cut_kernel(const uint64_t* __restrict__ all_keys, int64_t rs,
                                                          int64_t status_off, int G, int B, int n_sel, int NP2,
                                                          uint64_t* __restrict__ cut) {
  extern __shared__ uint64_t s_sel[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = G * n_sel;
  for (int i = tid; i < NP2; i += 1024) {
    uint64_t v = 0;
    if (i < n) {
      const int g = i / n_sel, j = i - g * n_sel;
      v = all_keys[(int64_t)g * rs + (int64_t)b * n_sel + j];
    }
    s_sel[i] = v;
  }
  bitonic_sort_desc(s_sel, NP2, tid, 1024);
  if (tid == 0) {
    const uint64_t c = (n_sel > 0) ? s_sel[n_sel - 1] : 0ull;
    uint64_t failed = 0;
    if (status_off >= 0)
      for (int g = 0; g < G; ++g) failed |= all_keys[(int64_t)g * rs + status_off];
    cut[b] = failed ? ~0ull : (c ? c : 1ull);
  }
}

// one u64 status word of a rank's exchange record: 0 = healthy, else np_status | (rank + 1) << 32
__global__ void set_status_kernel(uint64_t* __restrict__ word, uint64_t value) { *word = value; }

// out[w] = OR over the G gathered bitmaps (eligible centroids of a subset, search.rs:350-364, across document shards)
__global__ void or_reduce_kernel(const uint32_t* __restrict__ all, int G, int64_t words, uint32_t* __restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  uint32_t v = 0;
  for (int g = 0; g < G; ++g) v |= all[(int64_t)g * words + w];
  out[w] = v;
}

// merge G x top_k triples by (exact desc [finite first], approx key desc); rank by counting.  Rank g's arrays start
// rs_* elements after rank g-1's (contiguous [G][B][top_k] arrays, or one packed record per rank).  status (nullable):
// one u64 per rank, rs_status words apart; a non-zero word of any rank ABANDONS every query of the batch -- out_counts[b] = -1,
// a count no healthy batch produces, so a host that never polls np_hip_comm_status cannot mistake the batch for "no results" --
// and the first such word is left in *host_status (pinned host memory the caller reads after synchronising the stream).
__global__ void __launch_bounds__(256) merge_topk_kernel(const int64_t* __restrict__ ids, const float* __restrict__ scores,
                                                         const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ counts, int64_t rs_ids,
                                                         int64_t rs_scores, int64_t rs_keys, int64_t rs_counts, int G,
                                                         int B, int top_k, int64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores, int32_t* __restrict__ out_counts,
                                                         const uint64_t* __restrict__ status = nullptr, int64_t rs_status = 0,
                                                         uint64_t* __restrict__ host_status = nullptr) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (status) {
    uint64_t failed = 0;
    for (int g = 0; g < G && !failed; ++g) failed = status[(int64_t)g * rs_status];
    if (failed) {   // wave-uniform: every thread read the same words
      if (tid == 0) {
        out_counts[b] = -1;
        if (b == 0 && host_status) *host_status = failed;
      }
      return;
    }
  }
  const int n = G * top_k;
  int total = 0;
  for (int g = 0; g < G; ++g) total += counts[g * rs_counts + b];
  for (int i = tid; i < n; i += 256) {
    const int g = i / top_k, j = i - g * top_k;
    if (j >= counts[g * rs_counts + b]) continue;
    const int64_t e = (int64_t)b * top_k + j;
    const uint32_t ks = okey(scores[g * rs_scores + e]);
    const uint64_t ka = keys[g * rs_keys + e];
    int rank = 0;
    for (int g2 = 0; g2 < G; ++g2) {
      const int c2 = counts[g2 * rs_counts + b];
      for (int j2 = 0; j2 < c2; ++j2) {
        const int64_t e2 = (int64_t)b * top_k + j2;
        const uint32_t ks2 = okey(scores[g2 * rs_scores + e2]);
        if (ks2 > ks || (ks2 == ks && keys[g2 * rs_keys + e2] > ka)) ++rank;
      }
    }
    if (rank < top_k) {
      out_ids[(int64_t)b * top_k + rank] = ids[g * rs_ids + e];
      out_scores[(int64_t)b * top_k + rank] = scores[g * rs_scores + e];
    }
  }
  if (tid == 0) out_counts[b] = min(top_k, total);
}

// ---------------------------------------------------------------------------------------------
// N2  decompress_documents (index.rs:1159-1245, codec.rs:423-470): one wave per token, any dim.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) decompress_kernel(const int64_t* __restrict__ tok_src /*[n] shard token idx*/,
                                                         const int64_t* __restrict__ out_base /*[n] first output row of the token's document*/,
                                                         const uint16_t* __restrict__ tok_pos /* stored -> original position, or NULL */,
                                                         int64_t n, int dim /* stored row width */,
                                                         int ldim /* dims of the index files = output row width */,
                                                         int nbits, int pd,
                                                         const float* __restrict__ centroids,
                                                         const float* __restrict__ wlut,
                                                         CodeArr codes,
                                                         const uint8_t* __restrict__ residuals,
                                                         float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t tok = tok_src[i];
  // the index keeps a document's tokens in code order: row = the token's position in the on-disk order
  const int64_t row = tok_pos ? out_base[i] + (int64_t)tok_pos[tok] : i;
  const uint32_t code = codes[tok];
  const int per = 8 / nbits;
  const uint32_t mask = (1u << nbits) - 1u;
  float ss = 0.f;
  for (int j = lane; j < ldim; j += 64) {
    const uint32_t byte = residuals[tok * pd + j / per];
    const int e = j % per;
    const float x = centroids[(int64_t)code * dim + j] + wlut[(byte >> (8 - nbits * (e + 1))) & mask];
    out[row * ldim + j] = x;
    ss += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float norm = fmaxf(sqrtf(ss), 1e-12f);
  for (int j = lane; j < ldim; j += 64) out[row * ldim + j] = out[row * ldim + j] / norm;
}

// ---------------------------------------------------------------------------------------------
// N3  index-time encode: nearest centroid + residual quantisation (codec.rs:297-411,
// index.rs:17-40,289-371; the reference's only CUDA code, cuda.rs:185-237).  Scores come from the same
// exact-f32 MFMA GEMM as S1 (tokens fed as 32-token "queries"), so the argmax sees bit-identical values.
// encode_argmax: one wave per token.  Iterator::max_by(cmp_f32_for_max) keeps the LAST of equal maxima;
// okey() is cmp_f32_for_max as an integer (non-finite = 0 = below every finite value, all equal).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) encode_argmax_kernel(const float* __restrict__ QCT,
                                                            const uint32_t* __restrict__ gmax, int64_t K, int64_t KP,
                                                            int LQP, int64_t n_tokens, int64_t* __restrict__ codes) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tokens) return;
  const int64_t b = t / LQP;
  const int q = (int)(t - b * LQP);
  const int64_t G = KP >> 5;
  const uint32_t* gm = gmax + b * G * LQP + q;
  uint32_t mk = 0;
  for (int64_t g = lane; g < G; g += 64) mk = max(mk, gm[g * LQP]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, o));
  int64_t code = K - 1;   // every score non-finite: all equal, the last one wins
  if (mk != 0u) {
    int gl = -1;
    for (int64_t g = lane; g < G; g += 64)
      if (gm[g * LQP] == mk) gl = (int)g;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gl = max(gl, __shfl_xor(gl, o));
    const int64_t c = (int64_t)gl * 32 + (lane & 31);
    const bool hit = lane < 32 && c < K && okey(QCT[(b * KP + c) * LQP + q]) == mk;
    const unsigned long long bal = __ballot(hit);
    code = (int64_t)gl * 32 + (63 - __clzll(bal));
  }
  if (lane == 0) codes[t] = code;
}

// one thread per packed byte: residual = x - centroid[code] (f32), bucket = #cutoffs strictly below it,
// bucket bits LSB-first into the byte MSB-first (codec.rs:356-411)
__global__ void __launch_bounds__(256) encode_pack_kernel(const float* __restrict__ x, const float* __restrict__ C,
                                                          const int64_t* __restrict__ codes,
                                                          const float* __restrict__ cutoffs, int64_t n_tokens, int dim,
                                                          int cdim /* row width of C */, int nbits,
                                                          uint8_t* __restrict__ packed) {
  const int pd = dim * nbits / 8, per = 8 / nbits, ncut = (1 << nbits) - 1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_tokens * pd) return;
  const int64_t t = i / pd;
  const int jb = (int)(i - t * pd);
  const float* xr = x + t * dim + jb * per;
  const float* cr = C + codes[t] * cdim + jb * per;
  uint32_t byte = 0;
  for (int e = 0; e < per; ++e) {
    const float v = __fsub_rn(xr[e], cr[e]);
    int bucket = 0;
    for (int c = 0; c < ncut; ++c) bucket += (v > cutoffs[c]) ? 1 : 0;
    for (int bb = 0; bb < nbits; ++bb) byte |= (uint32_t)((bucket >> bb) & 1) << (7 - (e * nbits + bb));
  }
  packed[i] = (uint8_t)byte;
}

// ---------------------------------------------------------------------------------------------
// N4  /rerank MaxSim on caller-supplied embeddings (next-plaid-api handlers/rerank.rs:57-94):
// sim = sequential sum of q*d products (multiply, then add: no FMA), max over document tokens, sum over query
// tokens in order; any non-finite sim or running total is the handler's BadRequest (flag).  One block per
// document; 8 query rows at a time sit in LDS, a thread walks one document token against them.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rerank_kernel(const float* __restrict__ Q, int lq, int dim,
                                                     const float* __restrict__ D, const int64_t* __restrict__ doc_off,
                                                     float* __restrict__ scores, int* __restrict__ flags) {
  // the handler multiplies, then adds.  hipcc contracts a*b+c into an FMA by default -- also through __fmul_rn /
  // __fadd_rn, which are plain operators compiled with contraction allowed -- so it is switched off for this body
#pragma clang fp contract(off)
  extern __shared__ float sq[];   // [8][dim]
  __shared__ float s_wmax[4][8];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t doc = blockIdx.x, t0 = doc_off[doc], t1 = doc_off[doc + 1];
  if (tid == 0) s_bad = 0;
  float total = 0.0f;
  int bad0 = 0;
  for (int q0 = 0; q0 < lq; q0 += 8) {
    const int nq = min(8, lq - q0);
    __syncthreads();
    for (int i = tid; i < 8 * dim; i += 256) sq[i] = (i < nq * dim) ? Q[(int64_t)q0 * dim + i] : 0.0f;
    __syncthreads();
    float mx[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) mx[r] = NP_NEG_INF;
    int bad = 0;
    for (int64_t t = t0 + tid; t < t1; t += 256) {
      float sim[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) sim[r] = 0.0f;
      const float* dr = D + t * dim;
      for (int k = 0; k < dim; ++k) {
        const float dv = dr[k];
#pragma unroll
        for (int r = 0; r < 8; ++r) sim[r] = sim[r] + sq[r * dim + k] * dv;   // not contracted (pragma above)
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < nq) {
          if (!finitef(sim[r])) bad = 1;
          if (sim[r] > mx[r]) mx[r] = sim[r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o));
      if (lane == 0) s_wmax[wave][r] = mx[r];
    }
    if (bad) atomicOr(&s_bad, 1);
    __syncthreads();
    if (tid == 0) {
      for (int r = 0; r < nq; ++r) {
        const float m = fmaxf(fmaxf(s_wmax[0][r], s_wmax[1][r]), fmaxf(s_wmax[2][r], s_wmax[3][r]));
        if (m > NP_NEG_INF) {
          total = total + m;
          if (!finitef(total)) bad0 = 1;
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    scores[doc] = total;
    flags[doc] = (bad0 || s_bad) ? 1 : 0;
  }
}

}  // namespace np
