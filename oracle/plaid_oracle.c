/*
 * plaid_oracle.c -- CPU ORACLE for the next-plaid PLAID search path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (next-plaid_amd/csrc, libnextplaid_hip.so) never links, imports or calls it.
 *
 * What it is: a plain-C restatement of the reference's CPU search algorithm
 * (lightonai/next-plaid v1.6.1, Rust), function by function, each citing the
 * reference file:line it follows (paths relative to /root/reference/next-plaid/src).
 *
 * Parity pinning status
 *   - The reference cannot be compiled here (no cargo/rustc, no network), so there is
 *     no oracle/_ref binary.  The restatement is pinned against every known-answer test
 *     the reference holds for this path (tests/test_oracle_known_answers.py):
 *     maxsim 1.7 (search.rs:684-705, maxsim.rs:392-413), NaN-row 8.0 (maxsim.rs:497-507),
 *     simd_max cases (maxsim.rs:415-441), comparator rules (search.rs:717-742),
 *     codec round-trip / packed width (codec.rs:665-730), packbits MSB-first
 *     (utils.rs:296-304), rerank 2.0/1.0/0.0 (next-plaid-api/tests/integration_tests.rs:2301-2376).
 *   - The reference's tests pin NO rankings/scores for search() end to end (its fixtures
 *     are unseeded random): AT THE search() BOUNDARY PARITY IS UNPINNED by the reference.
 *     There it is pinned by cross-checking this file against the independent numpy
 *     restatement oracle/plaid_numpy.py on seeded indices (tests/golden/).
 *
 * Third-party arithmetic not under /root/reference (named + restated):
 *   - ndarray 0.16.1 `.dot()` on 2-D x 2-D -> matrixmultiply 0.3.10 sgemm: every output
 *     element is a k-ordered FMA chain (AVX2+FMA micro-kernel, kc >= 128 so one k-block).
 *     Restated as po_dot_fma() (sequential fmaf over k).  Call sites: search.rs:174,345,
 *     maxsim.rs:281.
 *   - ndarray 0.16.1 1-D dot / mat-vec (no BLAS feature): numeric_util::unrolled_dot,
 *     8 partial sums, no FMA.  Restated as po_unrolled_dot().  Call sites: search.rs:268,
 *     maxsim.rs:304, codec.rs:465 (row.dot(&row)).
 *   Summation order inside these is implementation-defined in the reference (BLAS builds
 *   differ), so parity vs the *reference* is tolerance-based; parity of the HIP path vs
 *   *this oracle* is bit-exact for all integer/index stages (cells, candidates, approx
 *   scores, selection) because both use the k-ordered FMA chain for Q.C^T.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* comparators: search.rs:110-133, maxsim.rs:22-35, codec.rs:22-30            */
/* ------------------------------------------------------------------------- */

static inline int po_is_finite(float x) { return isfinite(x); }

/* f32::total_cmp (Rust std): compare bit patterns after the sign-magnitude fixup. */
static inline int po_total_cmp(float a, float b) {
  int32_t ia, ib;
  memcpy(&ia, &a, 4);
  memcpy(&ib, &b, 4);
  ia ^= (int32_t)(((uint32_t)(ia >> 31)) >> 1);
  ib ^= (int32_t)(((uint32_t)(ib >> 31)) >> 1);
  return (ia > ib) - (ia < ib);
}

/* search.rs:110-117 cmp_score_ascending */
PO_API int po_cmp_score_ascending(float a, float b) {
  int fa = po_is_finite(a), fb = po_is_finite(b);
  if (fa && fb) return po_total_cmp(a, b);
  if (fa && !fb) return 1;
  if (!fa && fb) return -1;
  return 0;
}
/* search.rs:119-121 */
static inline int po_cmp_score_descending(float a, float b) { return po_cmp_score_ascending(b, a); }
/* search.rs:123-125, maxsim.rs:32-35 */
PO_API int po_is_score_better(float candidate, float current) {
  return po_cmp_score_ascending(candidate, current) > 0;
}
/* search.rs:127-133 */
PO_API float po_max_score(float a, float b) { return po_is_score_better(b, a) ? b : a; }

/* ------------------------------------------------------------------------- */
/* third-party dot products (see header)                                      */
/* ------------------------------------------------------------------------- */

/* matrixmultiply sgemm element: k-ordered FMA chain. */
static inline float po_dot_fma(const float* a, const float* b, int64_t d) {
  float acc = 0.0f;
  for (int64_t k = 0; k < d; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
  return acc;
}

/* ndarray numeric_util::unrolled_dot (8 partial sums, mul then add, no FMA). */
PO_API float po_unrolled_dot(const float* xs, const float* ys, int64_t n) {
  float sum = 0.0f;
  float p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    p0 = p0 + xs[i + 0] * ys[i + 0];
    p1 = p1 + xs[i + 1] * ys[i + 1];
    p2 = p2 + xs[i + 2] * ys[i + 2];
    p3 = p3 + xs[i + 3] * ys[i + 3];
    p4 = p4 + xs[i + 4] * ys[i + 4];
    p5 = p5 + xs[i + 5] * ys[i + 5];
    p6 = p6 + xs[i + 6] * ys[i + 6];
    p7 = p7 + xs[i + 7] * ys[i + 7];
  }
  sum = sum + (p0 + p4);
  sum = sum + (p1 + p5);
  sum = sum + (p2 + p6);
  sum = sum + (p3 + p7);
  for (; i < n; ++i) sum = sum + xs[i] * ys[i];
  return sum;
}

/* ------------------------------------------------------------------------- */
/* codec: LUTs (codec.rs:168-214), decompress (codec.rs:423-470),             */
/*        quantize_residuals (codec.rs:356-411), packbits (utils.rs:190-201)  */
/* ------------------------------------------------------------------------- */

/* codec.rs:168-196: reverse the bits inside each nbits-wide segment of a byte. */
PO_API void po_byte_reversed_bits_map(int nbits, uint8_t out[256]) {
  uint32_t nbits_mask = (1u << nbits) - 1u;
  for (int i = 0; i < 256; ++i) {
    uint32_t val = (uint32_t)i, o = 0;
    int pos = 8;
    while (pos >= nbits) {
      uint32_t segment = (val >> (uint32_t)(pos - nbits)) & nbits_mask;
      uint32_t rev = 0;
      for (int k = 0; k < nbits; ++k)
        if (segment & (1u << k)) rev |= 1u << (nbits - 1 - k);
      o |= rev;
      if (pos > nbits) o <<= nbits;
      pos -= nbits;
    }
    out[i] = (uint8_t)o;
  }
}

/* codec.rs:198-214: table[byte][j] = bucket index of the j-th dim packed in byte
 * (column 0 = highest segment).  keys_per_byte = 8/nbits. */
PO_API void po_bucket_weight_indices_lookup(int nbits, int32_t* table /*256*kpb*/) {
  int kpb = 8 / nbits;
  uint32_t mask = (1u << nbits) - 1u;
  for (int b = 0; b < 256; ++b)
    for (int k = kpb - 1; k >= 0; --k) {
      uint32_t idx = ((uint32_t)b >> (k * nbits)) & mask;
      table[b * kpb + (kpb - 1 - k)] = (int32_t)idx;
    }
}

/* utils.rs:190-201 packbits (big-endian / MSB-first). out has ceil(n/8) bytes. */
PO_API void po_packbits(const uint8_t* bits, int64_t n, uint8_t* out) {
  for (int64_t c = 0; c * 8 < n; ++c) {
    uint8_t byte = 0;
    for (int i = 0; i < 8 && c * 8 + i < n; ++i) byte |= (uint8_t)(bits[c * 8 + i] << (7 - i));
    out[c] = byte;
  }
}

/* codec.rs:356-411 quantize_residuals: bucket = #cutoffs strictly below val; bucket bits
 * emitted LSB-first, written MSB-first into bytes. packed: [n, dim*nbits/8]. */
PO_API void po_quantize_residuals(const float* residuals, int64_t n, int64_t dim, int nbits,
                                  const float* cutoffs, int64_t n_cutoffs, uint8_t* packed) {
  int64_t pd = dim * nbits / 8;
  memset(packed, 0, (size_t)(n * pd));
  for (int64_t r = 0; r < n; ++r) {
    int64_t bit_idx = 0;
    uint8_t* row = packed + r * pd;
    for (int64_t j = 0; j < dim; ++j) {
      float val = residuals[r * dim + j];
      int bucket = 0;
      for (int64_t c = 0; c < n_cutoffs; ++c) bucket += (val > cutoffs[c]);
      for (int b = 0; b < nbits; ++b) {
        uint8_t bit = (uint8_t)((bucket >> b) & 1);
        int64_t byte_idx = bit_idx / 8;
        int bit_pos = 7 - (int)(bit_idx % 8);
        row[byte_idx] |= (uint8_t)(bit << bit_pos);
        ++bit_idx;
      }
    }
  }
}

/* codec.rs:22-30 cmp_f32_for_max as an integer key: finite values keep total_cmp order, every
 * non-finite value sits below them and all non-finite values are equal. */
static uint32_t po_key_for_max(float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  if ((b & 0x7F800000u) == 0x7F800000u) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* codec.rs:297-345 compress_into_codes_cpu (N3 row): scores = batch.dot(centroids.t())
 * (matrixmultiply: k-ordered fused chain, po_dot_fma), then per row
 * enumerate().max_by(cmp_f32_for_max) -- Iterator::max_by keeps the LAST of equal maxima. */
PO_API void po_compress_into_codes(const float* X, int64_t n, const float* C, int64_t K, int64_t d,
                                   int64_t* codes) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < n; ++t) {
    int64_t best = 0;
    uint32_t bk = 0;
    for (int64_t c = 0; c < K; ++c) {
      const uint32_t k = po_key_for_max(po_dot_fma(X + t * d, C + c * d, d));
      if (c == 0 || k >= bk) { best = c; bk = k; }
    }
    codes[t] = best;
  }
}

/* index.rs:17-40 compress_and_residuals_cpu + index.rs:352 quantize_residuals (encode_index_chunk):
 * codes, residual = embedding - centroid[code] (f32), packed buckets. */
PO_API void po_encode_tokens(const float* X, int64_t n, const float* C, int64_t K, int64_t d, int nbits,
                             const float* cutoffs, int64_t n_cutoffs, int64_t* codes, uint8_t* packed) {
  po_compress_into_codes(X, n, C, K, d, codes);
  float* res = (float*)malloc((size_t)(n > 0 ? n : 1) * (size_t)d * sizeof(float));
  for (int64_t t = 0; t < n; ++t)
    for (int64_t j = 0; j < d; ++j) res[t * d + j] = X[t * d + j] - C[codes[t] * d + j];
  po_quantize_residuals(res, n, d, nbits, cutoffs, n_cutoffs, packed);
  free(res);
}

/* codec.rs:423-470 decompress: out[i,j] = centroid[codes[i]][j] + weights[bucket(i,j)];
 * then each row /= max(sqrt(row.row), 1e-12).  codes are i64 (as usize in the reference). */
PO_API void po_decompress(const uint8_t* packed, const int64_t* codes, int64_t n, int64_t dim,
                          int nbits, const float* centroids, const float* bucket_weights,
                          float* out) {
  uint8_t rev[256];
  int kpb = 8 / nbits;
  int32_t lookup[256 * 8];
  po_byte_reversed_bits_map(nbits, rev);
  po_bucket_weight_indices_lookup(nbits, lookup);
  int64_t pd = dim * nbits / 8;
  for (int64_t i = 0; i < n; ++i) {
    const float* centroid = centroids + codes[i] * dim;
    float* o = out + i * dim;
    int64_t residual_idx = 0;
    for (int64_t b = 0; b < pd; ++b) {
      uint8_t reversed = rev[packed[i * pd + b]];
      const int32_t* indices = lookup + (int)reversed * kpb;
      for (int k = 0; k < kpb; ++k) {
        if (residual_idx < dim) {
          o[residual_idx] = centroid[residual_idx] + bucket_weights[indices[k]];
          ++residual_idx;
        }
      }
    }
    for (; residual_idx < dim; ++residual_idx) o[residual_idx] = 0.0f; /* Array2::zeros init */
    /* codec.rs:464-467 */
    float nn = po_unrolled_dot(o, o, dim);
    float norm = sqrtf(nn);
    if (!(norm > 1e-12f)) norm = 1e-12f; /* f32::max(1e-12): NaN norm -> 1e-12 */
    for (int64_t j = 0; j < dim; ++j) o[j] = o[j] / norm;
  }
}

/* ------------------------------------------------------------------------- */
/* MaxSim: maxsim.rs:49-63 scalar_max, :79-149 simd_max, :270-315              */
/* ------------------------------------------------------------------------- */

/* maxsim.rs:57-63: slice.iter().copied().max_by(cmp_f32_for_max).unwrap_or(-inf)
 * Iterator::max_by returns the LAST of several equally-maximum elements. */
static float po_scalar_max(const float* s, int64_t n) {
  if (n == 0) return -INFINITY;
  float best = s[0];
  for (int64_t i = 1; i < n; ++i)
    if (po_cmp_score_ascending(s[i], best) >= 0) best = s[i];
  return best;
}

/* maxsim.rs:79-149 (AVX2 form).  The vector body computes a plain max; whenever the result
 * or any input is non-finite the function returns scalar_max(slice) (maxsim.rs:143-145), and
 * for all-finite input plain max == scalar_max.  So simd_max(slice) == scalar_max(slice). */
PO_API float po_simd_max(const float* s, int64_t n) { return po_scalar_max(s, n); }

/* maxsim.rs:298-315 maxsim_score_simple */
static float po_maxsim_simple(const float* Q, int64_t Lq, const float* D, int64_t n, int64_t d) {
  float total = 0.0f;
  for (int64_t q = 0; q < Lq; ++q) {
    float max_sim = -INFINITY;
    for (int64_t t = 0; t < n; ++t) {
      float sim = po_unrolled_dot(Q + q * d, D + t * d, d);
      if (po_is_score_better(sim, max_sim)) max_sim = sim;
    }
    if (po_is_finite(max_sim)) total += max_sim;
  }
  return total;
}

/* maxsim.rs:270-294 maxsim_score.  scratch: row >= n floats, Dt >= n*d floats.
 * scores = Q.D^T (maxsim.rs:281) is evaluated with D transposed so the loop vectorises across
 * document tokens; every element is still its own k-ordered FMA chain (bitwise == po_dot_fma). */
static float po_maxsim_score_scratch(const float* Q, int64_t Lq, const float* D, int64_t n,
                                     int64_t d, float* row, float* Dt) {
  if (Lq * n < 256) return po_maxsim_simple(Q, Lq, D, n, d);              /* :274-277 */
  for (int64_t t = 0; t < n; ++t)
    for (int64_t k = 0; k < d; ++k) Dt[k * n + t] = D[t * d + k];
  float total = 0.0f;
  for (int64_t q = 0; q < Lq; ++q) {
    for (int64_t t = 0; t < n; ++t) row[t] = 0.0f;
    for (int64_t k = 0; k < d; ++k) {
      const float qv = Q[q * d + k];
      const float* dk = Dt + k * n;
#pragma omp simd
      for (int64_t t = 0; t < n; ++t) row[t] = __builtin_fmaf(qv, dk[t], row[t]);
    }
    float max_sim = po_simd_max(row, n);                                          /* :287 */
    if (po_is_finite(max_sim)) total += max_sim;                                  /* :288-290 */
  }
  return total;
}

PO_API float po_maxsim_score(const float* Q, int64_t Lq, const float* D, int64_t n, int64_t d) {
  float* row = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float* Dt = (float*)malloc(sizeof(float) * (size_t)(n * d > 0 ? n * d : 1));
  float s = po_maxsim_score_scratch(Q, Lq, D, n, d, row, Dt);
  free(row); free(Dt);
  return s;
}

/* next-plaid-api/src/handlers/rerank.rs:57-94 compute_maxsim (N4 row).  Returns 0 on
 * success, 1 if a non-finite similarity/score is hit (the handler's BadRequest). */
PO_API int po_rerank_maxsim(const float* Q, int64_t Lq, const float* D, int64_t n, int64_t d,
                            float* out) {
  float total = 0.0f;
  for (int64_t q = 0; q < Lq; ++q) {
    float max_sim = -INFINITY;
    for (int64_t t = 0; t < n; ++t) {
      float sim = 0.0f; /* iter().zip().map(q*d).sum(): sequential, mul then add */
      for (int64_t k = 0; k < d; ++k) sim = sim + Q[q * d + k] * D[t * d + k];
      if (!po_is_finite(sim)) return 1;
      if (sim > max_sim) max_sim = sim;
    }
    if (max_sim > -INFINITY) {
      total += max_sim;
      if (!po_is_finite(total)) return 1;
    }
  }
  *out = total;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* index view (index.rs:995-1016) -- borrows caller memory, owns derived bits  */
/* ------------------------------------------------------------------------- */

typedef struct po_index {
  int64_t K, d, N;
  int32_t nbits;
  int64_t pd;
  const float* centroids;       /* [K,d] centroids.npy */
  const float* bucket_weights;  /* [2^nbits] */
  const int64_t* ivf;           /* ivf.npy */
  const int32_t* ivf_lengths;   /* [K] */
  int64_t* ivf_offsets;         /* [K+1] owned; index.rs:1089-1094 */
  const int64_t* doc_lengths;   /* [N] */
  int64_t* doc_offsets;         /* [N+1] owned; index.rs:1106-1110 */
  const int64_t* codes;         /* merged codes (i64) */
  const uint8_t* residuals;     /* merged residuals [T(+pad), pd] */
  float* centroids_t;           /* [d,K] owned transposed copy (speed only; same values) */
} po_index;

PO_API po_index* po_index_create(int64_t K, int64_t d, int64_t N, int32_t nbits,
                                 const float* centroids, const float* bucket_weights,
                                 const int64_t* ivf, const int32_t* ivf_lengths,
                                 const int64_t* doc_lengths, const int64_t* codes,
                                 const uint8_t* residuals) {
  po_index* ix = (po_index*)calloc(1, sizeof(po_index));
  ix->K = K; ix->d = d; ix->N = N; ix->nbits = nbits; ix->pd = d * nbits / 8;
  ix->centroids = centroids; ix->bucket_weights = bucket_weights;
  ix->ivf = ivf; ix->ivf_lengths = ivf_lengths; ix->doc_lengths = doc_lengths;
  ix->codes = codes; ix->residuals = residuals;
  ix->ivf_offsets = (int64_t*)malloc(sizeof(int64_t) * (size_t)(K + 1));
  ix->ivf_offsets[0] = 0;
  for (int64_t i = 0; i < K; ++i) ix->ivf_offsets[i + 1] = ix->ivf_offsets[i] + ivf_lengths[i];
  ix->doc_offsets = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N + 1));
  ix->doc_offsets[0] = 0;
  for (int64_t i = 0; i < N; ++i) ix->doc_offsets[i + 1] = ix->doc_offsets[i] + doc_lengths[i];
  ix->centroids_t = (float*)malloc(sizeof(float) * (size_t)(K * d > 0 ? K * d : 1));
  for (int64_t c = 0; c < K; ++c)
    for (int64_t k = 0; k < d; ++k) ix->centroids_t[k * K + c] = centroids[c * d + k];
  return ix;
}

PO_API void po_index_destroy(po_index* ix) {
  if (!ix) return;
  free(ix->ivf_offsets); free(ix->doc_offsets); free(ix->centroids_t); free(ix);
}

/* SearchParameters, search.rs:26-69 (batch_size is unused by search). */
typedef struct po_params {
  int32_t top_k, n_full_scores, n_ivf_probe, centroid_batch_size;
  float centroid_score_threshold;
  int32_t has_threshold;
} po_params;

/* Optional per-stage trace (caller allocates; capacities noted). */
typedef struct po_trace {
  int64_t n_cells;  int64_t* cells;      /* cap K, ascending */
  int64_t n_cand;   int64_t* cand;       /* cap N, ascending doc ids (after subset retain) */
  float* approx;                         /* cap N, aligned with cand */
  int64_t n_sel;    int64_t* sel;        /* cap n_full_scores: docs exact-scored, approx-rank order */
  float* sel_exact;                      /* cap n_full_scores */
  int64_t n_ivf_ids;                     /* sum of probed posting-list lengths */
  int64_t n_cand_tokens;                 /* sum doclen over candidates */
  int64_t n_exact_tokens;                /* sum doclen over exact-scored docs */
  int32_t used_batched;
} po_trace;

/* Q.C^T for one query: out[q*K + c], k-ordered FMA chain per element (search.rs:345).
 * Vectorised across c using the transposed copy; each lane is still its own k-ordered chain. */
static void po_query_centroid_scores(const po_index* ix, const float* Q, int64_t Lq, float* out) {
  const int64_t K = ix->K, d = ix->d;
  for (int64_t q = 0; q < Lq; ++q) {
    float* o = out + q * K;
    for (int64_t c = 0; c < K; ++c) o[c] = 0.0f;
    for (int64_t k = 0; k < d; ++k) {
      const float qv = Q[q * d + k];
      const float* ct = ix->centroids_t + k * K;
#pragma omp simd
      for (int64_t c = 0; c < K; ++c) o[c] = __builtin_fmaf(qv, ct[c], o[c]);
    }
  }
}

typedef struct { float score; int64_t id; } po_pair;

/* Top-n of (score,id) pairs by cmp_score_descending; ties at the cut are unspecified in the
 * reference (select_nth_unstable_by, search.rs:405-409); here: lower id first. */
static int po_pair_desc_cmp(const void* a, const void* b) {
  const po_pair* x = (const po_pair*)a; const po_pair* y = (const po_pair*)b;
  int c = po_cmp_score_descending(x->score, y->score);
  if (c) return c;
  return (x->id > y->id) - (x->id < y->id);
}

/* O(m) average partial selection standing in for select_nth_unstable_by (search.rs:405-409):
 * afterwards a[0..n) are the n best under po_pair_desc_cmp (a strict total order: ids unique). */
static void po_select_top(po_pair* a, int64_t m, int64_t n) {
  if (n <= 0 || n >= m) return;
  int64_t lo = 0, hi = m - 1, kth = n - 1;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    po_pair pivot = a[mid];
    int64_t i = lo, j = hi;
    while (i <= j) {
      while (po_pair_desc_cmp(&a[i], &pivot) < 0) ++i;
      while (po_pair_desc_cmp(&a[j], &pivot) > 0) --j;
      if (i <= j) { po_pair t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
    }
    if (kth <= j) hi = j; else if (kth >= i) lo = i; else break;
  }
}

/* rayon's nested par_iter never oversubscribes; with OpenMP the inner regions get the
 * threads the outer (per-query) level leaves over. */
static int po_inner_threads = 0; /* 0 = all */
static int po_inner_nt(void) {
#ifdef _OPENMP
  if (po_inner_threads > 0) return po_inner_threads;
  return omp_in_parallel() ? 1 : omp_get_max_threads();
#else
  return 1;
#endif
}

static int po_i64_cmp(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* stable sort of indices by score descending (slice::sort_by is stable; search.rs:460,496) */
typedef struct { float score; int64_t pos; int64_t id; } po_rank;
static int po_rank_cmp(const void* a, const void* b) {
  const po_rank* x = (const po_rank*)a; const po_rank* y = (const po_rank*)b;
  int c = po_cmp_score_descending(x->score, y->score);
  if (c) return c;
  return (x->pos > y->pos) - (x->pos < y->pos);
}

/* index.rs:1142-1156 get_candidates: concat posting lists of cells < K, sort, dedup. */
static int64_t po_get_candidates(const po_index* ix, const int64_t* cells, int64_t n_cells,
                                 int64_t** out, int64_t* n_ivf_ids) {
  int64_t total = 0;
  for (int64_t i = 0; i < n_cells; ++i)
    if (cells[i] >= 0 && cells[i] < ix->K) total += ix->ivf_lengths[cells[i]];
  int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
  int64_t n = 0;
  for (int64_t i = 0; i < n_cells; ++i) {
    if (cells[i] < 0 || cells[i] >= ix->K) continue;
    int64_t s = ix->ivf_offsets[cells[i]], l = ix->ivf_lengths[cells[i]];
    memcpy(c + n, ix->ivf + s, sizeof(int64_t) * (size_t)l);
    n += l;
  }
  if (n_ivf_ids) *n_ivf_ids = n;
  qsort(c, (size_t)n, sizeof(int64_t), po_i64_cmp);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (m == 0 || c[m - 1] != c[i]) c[m++] = c[i];
  *out = c;
  return m;
}

/* search.rs:434-437 / 542-545: candidates.retain(|c| subset_set.contains(c)) */
static int64_t po_retain_subset(int64_t* cand, int64_t n, const int64_t* subset, int64_t sl) {
  int64_t* s = (int64_t*)malloc(sizeof(int64_t) * (size_t)(sl > 0 ? sl : 1));
  memcpy(s, subset, sizeof(int64_t) * (size_t)sl);
  qsort(s, (size_t)sl, sizeof(int64_t), po_i64_cmp);
  int64_t m = 0, j = 0;
  for (int64_t i = 0; i < n; ++i) {
    while (j < sl && s[j] < cand[i]) ++j;
    if (j < sl && s[j] == cand[i]) cand[m++] = cand[i];
  }
  free(s);
  return m;
}

/* search.rs:305-324 approximate_score_mmap, with QC stored [Lq][K]. */
static float po_approx_score_dense(const float* qc, int64_t Lq, int64_t K, const int64_t* codes,
                                   int64_t n) {
  float score = 0.0f;
  for (int64_t q = 0; q < Lq; ++q) {
    float max_score = -INFINITY;
    const float* row = qc + q * K;
    for (int64_t t = 0; t < n; ++t) {
      float cs = row[codes[t]];
      if (cs > max_score) max_score = cs;
    }
    if (max_score > -INFINITY) score += max_score;
  }
  return score;
}

/* Exact stage shared by both paths: search.rs:459-515 / 583-639. */
static int po_rank_and_rescore(const po_index* ix, const float* Q, int64_t Lq, const po_params* p,
                               const int64_t* cand, const float* approx, int64_t n_cand,
                               int64_t* out_ids, float* out_scores, int32_t* out_count,
                               po_trace* tr) {
  const int64_t d = ix->d;
  po_rank* r = (po_rank*)malloc(sizeof(po_rank) * (size_t)n_cand);
  for (int64_t i = 0; i < n_cand; ++i) { r[i].score = approx[i]; r[i].pos = i; r[i].id = cand[i]; }
  qsort(r, (size_t)n_cand, sizeof(po_rank), po_rank_cmp);          /* search.rs:460 */
  int64_t n_top = n_cand < p->n_full_scores ? n_cand : p->n_full_scores;   /* :461-465 */
  int64_t n_dec = p->n_full_scores / 4;                                   /* :468 */
  if (n_dec < p->top_k) n_dec = p->top_k;
  if (n_dec > n_top) n_dec = n_top;                                       /* :469 take() */
  if (tr) { tr->n_sel = n_dec; tr->n_exact_tokens = 0; }
  if (n_dec == 0) { free(r); *out_count = 0; return 0; }                  /* :471-477 */

  po_rank* e = (po_rank*)malloc(sizeof(po_rank) * (size_t)n_dec);
  int64_t max_len = 1;
  for (int64_t i = 0; i < n_dec; ++i) {
    int64_t l = ix->doc_lengths[r[i].id];
    if (l > max_len) max_len = l;
  }
  /* search.rs:481-493: par_chunks(128) over docs; get_document_embeddings + colbert_score */
#pragma omp parallel num_threads(po_inner_nt())
  {
    float* D = (float*)malloc(sizeof(float) * (size_t)(max_len * d));
    float* Dt = (float*)malloc(sizeof(float) * (size_t)(max_len * d));
    float* row = (float*)malloc(sizeof(float) * (size_t)max_len);
#pragma omp for schedule(dynamic, 8)
    for (int64_t i = 0; i < n_dec; ++i) {
      int64_t doc = r[i].id;
      int64_t s = ix->doc_offsets[doc], n = ix->doc_lengths[doc];
      /* index.rs:1159-1179 */
      po_decompress(ix->residuals + s * ix->pd, ix->codes + s, n, d, ix->nbits, ix->centroids,
                    ix->bucket_weights, D);
      e[i].score = po_maxsim_score_scratch(Q, Lq, D, n, d, row, Dt);
      e[i].pos = i;
      e[i].id = doc;
    }
    free(D); free(Dt); free(row);
  }
  if (tr) {
    for (int64_t i = 0; i < n_dec; ++i) {
      if (tr->sel) tr->sel[i] = e[i].id;
      if (tr->sel_exact) tr->sel_exact[i] = e[i].score;
      tr->n_exact_tokens += ix->doc_lengths[e[i].id];
    }
  }
  qsort(e, (size_t)n_dec, sizeof(po_rank), po_rank_cmp);                  /* :496 */
  int64_t rc = p->top_k < n_dec ? p->top_k : n_dec;                       /* :499 */
  for (int64_t i = 0; i < rc; ++i) { out_ids[i] = e[i].id; out_scores[i] = e[i].score; }
  *out_count = (int32_t)rc;
  free(e); free(r);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* dense path: search.rs:327-516                                              */
/* ------------------------------------------------------------------------- */
/* Document-shard view (no reference counterpart; used only by tests/shard_oracle_backend.py to emulate ONE shard of
 * the sharded protocol): the eligible-centroid set, N and |subset| of search.rs:350-382 are those of the WHOLE index,
 * handed in, while `subset` holds the shard-local ids.  NULL = plain reference behaviour. */
typedef struct { const uint8_t* eligible; int64_t n_total; int64_t subset_len; } po_shard_view;

static int po_search_dense(const po_index* ix, const float* Q, int64_t Lq, const po_params* p,
                           const int64_t* subset, int64_t subset_len, int64_t* out_ids,
                           float* out_scores, int32_t* out_count, po_trace* tr, const po_shard_view* sv) {
  const int64_t K = ix->K, N = ix->N;
  *out_count = 0;
  float* qc = (float*)malloc(sizeof(float) * (size_t)(Lq * K > 0 ? Lq * K : 1));
  po_query_centroid_scores(ix, Q, Lq, qc);                                /* :345 */

  /* :350-364 eligible centroids */
  uint8_t* eligible = NULL;
  int64_t n_eligible = 0;
  if (subset_len >= 0) {
    eligible = (uint8_t*)calloc((size_t)(K > 0 ? K : 1), 1);
    if (sv && sv->eligible) {
      for (int64_t c = 0; c < K; ++c) if (sv->eligible[c]) { eligible[c] = 1; ++n_eligible; }
    } else
    for (int64_t i = 0; i < subset_len; ++i) {
      int64_t doc = subset[i];
      if (doc >= 0 && doc < N) { /* `doc_id as usize < len`; negative wraps to huge */
        for (int64_t t = ix->doc_offsets[doc]; t < ix->doc_offsets[doc + 1]; ++t) {
          int64_t c = ix->codes[t];
          if (!eligible[c]) { eligible[c] = 1; ++n_eligible; }
        }
      }
    }
  }
  /* :370-382 effective n_ivf_probe */
  int64_t eff = p->n_ivf_probe;
  if (eligible && n_eligible > 0) {
    int64_t scaled = p->n_ivf_probe;
    const int64_t n_all = sv ? sv->n_total : N, sl_all = sv ? sv->subset_len : subset_len;
    if (sl_all > 0) scaled = (int64_t)((uint64_t)p->n_ivf_probe * (uint64_t)n_all / (uint64_t)sl_all);
    if (scaled < p->n_ivf_probe) scaled = p->n_ivf_probe;
    if (scaled > n_eligible) scaled = n_eligible;
    eff = scaled;
  }

  /* :388-414 per-token top-n selection, union */
  uint8_t* selected = (uint8_t*)calloc((size_t)(K > 0 ? K : 1), 1);
  int64_t pool = eligible ? n_eligible : K;
  po_pair* pairs = (po_pair*)malloc(sizeof(po_pair) * (size_t)(pool > 0 ? pool : 1));
  for (int64_t q = 0; q < Lq; ++q) {
    int64_t m = 0;
    for (int64_t c = 0; c < K; ++c)
      if (!eligible || eligible[c]) { pairs[m].score = qc[q * K + c]; pairs[m].id = c; ++m; }
    int64_t n_probe = eff < m ? eff : m;
    if (m > n_probe) po_select_top(pairs, m, n_probe);
    for (int64_t i = 0; i < n_probe; ++i) selected[pairs[i].id] = 1;
  }
  free(pairs);
  /* :417-425 threshold (max_by returns the last of equal maxima) */
  if (p->has_threshold) {
    for (int64_t c = 0; c < K; ++c) {
      if (!selected[c]) continue;
      float mx = -INFINITY;
      if (Lq > 0) {
        mx = qc[c];
        for (int64_t q = 1; q < Lq; ++q)
          if (po_cmp_score_ascending(qc[q * K + c], mx) >= 0) mx = qc[q * K + c];
      }
      if (!(mx >= p->centroid_score_threshold)) selected[c] = 0;
    }
  }
  int64_t n_cells = 0;
  int64_t* cells = (int64_t*)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
  for (int64_t c = 0; c < K; ++c) if (selected[c]) cells[n_cells++] = c;
  if (tr) { tr->n_cells = n_cells; if (tr->cells) memcpy(tr->cells, cells, sizeof(int64_t) * (size_t)n_cells); }

  /* :431-437 */
  int64_t* cand = NULL; int64_t n_ivf = 0;
  int64_t n_cand = po_get_candidates(ix, cells, n_cells, &cand, &n_ivf);
  if (subset_len >= 0) n_cand = po_retain_subset(cand, n_cand, subset, subset_len);
  if (tr) { tr->n_ivf_ids = n_ivf; tr->n_cand = n_cand; tr->n_cand_tokens = 0; tr->n_sel = 0; tr->n_exact_tokens = 0;
            if (tr->cand) memcpy(tr->cand, cand, sizeof(int64_t) * (size_t)n_cand); }
  free(cells); free(selected); free(eligible);
  if (n_cand == 0) { free(cand); free(qc); return 0; }                    /* :439-445 */

  /* :448-457 approximate scores (rayon par_iter over candidates) */
  float* approx = (float*)malloc(sizeof(float) * (size_t)n_cand);
#pragma omp parallel for schedule(dynamic, 64) num_threads(po_inner_nt())
  for (int64_t i = 0; i < n_cand; ++i) {
    int64_t s = ix->doc_offsets[cand[i]], n = ix->doc_lengths[cand[i]];
    approx[i] = po_approx_score_dense(qc, Lq, K, ix->codes + s, n);
  }
  if (tr) {
    if (tr->approx) memcpy(tr->approx, approx, sizeof(float) * (size_t)n_cand);
    for (int64_t i = 0; i < n_cand; ++i) tr->n_cand_tokens += ix->doc_lengths[cand[i]];
  }
  free(qc);
  int rc = po_rank_and_rescore(ix, Q, Lq, p, cand, approx, n_cand, out_ids, out_scores, out_count, tr);
  free(approx); free(cand);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* batched path: search.rs:140-254 (ivf_probe_batched), :259-302, :521-640     */
/* ------------------------------------------------------------------------- */

/* One (Reverse(OrdF32(score)), usize) max-heap of capacity n_probe, restated as a flat
 * array: peek() = the max tuple = (lowest score by cmp_score_ascending; ties: largest id). */
typedef struct { po_pair* e; int64_t n; int64_t cap; } po_heap;

static int64_t po_heap_peek(const po_heap* h) {
  int64_t w = 0;
  for (int64_t i = 1; i < h->n; ++i) {
    int c = po_cmp_score_ascending(h->e[i].score, h->e[w].score);
    if (c < 0 || (c == 0 && h->e[i].id > h->e[w].id)) w = i;
  }
  return w;
}

static int po_search_batched(const po_index* ix, const float* Q, int64_t Lq, const po_params* p,
                             const int64_t* subset, int64_t subset_len, int64_t* out_ids,
                             float* out_scores, int32_t* out_count, po_trace* tr) {
  const int64_t K = ix->K, d = ix->d;
  const int64_t n_probe = p->n_ivf_probe, bs = p->centroid_batch_size;
  *out_count = 0;
  /* final_max_scores: HashMap<centroid,f32>; NaN-free marker via has_max */
  float* final_max = (float*)malloc(sizeof(float) * (size_t)K);
  uint8_t* has_max = (uint8_t*)calloc((size_t)K, 1);
  po_heap* fin = (po_heap*)malloc(sizeof(po_heap) * (size_t)(Lq > 0 ? Lq : 1));
  for (int64_t q = 0; q < Lq; ++q) { fin[q].e = (po_pair*)malloc(sizeof(po_pair) * (size_t)(n_probe + 1)); fin[q].n = 0; fin[q].cap = n_probe; }
  float* slab = (float*)malloc(sizeof(float) * (size_t)(bs > 0 ? bs : 1));
  po_heap loc; loc.e = (po_pair*)malloc(sizeof(po_pair) * (size_t)(n_probe + 1)); loc.cap = n_probe;

  /* search.rs:151-203: slabs processed independently (par_iter), merged in slab order :212-232 */
  for (int64_t b0 = 0; b0 < K; b0 += bs) {
    int64_t b1 = b0 + bs < K ? b0 + bs : K;
    for (int64_t q = 0; q < Lq; ++q) {
      /* :174 batch_scores row q (GEMM element = k-ordered FMA chain) */
      for (int64_t c = b0; c < b1; ++c) slab[c - b0] = 0.0f;
      for (int64_t k = 0; k < d; ++k) {
        const float qv = Q[q * d + k];
        const float* ct = ix->centroids_t + k * K + b0;
#pragma omp simd
        for (int64_t c = 0; c < b1 - b0; ++c) slab[c] = __builtin_fmaf(qv, ct[c], slab[c]);
      }
      /* :177-199 local heap for token q over this slab */
      loc.n = 0;
      int64_t w = -1; /* cached peek index */
      for (int64_t c = b0; c < b1; ++c) {
        float score = slab[c - b0];
        int pushed = 0;
        if (loc.n < n_probe) {
          loc.e[loc.n].score = score; loc.e[loc.n].id = c; ++loc.n; pushed = 1; w = -1;
        } else if (loc.n > 0) {
          if (w < 0) w = po_heap_peek(&loc);
          if (po_is_score_better(score, loc.e[w].score)) {
            loc.e[w].score = score; loc.e[w].id = c; pushed = 1; w = -1;
          }
        }
        if (pushed) { /* :184-187,192-195 max_scores entry: max_score() finite-first */
          if (has_max[c]) {
            /* local max_scores then merged with f32::max (:226-231); one slab owns c, so the
             * merge never combines two values for the same c; local update uses max_score */
            final_max[c] = po_max_score(final_max[c], score);
          } else { final_max[c] = score; has_max[c] = 1; }
        }
      }
      /* :212-225 merge the local heap of token q into the final heap */
      qsort(loc.e, (size_t)loc.n, sizeof(po_pair), po_pair_desc_cmp); /* iteration order of a
        BinaryHeap is unspecified; descending keeps ties -> lower id (documented choice) */
      for (int64_t i = 0; i < loc.n; ++i) {
        po_heap* f = &fin[q];
        if (f->n < n_probe) { f->e[f->n++] = loc.e[i]; }
        else if (f->n > 0) {
          int64_t fw = po_heap_peek(f);
          if (po_is_score_better(loc.e[i].score, f->e[fw].score)) f->e[fw] = loc.e[i];
        }
      }
    }
  }
  free(slab); free(loc.e);
  /* :235-251 union + threshold on final_max_scores (unwrap_or(-inf) >= t) */
  uint8_t* selected = (uint8_t*)calloc((size_t)K, 1);
  for (int64_t q = 0; q < Lq; ++q) { for (int64_t i = 0; i < fin[q].n; ++i) selected[fin[q].e[i].id] = 1; free(fin[q].e); }
  free(fin);
  if (p->has_threshold)
    for (int64_t c = 0; c < K; ++c)
      if (selected[c]) { float m = has_max[c] ? final_max[c] : -INFINITY; if (!(m >= p->centroid_score_threshold)) selected[c] = 0; }
  free(final_max); free(has_max);
  int64_t n_cells = 0;
  int64_t* cells = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
  for (int64_t c = 0; c < K; ++c) if (selected[c]) cells[n_cells++] = c;
  free(selected);
  if (tr) { tr->n_cells = n_cells; if (tr->cells) memcpy(tr->cells, cells, sizeof(int64_t) * (size_t)n_cells); }

  /* :539-553 */
  int64_t* cand = NULL; int64_t n_ivf = 0;
  int64_t n_cand = po_get_candidates(ix, cells, n_cells, &cand, &n_ivf);
  free(cells);
  if (subset_len >= 0) n_cand = po_retain_subset(cand, n_cand, subset, subset_len);
  if (tr) { tr->n_ivf_ids = n_ivf; tr->n_cand = n_cand; tr->n_cand_tokens = 0; tr->n_sel = 0; tr->n_exact_tokens = 0;
            if (tr->cand) memcpy(tr->cand, cand, sizeof(int64_t) * (size_t)n_cand); }
  if (n_cand == 0) { free(cand); return 0; }

  /* :556-568 sparse centroid scores: one mat-vec (unrolled_dot per row) per unique code */
  uint8_t* have = (uint8_t*)calloc((size_t)K, 1);
  for (int64_t i = 0; i < n_cand; ++i)
    for (int64_t t = ix->doc_offsets[cand[i]]; t < ix->doc_offsets[cand[i] + 1]; ++t) have[ix->codes[t]] = 1;
  float* sparse = (float*)malloc(sizeof(float) * (size_t)(K * (Lq > 0 ? Lq : 1)));  /* [K][Lq] */
#pragma omp parallel for schedule(dynamic, 256) num_threads(po_inner_nt())
  for (int64_t c = 0; c < K; ++c)
    if (have[c])
      for (int64_t q = 0; q < Lq; ++q) sparse[c * Lq + q] = po_unrolled_dot(Q + q * d, ix->centroids + c * d, d);
  free(have);
  /* :571-581 approximate_score_sparse (:275-302); every code is present in the map */
  float* approx = (float*)malloc(sizeof(float) * (size_t)n_cand);
#pragma omp parallel for schedule(dynamic, 64) num_threads(po_inner_nt())
  for (int64_t i = 0; i < n_cand; ++i) {
    int64_t s = ix->doc_offsets[cand[i]], n = ix->doc_lengths[cand[i]];
    float score = 0.0f;
    for (int64_t q = 0; q < Lq; ++q) {
      float mx = -INFINITY;
      for (int64_t t = 0; t < n; ++t) { float cs = sparse[ix->codes[s + t] * Lq + q]; if (cs > mx) mx = cs; }
      if (mx > -INFINITY) score += mx;
    }
    approx[i] = score;
  }
  free(sparse);
  if (tr) {
    if (tr->approx) memcpy(tr->approx, approx, sizeof(float) * (size_t)n_cand);
    for (int64_t i = 0; i < n_cand; ++i) tr->n_cand_tokens += ix->doc_lengths[cand[i]];
  }
  int rc = po_rank_and_rescore(ix, Q, Lq, p, cand, approx, n_cand, out_ids, out_scores, out_count, tr);
  free(approx); free(cand);
  return rc;
}

/* search.rs:327-342 search_one_mmap.  subset_len < 0 means None. Returns 0 ok, 2 invalid. */
PO_API int po_search_one(const po_index* ix, const float* Q, int64_t Lq, const po_params* p,
                         const int64_t* subset, int64_t subset_len, int64_t* out_ids,
                         float* out_scores, int32_t* out_count, po_trace* tr) {
  *out_count = 0;
  if (p->n_ivf_probe < 1 || p->top_k < 0 || p->n_full_scores < 0) return 2;
  int use_batched = p->centroid_batch_size > 0 && ix->K > p->centroid_batch_size; /* :337 */
  if (tr) tr->used_batched = use_batched;
  if (use_batched) return po_search_batched(ix, Q, Lq, p, subset, subset_len, out_ids, out_scores, out_count, tr);
  return po_search_dense(ix, Q, Lq, p, subset, subset_len, out_ids, out_scores, out_count, tr, NULL);
}

/* One document shard's view of a subset search (see po_shard_view): eligible[K] / n_total / subset_len_total are
 * the whole index's; `subset` holds this shard's local ids. */
PO_API int po_search_one_shard(const po_index* ix, const float* Q, int64_t Lq, const po_params* p,
                               const int64_t* subset, int64_t subset_len, const uint8_t* eligible, int64_t n_total,
                               int64_t subset_len_total, int64_t* out_ids, float* out_scores, int32_t* out_count,
                               po_trace* tr) {
  *out_count = 0;
  if (p->n_ivf_probe < 1 || p->top_k < 0 || p->n_full_scores < 0) return 2;
  int use_batched = p->centroid_batch_size > 0 && ix->K > p->centroid_batch_size;
  if (tr) tr->used_batched = use_batched;
  if (use_batched) return po_search_batched(ix, Q, Lq, p, subset, subset_len, out_ids, out_scores, out_count, tr);
  po_shard_view sv = { eligible, n_total, subset_len_total };
  return po_search_dense(ix, Q, Lq, p, subset, subset_len, out_ids, out_scores, out_count, tr, &sv);
}

/* search.rs:643-675 search_many_mmap.  queries concatenated row-major, tok_off[B+1].
 * out_* have stride top_k per query.  parallel!=0: outer parallel over queries, a failing
 * query yields an empty result; parallel==0: sequential, first error returned. */
PO_API int po_search_many(const po_index* ix, const float* queries, const int32_t* tok_off, int32_t B,
                          const po_params* p, int parallel, const int64_t* subset, int64_t subset_len,
                          int64_t* out_ids, float* out_scores, int32_t* out_counts) {
  int err = 0;
  if (parallel) {
    int outer = 1;
#ifdef _OPENMP
    int T = omp_get_max_threads();
    outer = B < T ? B : T; if (outer < 1) outer = 1;
    omp_set_max_active_levels(2); /* nested rayon: outer over queries, inner over candidates */
    po_inner_threads = T / outer > 0 ? T / outer : 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(outer)
    for (int32_t i = 0; i < B; ++i) {
      int32_t cnt = 0;
      int rc = po_search_one(ix, queries + (int64_t)tok_off[i] * ix->d, tok_off[i + 1] - tok_off[i], p,
                             subset, subset_len, out_ids + (int64_t)i * p->top_k,
                             out_scores + (int64_t)i * p->top_k, &cnt, NULL);
      out_counts[i] = rc ? 0 : cnt;
    }
    po_inner_threads = 0;
  } else {
    for (int32_t i = 0; i < B && !err; ++i) {
      int32_t cnt = 0;
      err = po_search_one(ix, queries + (int64_t)tok_off[i] * ix->d, tok_off[i + 1] - tok_off[i], p,
                          subset, subset_len, out_ids + (int64_t)i * p->top_k,
                          out_scores + (int64_t)i * p->top_k, &cnt, NULL);
      out_counts[i] = cnt;
    }
  }
  return err;
}

PO_API int po_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
