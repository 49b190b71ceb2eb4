// Micro-benchmark for the S4 upper-bound filter design: random gathers of SHORT rows (32 / 64 / 128 bytes)
// from a table that sits in ONE XCD's L2 (block b -> XCD b % 8 reads only table b % 8; a 2 MB table is one
// query's u8 score table at K = 65536, Lq = 32).  Sweeps row bytes, lanes per row (bytes per lane), loads in
// flight and waves per CU.  Rates are G rows/s over the whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 gather_probe2.hip -o gather_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int VB> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };
__device__ inline uint32_t fold(uint32_t a, uint32_t v) { return max(a, v); }
__device__ inline uint32_t fold(uint32_t a, uint2 v) { return max(a, max(v.x, v.y)); }
__device__ inline uint32_t fold(uint32_t a, uint4 v) { return max(a, max(max(v.x, v.y), max(v.z, v.w))); }

// ROWB bytes per row, VB bytes per lane per load, LPR = ROWB / VB lanes cover a row (LPR >= 1);
// when VB * 1 lane < ROWB is wanted (one lane, several loads per row) use NL loads per lane per row.
template <int ROWB, int VB, int NL, int INF>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ T, unsigned row_mask, size_t table_bytes, int iters,
                                              uint32_t* out) {
  constexpr int LPR = ROWB / (VB * NL) > 0 ? ROWB / (VB * NL) : 1;
  typedef typename Vec<VB>::T V;
  const int lane = threadIdx.x & 63;
  const int jq = lane % LPR;
  const char* Tb = T + (size_t)(blockIdx.x & 7) * table_bytes;
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / LPR) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    V v[INF][NL];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned r = (unsigned)(s >> 33) & row_mask;
#pragma unroll
      for (int k = 0; k < NL; ++k)
        v[u][k] = *reinterpret_cast<const V*>(Tb + ((size_t)r * ROWB + (jq * NL + k) * VB));
    }
#pragma unroll
    for (int u = 0; u < INF; ++u)
#pragma unroll
      for (int k = 0; k < NL; ++k) acc = fold(acc, v[u][k]);
  }
  if (acc == 0x12345678u) out[lane] = acc;
}

template <int ROWB, int VB, int NL, int INF>
double run(const char* T, size_t table_bytes, int blocks, int lds, uint32_t* out) {
  constexpr int LPR = ROWB / (VB * NL) > 0 ? ROWB / (VB * NL) : 1;
  const int iters = 2048 / INF;
  const unsigned rows = (unsigned)(table_bytes / ROWB);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  gather<ROWB, VB, NL, INF><<<blocks, 256, lds>>>(T, rows - 1, table_bytes, iters, out);
  CK(hipEventRecord(a));
  gather<ROWB, VB, NL, INF><<<blocks, 256, lds>>>(T, rows - 1, table_bytes, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double nrows = (double)blocks * 256 / LPR * iters * INF;
  return nrows / (ms * 1e-3) * 1e-9;
}

int main() {
  const size_t bytes = 256ull << 20;
  char* T;
  CK(hipMalloc(&T, bytes));
  CK(hipMemset(T, 1, bytes));
  uint32_t* out;
  CK(hipMalloc(&out, 4096));
  printf("rowB laneB loads/lane/row table_MB waves/CU inflight Grows/s TB/s\n");
  for (unsigned mb : {1u, 2u, 4u, 8u}) {
    const size_t tb = (size_t)mb << 20;
    for (int wpc : {16, 32}) {
      const int bpc = wpc / 4;
      int lds = 160 * 1024 / bpc - 1024;
      if (lds > 65536) lds = 65536;
      const int blocks = 256 * bpc;
#define R(ROWB, VB, NL, INF)                                                                  \
  {                                                                                           \
    double g = run<ROWB, VB, NL, INF>(T, tb, blocks, lds, out);                               \
    printf("%d %d %d %u %d %d %.1f %.2f\n", ROWB, VB, NL, mb, wpc, INF, g, g * ROWB * 1e-3); \
  }
      R(32, 16, 1, 8) R(32, 16, 1, 16)      // 2 lanes per 32-B row, uint4 each
      R(32, 8, 1, 8) R(32, 8, 1, 16)        // 4 lanes per row, uint2 each
      R(32, 16, 2, 4) R(32, 16, 2, 8)       // 1 lane per row, two uint4 loads
      R(32, 4, 1, 8) R(32, 4, 1, 16)        // 8 lanes per row, dword each
      R(64, 16, 1, 8) R(64, 16, 1, 16)      // 64-B rows, 4 lanes
      R(128, 16, 1, 8) R(128, 16, 1, 16)    // the f32 row of today, 8 lanes
#undef R
    }
  }
  return 0;
}
