// Micro-benchmark: random 128-byte row gathers (the S4 access pattern) on MI355X.
// Sweeps table size, loads in flight per wave, resident waves per CU, and an XCD-sliced variant
// (block b -> XCD b%8 reads only slice b%8 of the table).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int INF, int LPR>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ T, unsigned row_mask, int iters, int sliced,
                                              unsigned slice_shift, float* out, int lds_pad) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, jq = lane % LPR;
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / LPR) * 0x9E3779B97F4A7C15ull + 12345;
  float4 acc = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
  const unsigned xcd = blockIdx.x & 7;
  for (int it = 0; it < iters; ++it) {
    float4 v[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      unsigned r = (unsigned)(s >> 33) & row_mask;
      if (sliced) r = (r >> 3) | (xcd << slice_shift);
      v[u] = *reinterpret_cast<const float4*>(T + ((size_t)r * (LPR * 16) + jq * 16));
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      acc.x = fmaxf(acc.x, v[u].x); acc.y = fmaxf(acc.y, v[u].y);
      acc.z = fmaxf(acc.z, v[u].z); acc.w = fmaxf(acc.w, v[u].w);
    }
  }
  if (lds_pad < 0) pad[threadIdx.x] = 1;
  if (acc.x + acc.y + acc.z + acc.w == 123.f) out[grp] = acc.x;
}

template <int INF, int LPR>
double run(const char* T, unsigned rows, int sliced, int blocks, int lds, float* out) {
  const int iters = 4096 / INF;
  unsigned shift = 0; while ((1u << shift) < rows) ++shift;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<INF, LPR><<<blocks, 256, lds>>>(T, rows - 1, iters, sliced, shift - 3, out, lds);
  CK(hipEventRecord(a));
  gather<INF, LPR><<<blocks, 256, lds>>>(T, rows - 1, iters, sliced, shift - 3, out, lds);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double nrows = (double)blocks * 256 / LPR * iters * INF;
  return nrows / (ms * 1e-3) * 1e-9;
}

int main() {
  const size_t bytes = 1ull << 30;
  char* T; CK(hipMalloc(&T, bytes)); CK(hipMemset(T, 0, bytes));
  float* out; CK(hipMalloc(&out, 4096));
  printf("rowB table_MB sliced waves/CU inflight Grows/s TB/s\n");
  for (int rowB : {128, 64}) {
    for (unsigned mb : {1u, 4u, 8u, 32u, 128u, 512u}) {
      unsigned rows = (unsigned)((size_t)mb * 1048576 / rowB);
      for (int sliced : {0, 1}) {
        for (int wpc : {8, 16, 32}) {
          // waves per CU controlled by LDS: 160 KB / (blocks per CU); 4 waves per block
          int bpc = wpc / 4; int lds = 160 * 1024 / bpc - 1024; if (lds > 65536) lds = 65536;
          int blocks = 256 * bpc;
          for (int inf : {4, 8, 16}) {
            double g;
            if (rowB == 128) g = inf == 4 ? run<4, 8>(T, rows, sliced, blocks, lds, out) : inf == 8 ? run<8, 8>(T, rows, sliced, blocks, lds, out) : run<16, 8>(T, rows, sliced, blocks, lds, out);
            else g = inf == 4 ? run<4, 4>(T, rows, sliced, blocks, lds, out) : inf == 8 ? run<8, 4>(T, rows, sliced, blocks, lds, out) : run<16, 4>(T, rows, sliced, blocks, lds, out);
            printf("%d %u %d %d %d %.1f %.2f\n", rowB, mb, sliced, wpc, inf, g, g * rowB * 1e-3);
          }
        }
      }
    }
  }
  return 0;
}
