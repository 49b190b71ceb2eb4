// Micro-benchmark: does the ORDER of the row indices matter for L2-resident 32-B row gathers?  The S4 filter walks
// SORTED code lists in lockstep: at step k every document of a wave is near the k-th 68th of the table.
//   mode 0: iid random rows (gather_probe3)
//   mode 1: step k of a 68-step walk reads band k of 68 (960 rows), random inside the band -- all lanes of all waves
//           of a block are in the same band at the same step, different blocks start at different steps
//   mode 2: as 1, every lane pair starts at its own step (bands decorrelated inside the wave)
// Build: hipcc --offload-arch=gfx950 -O3 gather_probe4.hip -o gather_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void walk(const char* __restrict__ T, size_t table_bytes, int iters, uint32_t* out) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x & 63, jl = lane & 1;
  const char* Tb = T + (size_t)(blockIdx.x & 7) * table_bytes + jl * 16;
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / 2) * 0x9E3779B97F4A7C15ull + 12345;
  const unsigned rows = (unsigned)(table_bytes / 32), band = rows / 68;
  unsigned step = MODE == 2 ? (unsigned)((blockIdx.x * 128 + threadIdx.x / 2) * 7u) % 68u : (unsigned)(blockIdx.x * 13u + (threadIdx.x >> 6) * 5u) % 68u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      unsigned r;
      if (MODE == 0) r = (unsigned)(s >> 33) % rows;
      else {
        r = step * band + (unsigned)(s >> 33) % band;
        step = step + 1 == 68 ? 0 : step + 1;
      }
      v[u] = *reinterpret_cast<const uint4*>(Tb + (size_t)r * 32);
    }
    asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = max(acc, v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
  }
  if (acc == 0x12345678u) out[lane] = acc;
  if (iters < 0) pad[threadIdx.x] = 1;
}

template <int MODE>
double run(const char* T, size_t tb, uint32_t* out) {
  const int iters = 512, blocks = 768;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  walk<MODE><<<blocks, 256, 49152>>>(T, tb, iters, out);
  CK(hipEventRecord(a));
  walk<MODE><<<blocks, 256, 49152>>>(T, tb, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return (double)blocks * 128 * iters * 8 / (ms * 1e-3) * 1e-9;
}

int main() {
  char* T;
  CK(hipMalloc(&T, 64ull << 20));
  CK(hipMemset(T, 1, 64ull << 20));
  uint32_t* out;
  CK(hipMalloc(&out, 4096));
  printf("mode Grows/s (2 MB table per XCD, 12 waves/CU)\n");
  printf("0 %.1f\n", run<0>(T, 2u << 20, out));
  printf("1 %.1f\n", run<1>(T, 2u << 20, out));
  printf("2 %.1f\n", run<2>(T, 2u << 20, out));
  return 0;
}
