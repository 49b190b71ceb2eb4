// Micro-benchmark mimicking the S4 filter's walk: 32-B rows from a 2 MB table per XCD, 2 lanes per row (uint4 each),
// two batches of 8 gathers in flight, optional per-byte max folding (16 SDWA v_max_u32 per gathered 16 B), at 8 / 12 /
// 16 waves per CU.  Shows what the memory pipe gives at the kernel's occupancy and what the VALU folding costs.
// Build: hipcc --offload-arch=gfx950 -O3 gather_probe3.hip -o gather_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool FOLD, bool PIPE>
__global__ __launch_bounds__(256) void walk(const char* __restrict__ T, unsigned row_mask, size_t table_bytes, int iters,
                                            uint32_t* out) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x & 63, jl = lane & 1;
  const char* Tb = T + (size_t)(blockIdx.x & 7) * table_bytes + jl * 16;
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / 2) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t st[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) st[k] = 0;
  auto issue = [&](uint4 (&v)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned r = (unsigned)(s >> 33) & row_mask;
      v[u] = *reinterpret_cast<const uint4*>(Tb + (size_t)r * 32);
    }
  };
  auto fold = [&](uint4 (&v)[8]) {
    asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x), "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      if (FOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) st[4 * j + e] = max(st[4 * j + e], (w4[j] >> (8 * e)) & 0xFFu);
      } else {
        st[k & 15] = max(st[k & 15], w4[0] ^ w4[1] ^ w4[2] ^ w4[3]);
      }
    }
  };
  uint4 va[8], vb[8];
  if (PIPE) {
    issue(va);
    for (int it = 0; it < iters; it += 2) {
      issue(vb);
      fold(va);
      issue(va);
      fold(vb);
    }
    fold(va);
  } else {
    for (int it = 0; it < iters; ++it) {
      issue(va);
      fold(va);
    }
  }
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += st[k];
  if (sum == 0x12345678u) out[lane] = sum;
  if (iters < 0) pad[threadIdx.x] = 1;
}

template <bool FOLD, bool PIPE>
double run(const char* T, size_t tb, int bpc, uint32_t* out) {
  const int iters = 512;
  int lds = 160 * 1024 / bpc - 2048;
  if (lds > 65536) lds = 65536;
  const int blocks = 256 * bpc;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  walk<FOLD, PIPE><<<blocks, 256, lds>>>(T, (unsigned)(tb / 32) - 1, tb, iters, out);
  CK(hipEventRecord(a));
  walk<FOLD, PIPE><<<blocks, 256, lds>>>(T, (unsigned)(tb / 32) - 1, tb, iters, out);
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
  printf("waves/CU fold pipe Grows/s\n");
  for (int bpc : {2, 3, 4}) {
    printf("%d 0 0 %.1f\n", bpc * 4, run<false, false>(T, 2u << 20, bpc, out));
    printf("%d 0 1 %.1f\n", bpc * 4, run<false, true>(T, 2u << 20, bpc, out));
    printf("%d 1 0 %.1f\n", bpc * 4, run<true, false>(T, 2u << 20, bpc, out));
    printf("%d 1 1 %.1f\n", bpc * 4, run<true, true>(T, 2u << 20, bpc, out));
  }
  return 0;
}
