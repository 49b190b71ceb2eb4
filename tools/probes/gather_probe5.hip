// Does bypassing / streaming through the per-CU vector L1 change the L2-resident row-gather rate?
// gather_probe2 found ~266 G rows/s for 32-, 64- and 128-B rows alike (plain global loads).  If that ceiling is the L1's
// 128-B line fill (64 B/clk/CU from L2), cache-policy bits on the loads (sc0 / sc1 / nt through a buffer descriptor's
// aux field) could lift it for short rows.  Rows: 32 B (2 lanes x 16 B) and 64 B; table 2 MB per XCD; 16 loads in flight.
// Build: hipcc --offload-arch=gfx950 -O3 gather_probe5.hip -o gather_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int ROWB, int AUX, int INF>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ T, unsigned row_mask, size_t table_bytes, int iters,
                                              uint32_t* out) {
  constexpr int LPR = ROWB / 16;
  const int lane = threadIdx.x & 63;
  const int jq = lane % LPR;
  const char* Tb = T + (size_t)(blockIdx.x & 7) * table_bytes;
  const uint64_t a = reinterpret_cast<uint64_t>(Tb);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                                      (int)table_bytes, 0x00020000);
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / LPR) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned r = (unsigned)(s >> 33) & row_mask;
      if constexpr (AUX >= 0) {
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(r * ROWB + jq * 16), 0, AUX);
      } else {   // plain global load / nontemporal builtin
        const uint32_t* p = reinterpret_cast<const uint32_t*>(Tb + (size_t)r * ROWB + jq * 16);
        if constexpr (AUX == -1) {
          const uint4 x = *reinterpret_cast<const uint4*>(p);
          v[u] = u32x4{x.x, x.y, x.z, x.w};
        } else {
          v[u] = u32x4{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2),
                       __builtin_nontemporal_load(p + 3)};
        }
      }
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) acc = max(acc, max(max(v[u].x, v[u].y), max(v[u].z, v[u].w)));
  }
  if (acc == 0x12345678u) out[lane] = acc;
}

template <int ROWB, int AUX, int INF>
double run(const char* T, size_t table_bytes, int blocks, uint32_t* out) {
  constexpr int LPR = ROWB / 16;
  const int iters = 2048 / INF;
  const unsigned rows = (unsigned)(table_bytes / ROWB);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  gather<ROWB, AUX, INF><<<blocks, 256>>>(T, rows - 1, table_bytes, iters, out);
  CK(hipEventRecord(a));
  gather<ROWB, AUX, INF><<<blocks, 256>>>(T, rows - 1, table_bytes, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return (double)blocks * 256 / LPR * iters * INF / (ms * 1e-3) * 1e-9;
}

int main() {
  const size_t bytes = 64ull << 20;
  char* T;
  CK(hipMalloc(&T, bytes));
  CK(hipMemset(T, 1, bytes));
  uint32_t* out;
  CK(hipMalloc(&out, 4096));
  printf("rowB aux(-1 plain global, -2 nontemporal builtin, >=0 buffer aux bits sc0=1 sc1=2 nt=4) table_MB Grows/s\n");
  for (unsigned mb : {2u, 4u}) {
    const size_t tb = (size_t)mb << 20;
    const int blocks = 256 * 3;   // 12 waves per CU, as the filter kernel runs
#define R(ROWB, AUX) printf("%d %d %u %.1f\n", ROWB, AUX, mb, run<ROWB, AUX, 16>(T, tb, blocks, out));
    R(32, -1) R(32, -2) R(32, 0) R(32, 1) R(32, 2) R(32, 3) R(32, 4) R(32, 5) R(32, 6) R(32, 7)
    R(64, -1) R(64, 0) R(64, 2) R(64, 4) R(64, 6)
    R(128, -1) R(128, 2) R(128, 4)
#undef R
  }
  return 0;
}
