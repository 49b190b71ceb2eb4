// What does FETCH_SIZE (rocprofv3, gfx950) count for the access patterns of the S4 hot level, and how fast can HBM deliver them?
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide streaming reads (x2: 128-B requests tallied at 64 B) and asks for a
// calibration "on a known byte count in your own access pattern".  approx_hotp_kernel reads one list BLOCK per candidate
// document: 192 bytes at a 192-byte stride on the metric corpus, at random (1.9 % of the documents are candidates), 16 bytes
// per lane -- so a block always touches two 128-byte lines.  This probe reads a KNOWN number of blocks of a table far larger
// than the L2s and the Infinity Cache (8 GiB) in that pattern and in the alternatives a new block layout could use:
//
//   stream            16 B per lane, grid-stride over the whole table            (the guide's calibration case)
//   blk192_s192       192-B blocks at a 192-B stride          (today's layout: 2 lines per block, 256 B for 192)
//   blk128_s128       128-B blocks, 128-B aligned             (one line per block)
//   blk256_s256       256-B blocks, 256-B aligned             (two adjacent lines)
//   blk64_s64         64-B blocks, 64-B aligned               (half a line: does the fabric fetch 64 or 128 bytes?)
//   blk64_s128        64-B blocks at a 128-B stride           (the same, never two blocks in one line)
//   blk32_s128        32-B rows at a 128-B stride
//
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum in further passes):
// the kernel names carry the pattern, stdout carries the known byte counts and the time-derived rates;
// tools/probes/fetch_probe_summary.py joins the two.
// Build: hipcc --offload-arch=gfx950 -O3 fetch_probe.hip -o fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ T, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const u32x4 v = __builtin_nontemporal_load(T + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// every group of LPR = BS / 16 lanes reads one random block per step, INF steps in flight; `blocks_per_wave_step` blocks per
// wave instruction (idle lanes read nothing).  Block ids are a hash of (wave, step, slot): uniform over the table, no reuse
// to speak of (n_reads << n_blocks would still hit the same line twice now and then; the host reports the expected count).
template <int BS, int STRIDE, int INF>
__global__ __launch_bounds__(256) void blk_kernel(const char* __restrict__ T, uint64_t n_blocks, int steps, uint32_t* out) {
  constexpr int LPR = BS / 16;
  constexpr int BPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, piece = lane - slot * LPR;
  const bool live = slot < BPW;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s += INF) {
    u32x4 v[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      const uint64_t id = mix64((wave * 1000003ull + (uint64_t)(s + u)) * 64ull + (uint64_t)slot) % n_blocks;
      const u32x4* p = reinterpret_cast<const u32x4*>(T + id * (uint64_t)STRIDE + 16 * piece);
      v[u] = live ? __builtin_nontemporal_load(p) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[lane] = acc;
}

template <int BS, int STRIDE>
void run_blk(const char* name, const char* T, size_t table_bytes, uint32_t* out) {
  constexpr int INF = 8;
  constexpr int LPR = BS / 16, BPW = 64 / LPR;
  const uint64_t n_blocks = table_bytes / STRIDE;
  const int grid = 256 * 8, steps = 512;     // 8192 waves x 512 steps
  const double reads = (double)grid * 4 * steps * BPW;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  blk_kernel<BS, STRIDE, INF><<<grid, 256>>>(T, n_blocks, 64, out);   // warm the TLBs
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  blk_kernel<BS, STRIDE, INF><<<grid, 256>>>(T, n_blocks, steps, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  // 128-byte lines a block touches: a BS-byte block at offset id * STRIDE
  double lines = 0;
  for (int k = 0; k < 128; ++k) {   // offsets repeat with period lcm(STRIDE, 128) <= 128 * STRIDE
    const uint64_t o = (uint64_t)k * STRIDE;
    lines += (double)((o + BS - 1) / 128 - o / 128 + 1);
  }
  lines /= 128.0;
  printf("PROBE %s block_bytes %d stride %d reads %.0f payload_GB %.4f lines128_per_block %.3f line_GB %.4f ms %.3f payload_GBps %.1f line_GBps %.1f Mblocks_per_s %.1f\n",
         name, BS, STRIDE, reads, reads * BS / 1e9, lines, reads * lines * 128 / 1e9, ms, reads * BS / 1e9 / (ms * 1e-3),
         reads * lines * 128 / 1e9 / (ms * 1e-3), reads / 1e6 / (ms * 1e-3));
}

int main(int argc, char** argv) {
  const size_t table_bytes = (size_t)(argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);
  char* T = nullptr;
  uint32_t* out = nullptr;
  CK(hipMalloc(&T, table_bytes + 4096));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(T, 1, table_bytes + 4096));
  CK(hipDeviceSynchronize());
  {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    stream_kernel<<<256 * 16, 256>>>(reinterpret_cast<const u32x4*>(T), table_bytes / 16, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    stream_kernel<<<256 * 16, 256>>>(reinterpret_cast<const u32x4*>(T), table_bytes / 16, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("PROBE stream block_bytes 16 stride 16 reads %.0f payload_GB %.4f lines128_per_block 0.125 line_GB %.4f ms %.3f payload_GBps %.1f line_GBps %.1f Mblocks_per_s 0\n",
           (double)(table_bytes / 16), table_bytes / 1e9, table_bytes / 1e9, ms, table_bytes / 1e9 / (ms * 1e-3), table_bytes / 1e9 / (ms * 1e-3));
  }
  run_blk<192, 192>("blk192_s192", T, table_bytes, out);
  run_blk<128, 128>("blk128_s128", T, table_bytes, out);
  run_blk<256, 256>("blk256_s256", T, table_bytes, out);
  run_blk<64, 64>("blk64_s64", T, table_bytes, out);
  run_blk<64, 128>("blk64_s128", T, table_bytes, out);
  run_blk<32, 128>("blk32_s128", T, table_bytes, out);
  CK(hipFree(T));
  CK(hipFree(out));
  return 0;
}
