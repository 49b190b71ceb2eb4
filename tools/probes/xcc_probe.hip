// Which XCD does block b land on?  (HW_REG_XCC_ID, see MI355X_MICROARCH.md "Workgroup dispatch")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = x & 0xF;
  }
}
int main() {
  for (int n : {64, 4096}) {
    int* d; hipMalloc(&d, n * 4);
    k<<<n, 256>>>(d);
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int ok = 0; for (int i = 0; i < n; ++i) ok += (h[i] == (i % 8));
    printf("grid %d: blocks with xcc == b%%8: %d / %d ; first 24:", n, ok, n);
    for (int i = 0; i < 24; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
