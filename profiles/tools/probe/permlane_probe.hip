// What v_permlane16_swap / v_permlane32_swap do to a wave (probe for csrc/strip_gemm.h row_sum4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto u = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + threadIdx.x] = u[0]; out[192 + threadIdx.x] = u[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; i += 1) printf(" %u", h[j * 64 + i]); printf("\n"); }
}
