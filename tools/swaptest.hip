#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[v * 64 + l]); printf("\n"); }
  return 0;
}
