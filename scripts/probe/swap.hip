#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 31) printf("lane %d: r0=%u r1=%u\n", i, h[2*i], h[2*i+1]);
  printf("lane 32: r0=%u r1=%u\nlane 33: r0=%u r1=%u\n", h[64], h[65], h[66], h[67]);
  return 0;
}
