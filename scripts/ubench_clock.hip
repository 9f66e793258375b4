// dev tool: shader clock (s_memtime) vs real time (s_memrealtime, 100 MHz) inside short and long kernels
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_clk(unsigned long long *out, int iters) {
  unsigned long long c0 = clock64(), r0 = wall_clock64();
  float v = threadIdx.x; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  unsigned long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}
int main() {
  unsigned long long *d, h[3]; hipMalloc(&d, 24);
  int its[] = {1000, 10000, 100000, 1000000, 10000000};
  for (int rep = 0; rep < 2; ++rep)
  for (int k = 0; k < 5; ++k) {
    k_clk<<<352, 256>>>(d, its[k]); hipDeviceSynchronize(); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("iters %8d: shader cycles %llu, real %.2f us -> %.0f MHz, %.2f cycles/iter (dependent v_fma_f32)\n", its[k], h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / its[k]);
  }
  // many short kernels back to back, then measure again
  for (int i = 0; i < 20000; ++i) k_clk<<<352, 256>>>(d, 100);
  hipDeviceSynchronize();
  k_clk<<<352, 256>>>(d, 1000); hipDeviceSynchronize(); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("after 20000 short launches: iters 1000: %.0f MHz\n", h[0] / (h[1] / 100.0));
  return 0; }
