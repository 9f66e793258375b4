// dev tool: what does a dependent kernel launch cost on this box as a function of what is inside?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float *a) {}
__global__ void k_ldst(const float *in, float *out) { int i = blockIdx.x * 256 + threadIdx.x; out[i] = in[i] + 1.f; }
__global__ void k_dep2(const int *idx, const float *in, float *out) { int i = blockIdx.x * 256 + threadIdx.x; int j = idx[i]; out[i] = in[j] + 1.f; }
__global__ void k_dep3(const int *idx, const int *idx2, const float *in, float *out) { int i = blockIdx.x * 256 + threadIdx.x; int j = idx[i]; int k = idx2[j]; out[i] = in[k] + 1.f; }
__global__ void k_atomic(int *ctr, const float *in, float *out) { int i = blockIdx.x * 256 + threadIdx.x; out[i] = in[i]; if (threadIdx.x == 0) atomicMax(ctr, (int)blockIdx.x); }
__global__ void k_bar4(const float *in, float *out) { __shared__ float s[256]; int i = blockIdx.x * 256 + threadIdx.x; float v = in[i];
  for (int r = 0; r < 4; ++r) { s[threadIdx.x] = v; __syncthreads(); v = s[(threadIdx.x + 64) & 255] + 1.f; __syncthreads(); } out[i] = v; }
__global__ void k_alu(const float *in, float *out, int n) { int i = blockIdx.x * 256 + threadIdx.x; double v = in[i]; for (int r = 0; r < n; ++r) v = v * 1.0000001 + 0.5; out[i] = (float)v; }
__global__ void k_loads88(const unsigned *tab, float *out, int ngroups, int stride) { int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; unsigned t = 0;
  if (lane < 16) { const unsigned *col = tab + lane;
#pragma unroll 8
    for (int g = wave; g < ngroups; g += 4) t += col[(size_t)g * stride]; }
  out[blockIdx.x * 256 + threadIdx.x] = (float)t; }
__global__ void k_scatter16(const int *idx, float4 *out) { int i = blockIdx.x * 256 + threadIdx.x; out[idx[i]] = make_float4(1.f, 2.f, 3.f, (float)i); }

template <typename F> double time_graph(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch();
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3 / (5.0 * n);
}

int main() {
  const int NB = 352, N = NB * 256;
  float *a, *b; int *idx, *idx2, *ctr; unsigned *tab; float4 *ev;
  CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&idx2, N * 4)); CK(hipMalloc(&ctr, 4));
  CK(hipMalloc(&tab, 352 * 130 * 4)); CK(hipMalloc(&ev, N * 16));
  std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 7919) % N);
  CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(idx2, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(a, 0, N * 4)); CK(hipMemset(tab, 0, 352 * 130 * 4)); CK(hipMemset(ctr, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int L = 300;
  printf("empty            %.2f us/launch\n", time_graph(s, L, [&] { k_empty<<<NB, 256, 0, s>>>(a); }));
  printf("load+store       %.2f\n", time_graph(s, L, [&] { k_ldst<<<NB, 256, 0, s>>>(a, b); }));
  printf("ld+st pingpong   %.2f\n", time_graph(s, L / 2, [&] { k_ldst<<<NB, 256, 0, s>>>(a, b); k_ldst<<<NB, 256, 0, s>>>(b, a); }));
  printf("2 dependent lds  %.2f\n", time_graph(s, L, [&] { k_dep2<<<NB, 256, 0, s>>>(idx, a, b); }));
  printf("3 dependent lds  %.2f\n", time_graph(s, L, [&] { k_dep3<<<NB, 256, 0, s>>>(idx, idx2, a, b); }));
  printf("atomicMax/block  %.2f\n", time_graph(s, L, [&] { k_atomic<<<NB, 256, 0, s>>>(ctr, a, b); }));
  printf("4 barrier pairs  %.2f\n", time_graph(s, L, [&] { k_bar4<<<NB, 256, 0, s>>>(a, b); }));
  printf("f64 fma x200     %.2f\n", time_graph(s, L, [&] { k_alu<<<NB, 256, 0, s>>>(a, b, 200); }));
  printf("f64 fma x1000    %.2f\n", time_graph(s, L, [&] { k_alu<<<NB, 256, 0, s>>>(a, b, 1000); }));
  printf("88 col loads     %.2f\n", time_graph(s, L, [&] { k_loads88<<<NB, 256, 0, s>>>(tab, b, 352, 130); }));
  printf("scatter 16B      %.2f\n", time_graph(s, L, [&] { k_scatter16<<<NB, 256, 0, s>>>(idx, ev); }));
  printf("empty 88 blocks  %.2f\n", time_graph(s, L, [&] { k_empty<<<88, 256, 0, s>>>(a); }));
  printf("empty 1 block    %.2f\n", time_graph(s, L, [&] { k_empty<<<1, 64, 0, s>>>(a); }));
  printf("ld+st 1408 blk64 %.2f\n", time_graph(s, L, [&] { k_ldst<<<NB * 4, 64, 0, s>>>(a, b); }));
  return 0;
}
