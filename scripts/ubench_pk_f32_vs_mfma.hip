// Micro-benchmark behind -fno-slp-vectorize: does a packed-float32 VALU instruction return the same value as
// the two scalar instructions it stands for while a matrix-core kernel of ANOTHER stream runs on the same CUs?
//   victim: every lane computes z = x * a + b twice -- with v_pk_mul_f32 / v_pk_add_f32 (two values per instruction) and with
//           v_mul_f32 / v_add_f32 -- on operands that use all 24 mantissa bits, in a loop, and counts the results that differ
//           (and, separately, the packed results that differ from the value computed on the host)
//   noise:  waves issuing v_mfma_f32_32x32x16_bf16 / _f16 / v_mfma_f32_32x32x2_f32 back to back on a second stream
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_pk scripts/ubench_pk_f32_vs_mfma.hip ; run on the GPU box: ./ubench_pk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_victim(const float *__restrict__ x, float a, float b, int iters, unsigned *mism_pk_vs_scalar,
                                                unsigned *mism_pk_vs_host, const float *__restrict__ want)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    const float x0 = x[i], x1 = x[i + 1];
    const float w0 = want[i], w1 = want[i + 1];
    unsigned d_ps = 0, d_ph = 0;
    for (int it = 0; it < iters; ++it) {
        f2 xv = {x0, x1}, av = {a, a}, bv = {b, b}, pk;
        asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(pk) : "v"(xv), "v"(av), "v"(bv));
        float s0, s1;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_add_f32 %0, %0, %5\n\tv_mul_f32 %1, %3, %4\n\tv_add_f32 %1, %1, %5"
                     : "=&v"(s0), "=&v"(s1) : "v"(x0), "v"(x1), "v"(a), "v"(b));
        d_ps += (__float_as_uint(pk.x) != __float_as_uint(s0)) + (__float_as_uint(pk.y) != __float_as_uint(s1));
        d_ph += (__float_as_uint(pk.x) != __float_as_uint(w0)) + (__float_as_uint(pk.y) != __float_as_uint(w1));
    }
    if (d_ps) atomicAdd(mism_pk_vs_scalar, d_ps);
    if (d_ph) atomicAdd(mism_pk_vs_host, d_ph);
}

template <int KIND> __global__ __launch_bounds__(256) void k_noise(float *out, int iters)
{
    f16v acc = {0};
    const float s = (float)(threadIdx.x & 7) * 0.125f + 0.5f;
    if (KIND == 0) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(s + e); b[e] = (__bf16)(1.0f / (s + e)); }
        for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if (KIND == 1) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(1.0f / (s + e)); }
        for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 1.0f / s, acc, 0, 0, 0);
    }
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += acc[e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main()
{
    const int NB = 4096, N = NB * 256 * 2;
    std::vector<float> hx(N), hw(N);
    uint32_t st = 12345u;
    const float a = 1.2345678f, b = 0.7654321f;
    for (int i = 0; i < N; ++i) {
        st = st * 1664525u + 1013904223u;
        hx[i] = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
        volatile float m = hx[i] * a; // separately rounded, as the two instructions do
        volatile float r = m + b;
        hw[i] = r;
    }
    float *dx, *dw, *dn;
    unsigned *dc;
    CK(hipMalloc(&dx, N * 4)); CK(hipMalloc(&dw, N * 4)); CK(hipMalloc(&dn, 1024 * 256 * 4)); CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(dx, hx.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), N * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char *names[4] = {"no noise", "bf16 MFMA on another stream", "f16 MFMA on another stream", "f32 MFMA on another stream"};
    for (int kind = -1; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(dc, 0, 8));
            CK(hipDeviceSynchronize());
            if (kind == 0) k_noise<0><<<1024, 256, 0, s2>>>(dn, 200000);
            if (kind == 1) k_noise<1><<<1024, 256, 0, s2>>>(dn, 200000);
            if (kind == 2) k_noise<2><<<1024, 256, 0, s2>>>(dn, 50000);
            for (int l = 0; l < 20; ++l) k_victim<<<NB, 256, 0, s1>>>(dx, a, b, 2000, dc, dc + 1, dw);
            CK(hipStreamSynchronize(s1));
            hipError_t q = hipStreamQuery(s2); // still running = the victim launches did overlap it
            CK(hipDeviceSynchronize());
            unsigned c[2];
            CK(hipMemcpy(c, dc, 8, hipMemcpyDeviceToHost));
            printf("%-30s rep %d: packed != scalar %u, packed != host %u  (of %.3g results; noise still running at the end: %s)\n",
                   names[kind + 1], rep, c[0], c[1], 20.0 * 2000 * N, kind < 0 ? "-" : (q == hipErrorNotReady ? "yes" : "no"));
        }
    }
    return 0;
}
