// Minimal reproduction of the round-4 "concurrency corruption" (profiles/r05_concurrency_rootcause.txt), MI355X / gfx950:
//   victim  one wave-level instruction:  v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]   (low result = A.lo * B.hi, high = A.hi * B.lo)
//   noise   another kernel, on another stream, issuing INDEPENDENT v_mfma_f32_32x32x16_bf16 (or _f16) back to back
// While the noise runs, the LOW result of the swapped packed multiply comes out as exactly 0.0f for whole waves (about 1 result in 1000);
// the high result, the un-swapped v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32 and the scalar v_mul_f32 are always right; with no noise, with a
// float32 MFMA (32x32x2) noise or with a DEPENDENT chain of bf16 MFMAs there is not one wrong result.  No memory is shared; the victim
// compares every packed result with v_mul_f32 on the same registers in the same lane.  The only producer of that instruction form in
// this repository was the compiler's SLP vectoriser (k_upsample2): hence -fno-slp-vectorize in v2e_amd/csrc/Makefile.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o pk_repro scripts/pk_opsel_mfma_repro.hip && ./pk_repro
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int FORM> __global__ __launch_bounds__(256) void k_victim(const float *__restrict__ x, int iters, unsigned *wrong)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    f2 a = {x[i], x[i + 1]}, b = {x[i + 2], x[i + 3]}, r;
    unsigned lo = 0, hi = 0, zero = 0;
    for (int it = 0; it < iters; ++it) {
        float e0, e1;
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));  // swapped
        if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(b));                                // plain
        if (FORM == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b));  // the other swap
        if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(r) : "v"(a), "v"(b));  // broadcast B.hi
        if (FORM == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));  // swapped add
        const float bl = (FORM == 0 || FORM == 3 || FORM == 4) ? b.y : b.x, bh = (FORM == 3 || FORM == 1) ? b.y : (FORM == 2 ? b.y : b.x);
        const float al = FORM == 2 ? a.y : a.x, ah = FORM == 2 ? a.x : a.y;
        if (FORM == 4) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(al), "v"(bl)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(ah), "v"(bh)); }
        else { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(al), "v"(bl)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(ah), "v"(bh)); }
        lo += __float_as_uint(r.x) != __float_as_uint(e0);
        hi += __float_as_uint(r.y) != __float_as_uint(e1);
        zero += r.x == 0.f && e0 != 0.f;
        asm volatile("" : "+v"(a), "+v"(b));
    }
    if (lo) atomicAdd(wrong, lo);
    if (hi) atomicAdd(wrong + 1, hi);
    if (zero) atomicAdd(wrong + 2, zero);
}
template <int KIND> __global__ __launch_bounds__(256) void k_noise(float *out, int iters) // 0: four independent bf16 MFMAs, 1: one dependent chain, 2: f32 MFMAs
{
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    union { bf16x8 v; uint32_t u[4]; } a, b;
    for (int e = 0; e < 4; ++e) { a.u[e] = 0x3F803F80u + threadIdx.x * 0x00010001u + e; b.u[e] = 0x3F003F40u + blockIdx.x * 0x00010001u + e; }
    const float fa = 1.f + threadIdx.x * .01f, fb = .75f;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c1, 0, 0, 0);
                         c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fa, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fb, c3, 0, 0, 0); continue; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c0, 0, 0, 0);
        if (KIND == 0) { c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v, a.v, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, a.v, c2, 0, 0, 0);
                         c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v, b.v, c3, 0, 0, 0); }
        a.u[it & 3] ^= 0x00550055u;
    }
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main()
{
    const int NB = 2048, N = NB * 256 * 4, ITERS = 3000;
    std::vector<float> hx(N);
    uint32_t st = 12345u;
    for (int i = 0; i < N; ++i) { st = st * 1664525u + 1013904223u; hx[i] = 0.25f + (float)(st >> 8) * (1.0f / 16777216.0f); }
    float *dx, *dn; unsigned *dw;
    CK(hipMalloc(&dx, N * 4)); CK(hipMalloc(&dn, 1024 * 256 * 4)); CK(hipMalloc(&dw, 12));
    CK(hipMemcpy(dx, hx.data(), N * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char *forms[5] = {"v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]",
                            "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]"};
    const char *noises[4] = {"none", "4 independent bf16 MFMA 32x32x16", "dependent bf16 MFMA chain", "4 independent f32 MFMA 32x32x2"};
    int fails = 0;
    for (int nz = 0; nz < 4; ++nz)
        for (int f = 0; f < 5; ++f)
            for (int rep = 0; rep < (f == 0 && nz == 1 ? 10 : 2); ++rep) {
                CK(hipMemset(dw, 0, 12)); CK(hipDeviceSynchronize());
                if (nz == 1) k_noise<0><<<1024, 256, 0, s2>>>(dn, 60000);
                if (nz == 2) k_noise<1><<<1024, 256, 0, s2>>>(dn, 200000);
                if (nz == 3) k_noise<2><<<1024, 256, 0, s2>>>(dn, 30000);
                for (int l = 0; l < 6; ++l) {
                    if (f == 0) k_victim<0><<<NB, 256, 0, s1>>>(dx, ITERS, dw); if (f == 1) k_victim<1><<<NB, 256, 0, s1>>>(dx, ITERS, dw);
                    if (f == 2) k_victim<2><<<NB, 256, 0, s1>>>(dx, ITERS, dw); if (f == 3) k_victim<3><<<NB, 256, 0, s1>>>(dx, ITERS, dw);
                    if (f == 4) k_victim<4><<<NB, 256, 0, s1>>>(dx, ITERS, dw);
                }
                CK(hipStreamSynchronize(s1));
                const bool overlapped = nz == 0 || hipStreamQuery(s2) == hipErrorNotReady;
                CK(hipDeviceSynchronize());
                unsigned w[3];
                CK(hipMemcpy(w, dw, 12, hipMemcpyDeviceToHost));
                if (f == 0 && nz == 1) fails += w[0] != 0;
                printf("noise %-34s victim %-44s rep %d: wrong low %9u (of them 0.0f: %9u), wrong high %u, of %.3g%s\n", noises[nz], forms[f], rep, w[0], w[2], w[1],
                       6.0 * ITERS * NB * 256, overlapped ? "" : "  [noise ended before the victim]");
            }
    printf("swapped v_pk_mul_f32 beside independent bf16 MFMAs: wrong in %d runs of 10\n", fails);
    return 0;
}
