// Launchers for scripts/slp_repro (see run.sh / repro.py): three builds of the SAME bilinear-x2 kernel and a bare matrix-core loop.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern "C" __global__ void up_slp(const float *, float *, long long, int, int);
extern "C" __global__ void up_noslp(const float *, float *, long long, int, int);
extern "C" __global__ void up_slp_wait0(const float *, float *, long long, int, int);
// operands with toggling bits (a loop on constant operands draws far less power than a convolution does)
__global__ __launch_bounds__(256) void k_noise_bf16(float *out, int iters, uint32_t seed)
{
    f16v acc = {0};
    union { bf16x8 v; uint32_t u[4]; } a, b;
    uint32_t s = seed + threadIdx.x * 2654435761u + blockIdx.x;
    for (int e = 0; e < 4; ++e) { s = s * 1664525u + 1013904223u; a.u[e] = (s & 0x807F807Fu) | 0x3F003F00u; s = s * 1664525u + 1013904223u; b.u[e] = (s & 0x807F807Fu) | 0x3F003F00u; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
        a.u[it & 3] ^= 0x00550055u; b.u[(it + 1) & 3] ^= 0x002A002Au;
    }
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += acc[e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
extern "C" int slp_launch_victim(int which, const float *x, float *y, long long nc, int h, int w, hipStream_t st)
{
    const long long total = nc * ((h >> 1) + 1) * (w >> 2);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (which == 0) up_slp<<<g, b, 0, st>>>(x, y, nc, h, w);
    else if (which == 1) up_noslp<<<g, b, 0, st>>>(x, y, nc, h, w);
    else up_slp_wait0<<<g, b, 0, st>>>(x, y, nc, h, w);
    return (int)hipGetLastError();
}
extern "C" int slp_launch_noise(float *out, int blocks, int iters, hipStream_t st)
{
    k_noise_bf16<<<blocks, 256, 0, st>>>(out, iters, 12345u);
    return (int)hipGetLastError();
}

// ---- experiment 2: single-instruction noises and single-instruction victims
// noise kinds: 0 v_cvt_pk_bf16_f32, 1 v_cvt_pk_f16_f32 (v_cvt_pkrtz), 2 ds_read_b128, 3 v_cvt_f32_f16 sdwa, 4 ds_bpermute_b32, 5 plain v_mul_f32
template <int KIND> __global__ __launch_bounds__(256) void k_noise_one(float *out, int iters)
{
    __shared__ float4 s_buf[256];
    float a = 1.0f + threadIdx.x * 0.001f, b = 0.5f + blockIdx.x * 0.002f;
    uint32_t r = 0;
    float4 q = make_float4(a, b, a, b);
    s_buf[threadIdx.x] = q;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        if (KIND == 2) { asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((uint32_t)(threadIdx.x * 16)) : "memory"); }
        if (KIND == 3) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a) : "v"(r | 0x3c003c00u));
        if (KIND == 4) asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)((threadIdx.x * 4 + 4) & 255)), "v"(r + 1u) : "memory");
        if (KIND == 5) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + (float)r + q.x;
}
// victim forms: 0 v_pk_mul_f32 plain, 1 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], 2 v_pk_add_f32 plain,
// 3 v_pk_mul_f32 with an SGPR-pair source, 4 v_pk_fma_f32 plain, 5 the dependent pair of k_upsample2: v_sub_f32 writes the high half, the
// swapped pk_mul reads it in the next-but-one instruction.  Every lane compares with v_mul_f32 / v_add_f32 on the same registers.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <int FORM> __global__ __launch_bounds__(256) void k_victim_one(const float *__restrict__ x, int iters, unsigned *mism)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    f2 a = {x[i], x[i + 1]}, b = {x[i + 2], x[i + 3]};
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        f2 r;
        float e0, e1;
        if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(b)); e0 = s_mul(a.x, b.x); e1 = s_mul(a.y, b.y); }
        if (FORM == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = s_mul(a.x, b.y); e1 = s_mul(a.y, b.x); }
        if (FORM == 2) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(b)); e0 = s_add(a.x, b.x); e1 = s_add(a.y, b.y); }
        if (FORM == 3) { asm volatile("s_mov_b32 s20, 0x3e800000\n\ts_mov_b32 s21, 0x3f400000\n\tv_pk_mul_f32 %0, %1, s[20:21]" : "=&v"(r) : "v"(a) : "s20", "s21"); e0 = s_mul(a.x, 0.25f); e1 = s_mul(a.y, 0.75f); }
        if (FORM == 4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=&v"(r) : "v"(a), "v"(b)); e0 = s_fma(a.x, b.x, a.x); e1 = s_fma(a.y, b.y, a.y); }
        if (FORM == 5) { // as in k_upsample2: v41 written right before; the swapped pk_mul routes it to the LOW result
            asm volatile("v_sub_f32 v40, %1, %2\n\tv_sub_f32 v41, 1.0, v40\n\tv_pk_mul_f32 %0, %3, v[40:41]\n\tv_pk_mul_f32 %0, %3, v[40:41] op_sel:[0,1] op_sel_hi:[1,0]"
                         : "=&v"(r) : "v"(a.x), "v"(b.x), "v"(a) : "v40", "v41");
            const float ly = s_sub(a.x, b.x), hy = s_sub(1.0f, ly);
            e0 = s_mul(a.x, hy); e1 = s_mul(a.y, ly);
        }
        bad_lo += __float_as_uint(r.x) != __float_as_uint(e0);
        bad_hi += __float_as_uint(r.y) != __float_as_uint(e1);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    if (bad_lo) atomicAdd(mism, bad_lo);
    if (bad_hi) atomicAdd(mism + 1, bad_hi);
}
extern "C" int slp_launch_noise_one(int kind, float *out, int blocks, int iters, hipStream_t st)
{
    switch (kind) {
    case 0: k_noise_one<0><<<blocks, 256, 0, st>>>(out, iters); break;
    case 1: k_noise_one<1><<<blocks, 256, 0, st>>>(out, iters); break;
    case 2: k_noise_one<2><<<blocks, 256, 0, st>>>(out, iters); break;
    case 3: k_noise_one<3><<<blocks, 256, 0, st>>>(out, iters); break;
    case 4: k_noise_one<4><<<blocks, 256, 0, st>>>(out, iters); break;
    default: k_noise_one<5><<<blocks, 256, 0, st>>>(out, iters); break;
    }
    return (int)hipGetLastError();
}
extern "C" int slp_launch_victim_one(int form, const float *x, int blocks, int iters, unsigned *mism, hipStream_t st)
{
    switch (form) {
    case 0: k_victim_one<0><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    case 1: k_victim_one<1><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    case 2: k_victim_one<2><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    case 3: k_victim_one<3><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    case 4: k_victim_one<4><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    default: k_victim_one<5><<<blocks, 256, 0, st>>>(x, iters, mism); break;
    }
    return (int)hipGetLastError();
}

// ---- experiment 3: which PART of the split-operand convolution is the noise (victim: the swapped v_pk_mul_f32), and what the wrong low half IS
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// kinds: 0 bf16 MFMA, four independent accumulators (the matrix pipe never idles)   1 bf16 MFMA + v_cvt_pk_bf16_f32 between them
//        2 bf16 MFMA fed by ds_read_b128 (operands from LDS every iteration)         3 bf16 MFMA + v_accvgpr_write / read traffic
//        4 ds_write_b128 + s_barrier + ds_read_b128, no MFMA                         5 f32 MFMA 32x32x2, four accumulators (control)
//        6 f16 MFMA 32x32x16, four accumulators                                      7 bf16 MFMA + global_load_dword stream
template <int KIND> __global__ __launch_bounds__(256) void k_noise_mix(float *out, const float *gsrc, int iters, uint32_t seed)
{
    __shared__ u4v s_buf[512];
    f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    union { bf16x8 v; f16x8 h; uint32_t u[4]; u4v q; } a, b;
    uint32_t s = seed + threadIdx.x * 2654435761u + blockIdx.x;
    for (int e = 0; e < 4; ++e) { s = s * 1664525u + 1013904223u; a.u[e] = (s & 0x807F807Fu) | 0x3F003F00u; s = s * 1664525u + 1013904223u; b.u[e] = (s & 0x807F807Fu) | 0x3F003F00u; }
    s_buf[threadIdx.x] = a.q; s_buf[256 + threadIdx.x] = b.q;
    __syncthreads();
    float fa = 1.0f + threadIdx.x * 0.01f, fb = 0.75f, g = 0.f;
    uint32_t r = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 5) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fa, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fb, acc3, 0, 0, 0);
        } else if (KIND == 6) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, a.h, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, b.h, acc3, 0, 0, 0);
        } else if (KIND != 4) {
            if (KIND == 2) {
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=v"(a.q), "=v"(b.q) : "v"((uint32_t)(((threadIdx.x + it) & 255) * 16)) : "memory");
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc0, 0, 0, 0);
            if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(fa), "v"(fb));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v, a.v, acc1, 0, 0, 0);
            if (KIND == 1) { a.u[it & 3] ^= r & 0x00010001u; }
            if (KIND == 3) { asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_write_b32 a1, %1\n\ts_nop 2\n\tv_accvgpr_read_b32 %0, a0" : "=v"(r) : "v"(r + 1u) : "a0", "a1"); }
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, a.v, acc2, 0, 0, 0);
            if (KIND == 7) g += gsrc[(size_t)((blockIdx.x * 256 + threadIdx.x + it * 4099) & 0xFFFFF)];
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v, b.v, acc3, 0, 0, 0);
        } else {
            asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"((uint32_t)(threadIdx.x * 16)), "v"(a.q) : "memory");
            __syncthreads();
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a.q) : "v"((uint32_t)(((threadIdx.x + 1) & 255) * 16)) : "memory");
            __syncthreads();
        }
        a.u[it & 3] ^= 0x00550055u; b.u[(it + 1) & 3] ^= 0x002A002Au;
    }
    float t = g + (float)r + __uint_as_float(a.u[0]);
    for (int e = 0; e < 16; ++e) t += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
// the swapped multiply again, keeping what the first wrong low half of every lane was: dump[lane] = {a.x, a.y, b.x, b.y, r.x, r.y, iteration}
__global__ __launch_bounds__(256) void k_victim_dump(const float *__restrict__ x, int iters, float *dump, unsigned *mism)
{
    const int tidg = blockIdx.x * 256 + threadIdx.x, i = tidg * 4;
    f2 a = {x[i], x[i + 1]}, b = {x[i + 2], x[i + 3]};
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 r;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
        const float e0 = s_mul(a.x, b.y), e1 = s_mul(a.y, b.x);
        if (__float_as_uint(r.x) != __float_as_uint(e0) || __float_as_uint(r.y) != __float_as_uint(e1)) {
            if (!bad) { float *d = dump + (size_t)tidg * 8; d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y; d[4] = r.x; d[5] = r.y; d[6] = (float)it; d[7] = 1.f; }
            ++bad;
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    if (bad) atomicAdd(mism, bad);
}
extern "C" int slp_launch_noise_mix(int kind, float *out, const float *gsrc, int blocks, int iters, hipStream_t st)
{
    switch (kind) {
    case 0: k_noise_mix<0><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 1: k_noise_mix<1><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 2: k_noise_mix<2><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 3: k_noise_mix<3><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 4: k_noise_mix<4><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 5: k_noise_mix<5><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    case 6: k_noise_mix<6><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    default: k_noise_mix<7><<<blocks, 256, 0, st>>>(out, gsrc, iters, 77u); break;
    }
    return (int)hipGetLastError();
}
extern "C" int slp_launch_victim_dump(const float *x, int blocks, int iters, float *dump, unsigned *mism, hipStream_t st)
{
    k_victim_dump<<<blocks, 256, 0, st>>>(x, iters, dump, mism);
    return (int)hipGetLastError();
}

// ---- dev tool for scripts/step_stamps.py: a device time stamp (100 MHz constant clock) from a kernel of its own on the caller's stream
__global__ void k_stamp(unsigned long long *out, int idx) { out[idx] = wall_clock64(); }
extern "C" int slp_stamp(unsigned long long *out, int idx, hipStream_t st) { k_stamp<<<1, 1, 0, st>>>(out, idx); return (int)hipGetLastError(); }
__global__ void k_stamp_seq(unsigned long long *buf) { const unsigned long long i = atomicAdd(buf, 1ull); buf[1 + i] = wall_clock64(); }
extern "C" int slp_stamp_seq(unsigned long long *buf, hipStream_t st) { k_stamp_seq<<<1, 1, 0, st>>>(buf); return (int)hipGetLastError(); }
