// dev tool / evidence for DESIGN.md "what bounds the SloMo kernels": the rate of a bare stream of
// v_mfma_f32_32x32x16_bf16 (the instruction k_conv_s3 issues six of per split-f32 product slab) with nothing else in the
// kernel -- no loads, no LDS, no stores inside the loop -- as a function of
//   * the operand data (zeros / small integers / random bf16 bit patterns: the matrix cores' power draw, and with it the
//     clock the chip sustains, depends on how many operand bits toggle),
//   * the number of independent accumulators per wave (dependent issue vs. back to back),
//   * waves per SIMD.
// Prints bf16 TFLOP/s, the same divided by 6 (the "f32-equivalent" rate of the 6-product split), and the fraction of the
// 2.5 PFLOP/s dense peak.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma scripts/ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(const uint4 *__restrict__ opnd, float *__restrict__ out, int iters)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    // NACC A operands and NACC B operands per lane, loaded once
    bf16x8 a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        uint4 ua = opnd[(size_t)(2 * i) * 64 + (threadIdx.x & 63)], ub = opnd[(size_t)(2 * i + 1) * 64 + (threadIdx.x & 63)];
        a[i] = *(bf16x8 *)&ua;
        b[i] = *(bf16x8 *)&ub;
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; it += 8) { // unrolled: no branch between multiplies
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 1) % NACC], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[t] = s;
}

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

int main(int argc, char **argv)
{
    int ncu = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz (max)\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t nop = 2 * 8 * 64; // up to 8 A + 8 B operand registers of 64 lanes
    uint4 *d_op; float *d_out;
    CK(hipMalloc(&d_op, nop * 16));
    CK(hipMalloc(&d_out, (size_t)ncu * 16 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"zeros", "ones", "normal(0,1)", "random bits"};
    printf("%-12s %5s %10s %12s %14s %10s\n", "operands", "acc", "waves/SIMD", "bf16 TF/s", "f32-equiv TF/s", "of 2.5 PF");
    for (int data = 0; data < 4; ++data) {
        std::vector<uint16_t> h(nop * 8);
        srand(7);
        for (size_t i = 0; i < h.size(); ++i) {
            if (data == 0) h[i] = 0;
            else if (data == 1) h[i] = bf16_of(1.0f);
            else if (data == 2) { float u = 0.f; for (int k = 0; k < 12; ++k) u += (float)((double)rand() / (double)RAND_MAX); h[i] = bf16_of(u - 6.f); }
            else { uint16_t v = (uint16_t)(rand() & 0xFFFF); if (((v >> 7) & 0xFF) == 0xFF) v &= ~0x0080; h[i] = v & 0xBFFF; } // finite, |x| < 2
        }
        CK(hipMemcpy(d_op, h.data(), nop * 16, hipMemcpyHostToDevice));
        for (int nacc = 1; nacc <= 8; nacc *= 2) {
            for (int wps = 1; wps <= 2; ++wps) {
                if (nacc == 8 && wps == 2) continue; // 8 x 16 accumulator registers x 2 waves still fits, but skip: same rate
                const int blocks = ncu * wps; // one 256-thread workgroup = one wave per SIMD of a CU
                auto launch = [&] {
                    switch (nacc) {
                    case 1: k_mfma<1><<<blocks, 256>>>(d_op, d_out, iters); break;
                    case 2: k_mfma<2><<<blocks, 256>>>(d_op, d_out, iters); break;
                    case 4: k_mfma<4><<<blocks, 256>>>(d_op, d_out, iters); break;
                    default: k_mfma<8><<<blocks, 256>>>(d_op, d_out, iters); break;
                    }
                };
                launch(); CK(hipDeviceSynchronize());
                const int reps = 5;
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) launch();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = 2.0 * 32 * 32 * 16 * (double)nacc * iters * 4.0 * blocks * reps;
                const double tf = flop / (ms * 1e-3) / 1e12;
                printf("%-12s %5d %10d %12.0f %14.1f %9.1f%%\n", names[data], nacc, wps, tf, tf / 6.0, 100.0 * tf / 2500.0);
            }
        }
    }
    return 0;
}
