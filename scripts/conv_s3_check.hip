// conv_s3_check -- dev harness (no torch): the split-bf16 convolution (slomo_s3.h) against the f32-MFMA kernel and
// against double-precision sums at sampled outputs, with timings.  Links libv2e_amd.so through its C ABI.
//   hipcc --offload-arch=gfx950 -O2 scripts/conv_s3_check.hip -Iinclude -Lv2e_amd/csrc -lv2e_amd -Wl,-rpath,'$ORIGIN/../v2e_amd/csrc' -o scripts/conv_s3_check
//   scripts/conv_s3_check [ks cin cout n h w]        (no arguments: the layer shapes of the 80-sample interpolation UNet)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "v2e_amd.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)
#define CV(e) do { int _r = (e); if (_r) { printf("v2e error %d (%s) at %d\n", _r, v2e_last_error(), __LINE__); exit(3); } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static float frand() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng >> 40) & 0xFFFFFF) / 16777216.0f - 0.5f; }

struct Case { int ks, cin, cout, n, h, w; };

static int run_case(const Case &c, int reps)
{
    const int kk = c.ks * c.ks, pad = c.ks / 2;
    const size_t nx = (size_t)c.n * c.cin * c.h * c.w, ny = (size_t)c.n * c.cout * c.h * c.w, nw = (size_t)c.cout * c.cin * kk;
    std::vector<float> hx(nx), hw_(nw), hb(c.cout), y32(ny), y3(ny);
    for (auto &v : hx) v = 4.f * frand();
    const float ws = 2.f / sqrtf((float)c.cin * kk);
    for (auto &v : hw_) v = ws * 2.f * frand();
    for (auto &v : hb) v = frand();
    float *dx, *dw, *dwp, *db, *dy;
    void *dw3;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dwp, nw * 4)); CK(hipMalloc(&dw3, (size_t)c.cout * ((c.cin + 15) / 16 * 16) * kk * 6));
    CK(hipMalloc(&db, c.cout * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw_.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), c.cout * 4, hipMemcpyHostToDevice));
    CV(v2e_pack_conv_weight(dw, dwp, c.cout, c.cin, c.ks, nullptr));
    CV(v2e_pack_conv_weight_s3(dw, dw3, c.cout, c.cin, c.ks, nullptr));
    CK(hipDeviceSynchronize());
    v2e_conv_desc d;
    d.weight = dwp; d.bias = db; d.cin = c.cin; d.cout = c.cout; d.ksize = c.ks; d.split_kind = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[3] = {0, 0, 0};
    std::vector<float> y3p;
    void *dxs = nullptr;
    const bool presplit = getenv("S3_PRESPLIT") && c.ks == 3 && c.cin % 16 == 0 && c.w % 8 == 0;
    if (presplit) {
        CK(hipMalloc(&dxs, nx * 6));
        CV(v2e_split3_nchw(dx, dxs, c.n, c.cin, c.h, c.w, nullptr));
        y3p.resize(ny);
    }
    for (int mode = 0; mode < (presplit ? 3 : 2); ++mode) {
        d.weight_s3 = mode ? dw3 : nullptr;
        if (mode == 2) {
            CK(hipMemset(dy, 0xFF, ny * 4));
            for (int i = 0; i < 2; ++i) CV(v2e_conv2d_lrelu((const float *)dxs, c.cin, nullptr, 0, 3, &d, dy, c.n, c.h, c.w, nullptr));
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) CV(v2e_conv2d_lrelu((const float *)dxs, c.cin, nullptr, 0, 3, &d, dy, c.n, c.h, c.w, nullptr));
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[2], e0, e1));
            ms[2] /= reps;
            CK(hipMemcpy(y3p.data(), dy, ny * 4, hipMemcpyDeviceToHost));
            continue;
        }
        CK(hipMemset(dy, 0xFF, ny * 4));
        for (int i = 0; i < 2; ++i) CV(v2e_conv2d_lrelu(dx, c.cin, nullptr, 0, 0, &d, dy, c.n, c.h, c.w, nullptr));
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) CV(v2e_conv2d_lrelu(dx, c.cin, nullptr, 0, 0, &d, dy, c.n, c.h, c.w, nullptr));
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
        ms[mode] /= reps;
        CK(hipMemcpy(mode ? y3.data() : y32.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    }
    double dmax = 0, dscaled = 0;
    size_t nbad = 0;
    for (size_t i = 0; i < ny; ++i) {
        const double dd = fabs((double)y3[i] - (double)y32[i]);
        if (!(dd == dd)) { ++nbad; continue; }
        dmax = fmax(dmax, dd);
        dscaled = fmax(dscaled, dd / fmax(1.0, fabs((double)y32[i])));
    }
    // double-precision sums at sampled outputs
    double e32 = 0, e3 = 0, ymax = 0;
    for (int s = 0; s < 3000; ++s) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const size_t i = (size_t)((rng >> 11) % ny);
        const int ox = (int)(i % c.w), oy = (int)((i / c.w) % c.h), co = (int)((i / ((size_t)c.w * c.h)) % c.cout), n = (int)(i / ((size_t)c.w * c.h * c.cout));
        double acc = hb[co];
        for (int ci = 0; ci < c.cin; ++ci)
            for (int ky = 0; ky < c.ks; ++ky)
                for (int kx = 0; kx < c.ks; ++kx) {
                    const int gy = oy + ky - pad, gx = ox + kx - pad;
                    if (gy < 0 || gy >= c.h || gx < 0 || gx >= c.w) continue;
                    acc += (double)hw_[((size_t)co * c.cin + ci) * kk + ky * c.ks + kx] * (double)hx[(((size_t)n * c.cin + ci) * c.h + gy) * c.w + gx];
                }
        if (acc < 0) acc *= (double)0.1f;
        ymax = fmax(ymax, fabs(acc));
        e32 = fmax(e32, fabs(acc - y32[i]) / fmax(1.0, fabs(acc)));
        e3 = fmax(e3, fabs(acc - y3[i]) / fmax(1.0, fabs(acc)));
    }
    const double flop = 2.0 * c.n * c.h * c.w * (double)c.cout * c.cin * kk;
    printf("k%d %4d->%4d n%-3d %3dx%-3d  f32 %8.1f us %6.1f TF | s3 %8.1f us %6.1f TF (x%.2f) | s3-f32 max %.2e scaled %.2e nan %zu | vs f64: f32 %.2e  s3 %.2e  (|y|max %.1f)\n",
           c.ks, c.cin, c.cout, c.n, c.h, c.w, ms[0] * 1e3, flop / ms[0] / 1e9, ms[1] * 1e3, flop / ms[1] / 1e9, ms[0] / ms[1], dmax, dscaled, nbad,
           e32, e3, ymax);
    if (presplit) {
        size_t diff = 0;
        for (size_t i = 0; i < ny; ++i) diff += y3p[i] != y3[i];
        printf("      pre-split input: %8.1f us %6.1f TF (x%.2f vs in-kernel split), %zu outputs differ\n", ms[2] * 1e3, flop / ms[2] / 1e9, ms[1] / ms[2], diff);
        CK(hipFree(dxs));
    }
    fflush(stdout);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dwp)); CK(hipFree(dw3)); CK(hipFree(db)); CK(hipFree(dy));
    return (nbad == 0 && e3 < 1e-5 && dscaled < 5e-5) ? 0 : 1; // both kernels against the double-precision sums
}

extern "C" int v2e_slomo_debug_s3p_timeline(unsigned long long *out, int n_pairs); // dev entry point, not in the header

// S3P_TIMELINE=1: k_conv_s3p's workgroup 0, step by step (shader clocks per step, and the shader clock rate itself from the
// 100 MHz wall clock read at the same points)
static void print_s3p_timeline()
{
    std::vector<unsigned long long> tl(2 * 512);
    if (v2e_slomo_debug_s3p_timeline(tl.data(), 512)) return;
    int n = 0;
    while (n < 511 && tl[2 * n]) ++n;
    if (n < 3) { printf("   (no k_conv_s3p timeline)\n"); return; }
    const double clk = (double)(tl[2 * (n - 1)] - tl[0]), wall = (double)(tl[2 * (n - 1) + 1] - tl[1]);
    printf("   k_conv_s3p workgroup 0: %d steps, %.0f shader clocks per step, shader clock %.0f MHz (per step:", n, clk / (n - 1), clk / wall * 100.0);
    for (int i = 1; i < n && i < 14; ++i) printf(" %llu", tl[2 * i] - tl[2 * (i - 1)]);
    printf(" ...)\n");
}

int main(int argc, char **argv)
{
    int bad = 0;
    if (getenv("S3P_TIMELINE")) v2e_slomo_debug_s3p_timeline(nullptr, 0);
    if (argc >= 7) {
        Case c = {atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6])};
        const int r = run_case(c, argc > 7 ? atoi(argv[7]) : 5);
        if (getenv("S3P_TIMELINE")) print_s3p_timeline();
        return r;
    }
    const int full = getenv("S3_FULL") ? atoi(getenv("S3_FULL")) : 0;
    const int N = getenv("S3_N") ? atoi(getenv("S3_N")) : (full ? 80 : 8);
    const Case cases[] = {
        {3, 16, 32, 3, 40, 72},  {3, 32, 64, 2, 24, 32}, // ragged / masked tiles
        {3, 64, 32, N, 256, 320}, {3, 128, 64, N, 128, 160}, {3, 256, 128, N, 64, 80}, {3, 512, 256, N, 32, 40},
        {3, 512, 512, N, 16, 20}, {5, 32, 64, N, 128, 160},  {5, 64, 64, N, 128, 160}, {7, 32, 32, N, 256, 320}, {7, 12, 32, N, 256, 320},
    };
    for (const Case &c : cases) bad += run_case(c, full ? 3 : 5);
    printf(bad ? "MISMATCH in %d cases\n" : "all cases within 1e-5 (%d)\n", bad);
    return bad ? 1 : 0;
}
