// The victim of scripts/slp_repro: v2e_amd/csrc/slomo.hip's k_upsample2 (bilinear x2, align_corners=False), verbatim arithmetic,
// compiled three times by run.sh: with the SLP vectoriser (packed float32: v_pk_mul_f32 / v_pk_add_f32), without it, and with it
// plus -mllvm -amdgpu-waitcnt-forcezero.  VICTIM_NAME names the instantiation.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void VICTIM_NAME(const float *__restrict__ x, float *__restrict__ y, long long nc, int h, int w)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int wq = w >> 2;
    const int sh = h >> 1, sw = w >> 1;
    const long long total = nc * (sh + 1) * wq;
    if (q >= total) return;
    const int j = (int)(q % wq);
    const long long r = q / wq;
    const int t = (int)(r % (sh + 1));
    const long long c = r / (sh + 1);
    const int ya = t - 1 < 0 ? 0 : t - 1, yb = t == 0 ? (sh > 1 ? 1 : 0) : (t < sh ? t : sh - 1);
    const float *p0 = x + (c * sh + ya) * (long long)sw, *p1 = x + (c * sh + yb) * (long long)sw;
    const int cm = 2 * j - 1 < 0 ? 0 : 2 * j - 1, c0 = 2 * j, c1 = 2 * j + 1 < sw ? 2 * j + 1 : sw - 1,
              c2 = 2 * j + 2 < sw ? 2 * j + 2 : sw - 1;
    const float a_m = p0[cm], a_0 = p0[c0], a_1 = p0[c1], a_2 = p0[c2];
    const float b_m = p1[cm], b_0 = p1[c0], b_1 = p1[c1], b_2 = p1[c2];
    const float lx0 = j == 0 ? 0.f : 0.75f, hx0 = 1.f - lx0;
    const float t00 = j == 0 ? a_0 : a_m, t01 = j == 0 ? a_1 : a_0, u00 = j == 0 ? b_0 : b_m, u01 = j == 0 ? b_1 : b_0;
    const float ax = hx0 * t00 + lx0 * t01, ay = 0.75f * a_0 + 0.25f * a_1, az = 0.25f * a_0 + 0.75f * a_1, aw = 0.75f * a_1 + 0.25f * a_2;
    const float bx = hx0 * u00 + lx0 * u01, by = 0.75f * b_0 + 0.25f * b_1, bz = 0.25f * b_0 + 0.75f * b_1, bw = 0.75f * b_1 + 0.25f * b_2;
    if (t >= 1) {
        const float ly = ((float)(2 * t - 1) + 0.5f) * 0.5f - 0.5f - (float)(t - 1), hy = 1.f - ly;
        float4 o;
        o.x = hy * ax + ly * bx; o.y = hy * ay + ly * by; o.z = hy * az + ly * bz; o.w = hy * aw + ly * bw;
        *(float4 *)(y + (c * h + 2 * t - 1) * (long long)w + 4 * j) = o;
    }
    if (t < sh) {
        float ry = ((float)(2 * t) + 0.5f) * 0.5f - 0.5f;
        ry = ry < 0.f ? 0.f : ry;
        const float ly = ry - (float)ya, hy = 1.f - ly;
        float4 o;
        o.x = hy * ax + ly * bx; o.y = hy * ay + ly * by; o.z = hy * az + ly * bz; o.w = hy * aw + ly * bw;
        *(float4 *)(y + (c * h + 2 * t) * (long long)w + 4 * j) = o;
    }
}
