// sinks.hip -- integer event formats of the reference's file writers, packed on device
// (SURVEY.md section 8(f-2)).  The float32 (t,x,y,p) rows stay the primary product; these
// kernels apply the reference's conversion rules so a writer can stream bytes straight from HBM:
//   AEDAT-2.0  v2ecore/output/aedat2_output.py:155-173  (jAER: big-endian int32 address, int32 timestamp)
//   HDF5       v2ecore/emulator.py:955-965              (uint32 [N,4]: t_us, x, y, p with -1 -> 0)
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

__global__ __launch_bounds__(256) void k_pack_aedat2(const float4 *__restrict__ ev, uint2 *__restrict__ out, long long n, int sizex,
                                                     int sizey, int xshift, int yshift, int pshift, int flipx, int flipy,
                                                     long long noise_from, uint32_t noise_bit)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 e = ev[i];
    const int t = (int)(1e6f * e.x);             // (1e6 * events[:,0]).astype(np.int32): float32 product, truncation
    int x = (int)e.y, y = (int)e.z;
    if (flipx) x = (sizex - 1) - x;
    if (flipy) y = (sizey - 1) - y;
    const int p = (int)((e.w + 1.0f) / 2.0f);    // ((p + 1) / 2).astype(np.int32)
    uint32_t a = ((uint32_t)x << xshift) | ((uint32_t)y << yshift) | ((uint32_t)p << pshift);
    if (noise_from >= 0 && i >= noise_from) a |= noise_bit;
    out[i] = make_uint2(bswap32(a), bswap32((uint32_t)t)); // out.byteswap(): big-endian for jAER
}

__global__ __launch_bounds__(256) void k_pack_h5(const float4 *__restrict__ ev, uint4 *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 e = ev[i];
    const float tus = e.x * 1e6f;                // temp_events[:, 0] * 1e6 in float32
    const float p = e.w == -1.0f ? 0.0f : e.w;   // temp_events[temp_events[:, 3] == -1, 3] = 0
    out[i] = make_uint4((uint32_t)tus, (uint32_t)e.y, (uint32_t)e.z, (uint32_t)p);
}

// renderer.py:368-400 + v2e_utils.py:474-486 (hist2d_numba_seq): ON minus OFF event counts per output bin
__global__ __launch_bounds__(256) void k_hist_events(const float4 *__restrict__ ev, long long n, int *__restrict__ diff, int bins_y,
                                                     int bins_x, double y_lo, double x_lo, double delta_y, double delta_x)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 e = ev[i];
    const double fi = ((double)e.z - y_lo) * delta_y; // tracks[0] = y
    const double fj = ((double)e.y - x_lo) * delta_x; // tracks[1] = x
    if (fi >= 0.0 && fi < (double)bins_y && fj >= 0.0 && fj < (double)bins_x)
        atomicAdd(&diff[(int)fi * bins_x + (int)fj], e.w == 1.0f ? 1 : -1); // pol_on = (p == 1), off = not on
}

// currentFrame = clip(currentFrame + (img_on - img_off), -full_scale, +full_scale)   (renderer.py:396-400)
__global__ __launch_bounds__(256) void k_frame_clip_add(double *__restrict__ cur, int *__restrict__ diff, int n, double full_scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = cur[i] + (double)diff[i];
    v = v < -full_scale ? -full_scale : (v > full_scale ? full_scale : v);
    cur[i] = v;
    diff[i] = 0; // ready for the next slice
}

// img = (currentFrame + full_scale_count) / float(full_scale_count * 2)   (renderer.py:245-247): a true float64 division
__global__ __launch_bounds__(256) void k_frame_normalize(const double *__restrict__ cur, double *__restrict__ out, int n, double full_scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (cur[i] + full_scale) / (full_scale * 2);
}

// 8-byte wire format of an event row (include/v2e_amd.h)
__global__ __launch_bounds__(256) void k_pack64(const float4 *__restrict__ ev, unsigned long long *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 e = ev[i];
    out[i] = ((unsigned long long)__float_as_uint(e.x) << 32) | ((unsigned long long)((unsigned)e.y & 0x3FFFu) << 18) |
             ((unsigned long long)((unsigned)e.z & 0x3FFFu) << 4) | (e.w > 0.f ? 1ull : 0ull);
}

__global__ __launch_bounds__(256) void k_unpack64(const unsigned long long *__restrict__ in, float4 *__restrict__ ev, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = in[i];
    ev[i] = make_float4(__uint_as_float((unsigned)(v >> 32)), (float)((unsigned)(v >> 18) & 0x3FFFu), (float)((unsigned)(v >> 4) & 0x3FFFu),
                        (v & 1ull) ? 1.0f : -1.0f);
}

} // namespace

extern "C" {

int v2e_events_pack_aedat2(const float *events, void *out_bytes, int64_t n, int sizex, int sizey, int xshift, int yshift,
                           int pshift, int flipx, int flipy, int64_t noise_from, void *stream)
{
    V2E_REQUIRE((events && out_bytes) || n == 0, "null");
    if (n <= 0) return 0;
    k_pack_aedat2<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)events, (uint2 *)out_bytes, n, sizex, sizey,
                                                                    xshift, yshift, pshift, flipx, flipy, noise_from, 1u << 10);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_events_pack_h5(const float *events, uint32_t *out, int64_t n, void *stream)
{
    V2E_REQUIRE((events && out) || n == 0, "null");
    if (n <= 0) return 0;
    k_pack_h5<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)events, (uint4 *)out, n);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_events_pack64(const float *events, uint64_t *out, int64_t n, void *stream)
{
    V2E_REQUIRE((events && out) || n == 0, "null");
    if (n <= 0) return 0;
    k_pack64<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)events, (unsigned long long *)out, n);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_events_unpack64(const uint64_t *in, float *events, int64_t n, void *stream)
{
    V2E_REQUIRE((events && in) || n == 0, "null");
    if (n <= 0) return 0;
    k_unpack64<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const unsigned long long *)in, (float4 *)events, n);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_events_accumulate_frame(const float *events, int64_t n, double *current_frame, int32_t *scratch_diff, int bins_y,
                                int bins_x, double y_lo, double y_hi, double x_lo, double x_hi, double full_scale, void *stream)
{
    V2E_REQUIRE(current_frame && scratch_diff && bins_y > 0 && bins_x > 0 && (events || n == 0), "bad args");
    hipStream_t s = (hipStream_t)stream;
    const double delta_y = 1 / ((y_hi - y_lo) / bins_y), delta_x = 1 / ((x_hi - x_lo) / bins_x); // v2e_utils.py:478
    if (n > 0)
        k_hist_events<<<v2e_cdiv(n, 256), 256, 0, s>>>((const float4 *)events, n, scratch_diff, bins_y, bins_x, y_lo, x_lo, delta_y, delta_x);
    k_frame_clip_add<<<v2e_cdiv((int64_t)bins_y * bins_x, 256), 256, 0, s>>>(current_frame, scratch_diff, bins_y * bins_x, full_scale);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_frame_normalize(const double *current_frame, double *out, int n, double full_scale, void *stream)
{
    V2E_REQUIRE(current_frame && out && n > 0 && full_scale > 0, "bad args");
    k_frame_normalize<<<v2e_cdiv((int64_t)n, 256), 256, 0, (hipStream_t)stream>>>(current_frame, out, n, full_scale);
    V2E_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
