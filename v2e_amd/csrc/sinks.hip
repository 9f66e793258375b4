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

} // extern "C"
