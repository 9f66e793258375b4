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


// ---- 4-byte wire format of an event stream (include/v2e_amd.h: v2e_events_pack32).  The rows of a run come in blocks of
// one time stamp -- all events of one (frame, iteration) carry the same t, emulator.py:793-796, 861-870 -- so t travels once
// per block: run table {float32 bits of t, index of the block's first event} + one word per event (x | y << 11 | p << 21).
constexpr int P32_BLOCK = 1024;

__device__ __forceinline__ bool p32_starts_run(const float4 *__restrict__ ev, int64_t i)
{
    return i == 0 || __float_as_uint(ev[i].x) != __float_as_uint(ev[i - 1].x);
}

// pass 1: payload words + run starts per block of 1024 events
__global__ __launch_bounds__(P32_BLOCK) void k_pack32_words(const float4 *__restrict__ ev, int64_t n, uint32_t *__restrict__ payload,
                                                            uint32_t *__restrict__ blk_runs, uint32_t *__restrict__ overflow)
{
    __shared__ uint32_t s_cnt[P32_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * P32_BLOCK + threadIdx.x;
    bool start = false;
    if (i < n) {
        const float4 e = ev[i];
        const uint32_t x = (uint32_t)e.y, y = (uint32_t)e.z;
        if (x >= 2048u || y >= 1024u) atomicOr(overflow, 1u); // does not fit 11 + 10 bits: the caller falls back to pack64
        payload[i] = (x & 0x7FFu) | ((y & 0x3FFu) << 11) | (e.w > 0.f ? 1u << 21 : 0u);
        start = p32_starts_run(ev, i);
    }
    const unsigned long long b = __ballot(start);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < P32_BLOCK / 64; ++w) t += s_cnt[w];
        blk_runs[blockIdx.x] = t;
    }
}

// pass 2: exclusive scan of the per-block run counts (one workgroup; nblk <= 2^22), total to runs[0]
__global__ __launch_bounds__(1024) void k_pack32_scan(uint32_t *__restrict__ blk_runs, int nblk, unsigned long long *__restrict__ runs,
                                                      int64_t cap_runs, uint32_t *__restrict__ overflow)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nblk + 1023) / 1024;
    uint32_t sum = 0;
    for (int k = 0; k < per; ++k) { const int j = tid * per + k; if (j < nblk) sum += blk_runs[j]; }
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int k = 0; k < per; ++k) {
        const int j = tid * per + k;
        if (j < nblk) { const uint32_t c = blk_runs[j]; blk_runs[j] = run; run += c; }
    }
    if (tid == 1023) {
        // runs[0]: the number of blocks IN THE TABLE (never more than it holds: a receiver searches runs[1 .. 1 + R)), and in
        // bits 62 / 63 what went wrong -- a coordinate that does not fit, a table that was too small -- so that every rank
        // that receives the table sees it, not only the sender
        const uint32_t total = s_part[1023];
        uint32_t fl = *overflow & 1u; // k_pack32_words ran before this kernel on the same stream
        if ((int64_t)total > cap_runs) { fl |= 2u; atomicOr(overflow, 2u); }
        const unsigned long long kept = (int64_t)total > cap_runs ? (unsigned long long)cap_runs : (unsigned long long)total;
        runs[0] = kept | ((unsigned long long)fl << 62);
    }
}

// pass 3: run table entries (t bits << 32 | first event index), in order
__global__ __launch_bounds__(P32_BLOCK) void k_pack32_runs(const float4 *__restrict__ ev, int64_t n, const uint32_t *__restrict__ blk_base,
                                                           unsigned long long *__restrict__ runs, int64_t cap_runs)
{
    __shared__ uint32_t s_cnt[P32_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * P32_BLOCK + threadIdx.x;
    const bool start = i < n && p32_starts_run(ev, i);
    const unsigned long long b = __ballot(start);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = blk_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += s_cnt[w];
    if (start) {
        const uint32_t r = before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if ((int64_t)r < cap_runs) runs[1 + r] = ((unsigned long long)__float_as_uint(ev[i].x) << 32) | (unsigned long long)(uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_unpack32(const uint32_t *__restrict__ payload, int64_t n, const unsigned long long *__restrict__ runs,
                                                  float4 *__restrict__ ev)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int nr = (int)(runs[0] & 0xFFFFFFFFull); // bits 62 / 63: the sender's overflow flags (the host refuses such a table)
    if (nr <= 0) return;
    int lo = 0, hi = nr; // last run whose first event index is <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)(uint32_t)runs[1 + mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t w = payload[i];
    ev[i] = make_float4(__uint_as_float((uint32_t)(runs[1 + lo] >> 32)), (float)(w & 0x7FFu), (float)((w >> 11) & 0x3FFu),
                        (w >> 21) & 1u ? 1.0f : -1.0f);
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

int v2e_events_pack32(const float *events, int64_t n, uint32_t *payload, uint64_t *runs, int64_t cap_runs, uint32_t *scratch, void *stream)
{
    V2E_REQUIRE(runs && scratch && cap_runs >= 0 && (n == 0 || (events && payload)), "null");
    V2E_REQUIRE(n < ((int64_t)1 << 32), "too many events for 32-bit run starts");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)((n + P32_BLOCK - 1) / P32_BLOCK);
    // scratch: [0] overflow flags, [1 .. 1 + nblk) per-block run counts / bases
    V2E_HIP(hipMemsetAsync(scratch, 0, sizeof(uint32_t), s));
    if (n == 0) { V2E_HIP(hipMemsetAsync(runs, 0, sizeof(uint64_t), s)); return 0; }
    k_pack32_words<<<nblk, P32_BLOCK, 0, s>>>((const float4 *)events, n, payload, scratch + 1, scratch);
    k_pack32_scan<<<1, 1024, 0, s>>>(scratch + 1, nblk, (unsigned long long *)runs, cap_runs, scratch);
    k_pack32_runs<<<nblk, P32_BLOCK, 0, s>>>((const float4 *)events, n, scratch + 1, (unsigned long long *)runs, cap_runs);
    V2E_HIP(hipGetLastError());
    return 0;
}

int64_t v2e_events_pack32_scratch_words(int64_t n) { return 2 + (n + P32_BLOCK - 1) / P32_BLOCK; }

int v2e_events_unpack32(const uint32_t *payload, int64_t n, const uint64_t *runs, float *events, void *stream)
{
    V2E_REQUIRE(runs && (n == 0 || (payload && events)), "null");
    if (n <= 0) return 0;
    k_unpack32<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(payload, n, (const unsigned long long *)runs, (float4 *)events);
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
