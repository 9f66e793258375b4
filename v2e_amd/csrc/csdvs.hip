// csdvs.hip -- the centre-surround DVS's horizontal-cell diffuser (EventEmulator._update_csdvs,
// v2ecore/emulator.py:1061-1124).
//
// What the reference computes per frame: up to num_steps explicit Euler steps of
//     change = alpha_p (p - h) + alpha_h conv2d(ReplicationPad2d(1)(h.float()), [[0,1,0],[1,-4,1],[0,1,0]])
//     h     += change
// on the surround plane h, driven by the low-passed photoreceptor plane p, stopping after the first step whose
// max |change| is <= 1e-5 (emulator.py:1105-1121; the host reads max_change back after EVERY step).
//
// Here a step is one launch over the plane (5-point stencil, h ping-ponged between the plane and a scratch plane, the
// step's max |change| collected by atomicMax on its float64 bit pattern), and the stop rule is evaluated on the device:
// step s runs only if step s - 1's maximum was above the threshold, so a chunk of CHUNK steps is enqueued without the host
// looking, the chunk's maxima come back in one copy, and the host counts the steps that ran.  A diffuser that needs 200
// steps a frame costs 7 synchronisations instead of 200.
//
// Arithmetic, tensor type by tensor type as torch evaluates it (R = the state planes' type: float64 with a photoreceptor
// cutoff, else float32):  diff = p - h in R;  p_term = alpha_p * diff in R (a Python scalar takes the tensor's type);
// h_conv in float32 from h rounded to float32;  h_term = alpha_h * h_conv in float32;  change = p_term + h_term in R.
// The float32 sum inside conv2d is the one thing torch does not pin down: its order is the convolution backend's, and
// on the reference's CPU path it differs with the plane size (measured, tests/golden/make_golden_csdvs.py: planes of
// 200 x 200 and larger, DAVIS346 included, sum in kernel order ((((t + l) - 4 c) + r) + b); 40 x 48 ... 128 x 128 sum
// ((t + l) + ((b + r) - 4 c))).  This kernel and the oracle's restatement (oracle/emu_oracle.c) fix the kernel order.
#include "common.h"

namespace {

constexpr int CHUNK = 32;

template <typename R>
__global__ __launch_bounds__(256) void k_cs_step(const R *__restrict__ p, const R *__restrict__ h_in, R *__restrict__ h_out,
                                                 int H, int W, R alpha_p, float alpha_h, double thr,
                                                 const unsigned long long *__restrict__ prev_max,
                                                 unsigned long long *__restrict__ cur_max)
{
    __shared__ unsigned long long smax[4];
    if (__longlong_as_double((long long)*prev_max) <= thr) return; // the loop ended at an earlier step (emulator.py:1107)
    const int i = blockIdx.x * 256 + threadIdx.x;
    double m = 0.0;
    if (i < H * W) {
        const int y = i / W, x = i - y * W;
        const R hc = h_in[i];
        const float c = (float)hc;
        const float t = (float)h_in[(y > 0 ? y - 1 : 0) * W + x];
        const float b = (float)h_in[(y < H - 1 ? y + 1 : H - 1) * W + x];
        const float l = (float)h_in[y * W + (x > 0 ? x - 1 : 0)];
        const float r = (float)h_in[y * W + (x < W - 1 ? x + 1 : W - 1)];
        const float h_conv = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(t, l), -4.0f * c), r), b);
        const float h_term = __fmul_rn(alpha_h, h_conv);
        const R diff = p[i] - hc;
        R p_term, change, hn;
        if constexpr (sizeof(R) == 8) {
            p_term = __dmul_rn(alpha_p, diff);
            change = __dadd_rn(p_term, (double)h_term);
            hn = __dadd_rn(hc, change);
        } else {
            p_term = __fmul_rn(alpha_p, diff);
            change = __fadd_rn(p_term, h_term);
            hn = __fadd_rn(hc, change);
        }
        h_out[i] = hn;
        m = fabs((double)change);
    }
    // non-negative doubles order like their bit patterns (a NaN change has the largest pattern: the loop goes on, as
    // `max_change > 1e-5` is False for NaN in Python the reference would stop -- a diverged diffuser (alpha >= 1) is refused
    // by the host before any step, emulator.py:1091-1096)
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(bits, o, 64);
        bits = other > bits ? other : bits;
    }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long v = smax[0];
        for (int q = 1; q < 4; ++q) v = smax[q] > v ? smax[q] : v;
        atomicMax(cur_max, v);
    }
}

// ---- device-resident variant (v2e_emu_run with a surround: no host step between the diffuser's steps).  Slot 0 = +inf, slot
// i + 1 = step i's max |change| (k_cs_step runs step i only if slot i is above the threshold, so every step after the first
// one that settles is a no-op launch); the steps that ran = the slots 0 .. n - 1 above the threshold.
__global__ void k_cs_begin(unsigned long long *slots, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) slots[i] = i == 0 ? 0x7FF0000000000000ull : 0ull;
}

__global__ __launch_bounds__(256) void k_cs_count(const unsigned long long *__restrict__ slots, int n, double thr, int *__restrict__ steps_out)
{
    __shared__ int s_part[4];
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += !(__longlong_as_double((long long)slots[i]) <= thr) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *steps_out = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// an odd number of steps leaves the result in the scratch plane
template <typename R>
__global__ __launch_bounds__(256) void k_cs_settle(const R *__restrict__ scratch, R *__restrict__ plane, int n, const int *__restrict__ steps)
{
    if (!(*steps & 1)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) plane[i] = scratch[i];
}

struct CsScratch {
    unsigned long long *dev = nullptr;  // [CHUNK + 1]: entry 0 carries the previous chunk's last maximum
    unsigned long long *host = nullptr; // pinned [CHUNK + 1]
    int device = -1;
};
thread_local CsScratch g_cs;

} // namespace

// One frame's surround update, all on the stream (declared in common.h; called by emu.hip's per-frame run loop)
int v2e_csdvs_enqueue_frame(const void *p_plane, void *h_plane, void *h_scratch, int H, int W, int f64, double alpha_p, double alpha_h,
                            int num_steps, double thr, unsigned long long *slots, int *steps_taken_dev, hipStream_t s)
{
    const int blocks = (int)v2e_cdiv((int64_t)H * W, 256);
    k_cs_begin<<<v2e_cdiv((int64_t)num_steps + 1, 256), 256, 0, s>>>(slots, num_steps);
    void *buf[2] = {h_plane, h_scratch};
    for (int i = 0; i < num_steps; ++i) {
        const void *hin = buf[i & 1];
        void *hout = buf[(i + 1) & 1];
        if (f64) k_cs_step<double><<<blocks, 256, 0, s>>>((const double *)p_plane, (const double *)hin, (double *)hout, H, W, alpha_p,
                                                         (float)alpha_h, thr, slots + i, slots + i + 1);
        else k_cs_step<float><<<blocks, 256, 0, s>>>((const float *)p_plane, (const float *)hin, (float *)hout, H, W, (float)alpha_p,
                                                     (float)alpha_h, thr, slots + i, slots + i + 1);
    }
    k_cs_count<<<1, 256, 0, s>>>(slots, num_steps, thr, steps_taken_dev);
    if (f64) k_cs_settle<double><<<blocks, 256, 0, s>>>((const double *)h_scratch, (double *)h_plane, H * W, steps_taken_dev);
    else k_cs_settle<float><<<blocks, 256, 0, s>>>((const float *)h_scratch, (float *)h_plane, H * W, steps_taken_dev);
    V2E_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int v2e_csdvs_update(const void *p_plane, void *h_plane, void *h_scratch, int H, int W, int f64, double alpha_p, double alpha_h,
                     int num_steps, double max_change_to_stop, int *steps_taken, double *last_max_change, void *stream)
{
    V2E_REQUIRE(p_plane && h_plane && h_scratch && steps_taken, "null");
    V2E_REQUIRE(H > 0 && W > 0 && num_steps >= 0, "bad size");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    {   // the device the planes live on, not whatever device is current on this thread
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h_plane) == hipSuccess) dev = at.device;
        else { (void)hipGetLastError(); V2E_HIP(hipGetDevice(&dev)); }
        V2E_HIP(hipSetDevice(dev));
    }
    if (g_cs.device != dev) {
        if (g_cs.dev) { hipFree(g_cs.dev); hipHostFree(g_cs.host); g_cs.dev = nullptr; g_cs.host = nullptr; }
        V2E_HIP(hipMalloc((void **)&g_cs.dev, sizeof(unsigned long long) * (CHUNK + 1)));
        V2E_HIP(hipHostMalloc((void **)&g_cs.host, sizeof(unsigned long long) * (CHUNK + 1)));
        g_cs.device = dev;
    }
    const int blocks = (int)v2e_cdiv((int64_t)H * W, 256);
    const size_t plane_bytes = (size_t)H * W * (f64 ? 8 : 4);
    void *buf[2] = {h_plane, h_scratch};
    int done = 0;         // steps enqueued so far
    int taken = 0;        // steps that ran
    bool stopped = false;
    double last = 2 * max_change_to_stop; // emulator.py:1105
    {
        const double inf = HUGE_VAL;
        memcpy(&g_cs.host[0], &inf, 8);
    }
    while (done < num_steps && !stopped) {
        const int n = num_steps - done < CHUNK ? num_steps - done : CHUNK;
        for (int i = 1; i <= n; ++i) g_cs.host[i] = 0ull;
        V2E_HIP(hipMemcpyAsync(g_cs.dev, g_cs.host, sizeof(unsigned long long) * (n + 1), hipMemcpyHostToDevice, s));
        for (int i = 0; i < n; ++i) {
            const void *hin = buf[(done + i) & 1];
            void *hout = buf[(done + i + 1) & 1];
            if (f64) k_cs_step<double><<<blocks, 256, 0, s>>>((const double *)p_plane, (const double *)hin, (double *)hout, H, W, alpha_p,
                                                             (float)alpha_h, max_change_to_stop, g_cs.dev + i, g_cs.dev + i + 1);
            else k_cs_step<float><<<blocks, 256, 0, s>>>((const float *)p_plane, (const float *)hin, (float *)hout, H, W, (float)alpha_p,
                                                         (float)alpha_h, max_change_to_stop, g_cs.dev + i, g_cs.dev + i + 1);
        }
        V2E_HIP(hipGetLastError());
        V2E_HIP(hipMemcpyAsync(g_cs.host + 1, g_cs.dev + 1, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, s));
        V2E_HIP(hipStreamSynchronize(s));
        for (int i = 1; i <= n; ++i) {
            double m;
            memcpy(&m, &g_cs.host[i], 8);
            ++taken;
            last = m;
            if (!(m > max_change_to_stop)) { stopped = true; break; }
        }
        g_cs.host[0] = g_cs.host[n];
        done += n;
    }
    if (taken & 1) V2E_HIP(hipMemcpyAsync(h_plane, h_scratch, plane_bytes, hipMemcpyDeviceToDevice, s));
    *steps_taken = taken;
    if (last_max_change) *last_max_change = last;
    return 0;
}

} // extern "C"
