// emu.hip -- DVS pixel model (EventEmulator.generate_events) for gfx950 / MI355X.
//
// What is computed, and where it comes from in the reference (SensorsINI/v2e):
//   k_init   first-frame state           v2ecore/emulator.py:681-717, _init :439-511
//   k_count  lin-log photoreceptor, intensity-dependent IIR low-pass, leak, ON/OFF
//            event counts, shot-noise decisions, global max
//                                        emulator.py:656-775, emulator_utils.py:18-173,297-351
//   k_shot   shot-noise decisions as a separate pass (tape mode draw order)
//   k_rank   refractory filter + per-wave (iteration,polarity) histograms
//                                        emulator.py:810-850
//   k_scan   exclusive scan of the histograms over waves (one workgroup per key)
//   k_emit   dense (t,x,y,p) list in reference order, shuffle, base/ts_mem update
//                                        emulator.py:861-870, 906-942, 1024-1059
//
// Layout: one thread per pixel, row-major, so a wave64 touches 64 consecutive x
// (256/512-byte fully coalesced requests per plane).  Per-pixel state is SoA, one
// plane per field, [n_clips][npx_pad].  Compaction is deterministic: a wave ranks its
// events with __ballot/__popcll, k_scan turns per-wave counts into row offsets, so the
// output order is exactly the reference's nonzero() order -- no atomic cursors.
//
// Numerics: every floating-point step reproduces the reference's dtype and operation
// order; this file must be compiled with -ffp-contract=off.
#include "common.h"
#include "../../include/v2e_detmath.h"

#include <cstdarg>
#include <cstdlib>
#include <chrono>
#include <vector>

namespace {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int RING = 64;        // record / control ring for the frame-at-a-time API
constexpr int SCAN_BLOCKS = 64; // workgroups of k_scan per clip
constexpr int STAMP_LAUNCHES = 128; // chain launches of one run that v2e_emu_launch_stamps keeps

// packed per-pixel scratch word written by k_count
constexpr uint32_t CNT_MASK = 0x00FFFFFFu;
constexpr uint32_t CNT_NEG = 1u << 24;
constexpr uint32_t CNT_SHOT_ON = 1u << 25;
constexpr uint32_t CNT_SHOT_OFF = 1u << 26;

struct FrameCtl {
    double t_prev, t_frame;
    // per-frame scalars the reference evaluates once in Python doubles; filled by the host with the
    // same IEEE operations (emulator_utils.py:80-84, 326-327) so kernels need not redo them per lane
    double dt_over_tau; // delta_time / (1/(math.pi*2*cutoff_hz)), 0 when cutoff_hz <= 0
    double shot_base;   // (shot_noise_rate_hz/2) * delta_time
    // timestamp generator (emulator.py:791-796) and refractory switch (:830) for every iteration
    // count n = 1..32, so the kernel needs no float64 division once the frame's max is known
    float ts_start[32], ts_stepf[32];
    uint32_t refr_mask; // bit n-1: refractory_period_s > delta_time / n
    float ts_end;       // (float)t_frame
    uint32_t refr_on_n; // smallest iteration count n >= 1 for which the rule is on (0xFFFFFFFF: never); the predicate is monotone in n
    uint32_t pad_;
};

__host__ __device__ inline FrameCtl make_ctl(double t_prev, double t_frame, double cutoff_hz, double shot_rate_hz, double refr_s)
{
    FrameCtl c;
    c.t_prev = t_prev; c.t_frame = t_frame;
    const double dt = t_frame - t_prev;
    c.dt_over_tau = 0.0;
    if (cutoff_hz > 0) { const double tau = 1.0 / (M_PI * 2 * cutoff_hz); c.dt_over_tau = dt / tau; }
    c.shot_base = (shot_rate_hz / 2) * dt;
    c.refr_mask = 0;
    const float end = (float)t_frame;
    c.ts_end = end;
    for (int n = 1; n <= 32; ++n) { // same IEEE operations as TsGen below
        const double ts_step = dt / (double)n;
        const float start = (float)(t_prev + ts_step);
        c.ts_start[n - 1] = start;
        c.ts_stepf[n - 1] = n > 1 ? (end - start) / (float)(n - 1) : 0.0f;
        if (refr_s > ts_step) c.refr_mask |= 1u << (n - 1);
    }
    c.refr_on_n = 0xFFFFFFFFu;
    c.pad_ = 0u;
    if (refr_s > 0) { // emulator.py:830 `refractory_period_s > ts_step`, ts_step = delta_time / n: first n that satisfies it
        const double g = dt / refr_s;
        if (!(g > 4.0e9)) {
            long long n0 = (long long)floor(g) - 2;
            if (n0 < 1) n0 = 1;
            while (!(refr_s > dt / (double)n0) && n0 < 0xFFFFFFFEll) ++n0;
            while (n0 > 1 && refr_s > dt / (double)(n0 - 1)) --n0;
            c.refr_on_n = (uint32_t)n0;
        }
    }
    return c;
}

// (1./20)*math.log(20) as evaluated by CPython (emulator_utils.py:34)
__device__ constexpr double LINLOG_F = 0x1.32c352f8fe941p-3;

struct KArgs {
    int W, npx, nwaves, nkeys_cap, max_iters;
    long long npx_pad;
    int scalar_thres, rng_mode, shuffle;
    int do_leak, do_shot, use_inten, has_cutoff, has_refr, log_input;
    double cutoff_two_pi; // math.pi*2*cutoff_hz
    double pos_div, neg_div;
    float pos_nom_f, neg_nom_f, pos_pre_scalar, neg_pre_scalar;
    float leak_hz_f, jit_f;
    float sigma_f, pos_mean_f, neg_mean_f, ln10cov_f;
    double shot_half_rate, inten_slope;
    double refr;
    float refr_f;
    unsigned long long seed;
    // state planes
    void *lp, *base;
    float *ts_mem, *pos_thres, *neg_thres, *noise_rate;
    // scratch
    uint32_t *cnt;
    uint32_t *hist; // [n_clips][nkeys_cap][nwaves]
    uint32_t *tot;  // [n_clips][nkeys_cap]
    // lin_log / inten01 of the 256 uint8 grey levels, built on device by k_lut with the same code
    const float *lut_L;
    const double *lut_I;
    // photoreceptor noise (frame-at-a-time API): state plane, tape draws (or nullptr), vrms as float32
    void *pn_arr;
    const float *pn_tape;
    float pn_vrms_f;
    // SCIDVS (emulator.py:56-80, 719-725; float64 state only): high-pass state, previous photoreceptor value, time constants
    void *sc_hp, *sc_prev;
    float *sc_tau;
    uint32_t sc_first_frame; // frame index at which scidvs_previous_photo is taken from the frame itself
    const void *cs_sur; // CSDVS: the surround plane the frame's photoreceptor output is compared against (emulator.py:753-754)
    int emit_guard; // k_emit: leave everything untouched (flag the record) when the frame's rows do not fit the buffer
    // model-state planes the reference keeps as attributes but the kernels otherwise never store (emulator.py:666, 749-754;
    // show_dvs_model_state, record_single_pixel_states): float64 [n_clips][npx_pad] each, written by k_count, nullptr = off
    double *dbg_lognew, *dbg_cms, *dbg_diff;
};

__device__ constexpr double SCIDVS_EFOLD = 1 / 0.7; // efold of the sinh conductance (emulator.py:78)

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float lin_log(double x)
{
    double y = (x <= 20.0) ? x * LINLOG_F : log(x);
    y = rint(y * 1e8) / 1e8;
    return (float)y;
}

// c10::div_floor_floating
template <typename R> __device__ __forceinline__ R div_floor(R a, R b)
{
    if (b == (R)0) return a / b;
    R mod = fmod(a, b);
    R div = (a - mod) / b;
    if (mod != (R)0 && ((b < (R)0) != (mod < (R)0))) div -= (R)1;
    R fd;
    if (div != (R)0) {
        fd = floor(div);
        if (div - fd > (R)0.5) fd += (R)1;
    } else {
        fd = copysign((R)0, a / b);
    }
    return fd;
}

// ---- wave64 reductions on the VALU data-parallel-primitive path (DPP), not through the LDS
// crossbar (__shfl_* lowers to ds_bpermute, an LDS-latency operation per step; with one wave per
// SIMD nothing hides it).  Within a 16-lane row: butterfly with quad_perm / row_half_mirror /
// row_mirror; across the four rows: v_readlane + scalar ops, so results are wave-uniform SGPRs.
#define V2E_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, V2E_DPP(v, 0xB1));  // quad_perm [1,0,3,2]
    v = max(v, V2E_DPP(v, 0x4E));  // quad_perm [2,3,0,1]
    v = max(v, V2E_DPP(v, 0x141)); // row_half_mirror
    v = max(v, V2E_DPP(v, 0x140)); // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t u)
{
    int v = (int)u;
    v += V2E_DPP(v, 0xB1);
    v += V2E_DPP(v, 0x4E);
    v += V2E_DPP(v, 0x141);
    v += V2E_DPP(v, 0x140);
    return (uint32_t)(__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
                      __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t u)
{
    int v = (int)u;
    v |= V2E_DPP(v, 0xB1);
    v |= V2E_DPP(v, 0x4E);
    v |= V2E_DPP(v, 0x141);
    v |= V2E_DPP(v, 0x140);
    return (uint32_t)(__builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) |
                      __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48));
}

// exclusive prefix sum across the 64 lanes: Hillis-Steele inside each 16-lane row with
// row_shr DPP (zero fill), row carries through v_readlane + scalar adds
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t u, int lane)
{
    int inc = (int)u;
    inc += V2E_DPP(inc, 0x111); // row_shr:1
    inc += V2E_DPP(inc, 0x112); // row_shr:2
    inc += V2E_DPP(inc, 0x114); // row_shr:4
    inc += V2E_DPP(inc, 0x118); // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(inc, 15), t1 = __builtin_amdgcn_readlane(inc, 31),
              t2 = __builtin_amdgcn_readlane(inc, 47);
    const int row = lane >> 4;
    const int off = row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : t0 + t1 + t2));
    return (uint32_t)(inc + off) - u;
}

// value of lane `idx` (idx wave-uniform) as a scalar
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int idx)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(idx));
}

// shot-noise decision, emulator_utils.py:326-349 (float64 compare of a float32 draw)
__device__ __forceinline__ uint32_t shot_bits(const KArgs &a, double inten01, double shot_base,
                                              float thp, float thn, float u)
{
    double F = shot_base * (a.inten_slope * inten01 + 1);
    float ppre = a.scalar_thres ? a.pos_pre_scalar : a.pos_nom_f / thp;
    float npre = a.scalar_thres ? a.neg_pre_scalar : a.neg_nom_f / thn;
    double on_thr = 1 - F * (double)ppre;
    double off_thr = F * (double)npre;
    uint32_t b = 0;
    if ((double)u > on_thr) b |= CNT_SHOT_ON;
    if ((double)u < off_thr) b |= CNT_SHOT_OFF;
    return b;
}

// timestamps of the n iterations of one frame (emulator.py:791-796)
struct TsGen {
    const float *tab;
    float start, end, step;
    uint32_t n;
    __device__ __forceinline__ TsGen(const FrameCtl &c, int n_, const float *tab_) : tab(tab_), n((uint32_t)n_)
    {
        double dt = c.t_frame - c.t_prev;
        double ts_step = dt / (double)n_;
        start = (float)(c.t_prev + ts_step);
        end = (float)c.t_frame;
        step = (n_ > 1) ? (end - start) / (float)(n_ - 1) : 0.0f;
    }
    __device__ __forceinline__ TsGen(float start_, float end_, float step_, int n_) : tab(nullptr), start(start_), end(end_), step(step_), n((uint32_t)n_) {}
    __device__ __forceinline__ float operator()(int i) const
    {
        return tab ? tab[i] : v2e_ts_formula((uint32_t)i, n, start, end, step);
    }
};

// lin_log(x) and (x+20)/275 for x = 0..255, evaluated by the very functions the per-pixel path
// uses, so a lookup returns bit-identical values (uint8 frames only)
__global__ void k_lut(float *lut_L, double *lut_I)
{
    const int i = threadIdx.x;
    const double x = (double)i;
    lut_L[i] = lin_log(x);
    lut_I[i] = (x + 20.0) / 275.0;
}

// ------------------------------------------------------------------ k_init
template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_init(KArgs a, const FT *__restrict__ frame, double t_frame,
                                                const float *__restrict__ tp_tape,
                                                const float *__restrict__ tn_tape,
                                                const float *__restrict__ nr_tape)
{
    const int clip = blockIdx.y;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= a.npx) return;
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const size_t fp = (size_t)clip * a.npx + p;
    double x = (double)frame[fp];
    const double L = a.log_input ? x : (double)lin_log(x); // emulator.py:666
    R lp;
    if (a.has_cutoff) {
        double delta_time = t_frame - 0.0;
        double tau = 1.0 / a.cutoff_two_pi;
        double dt_over_tau = delta_time / tau;
        double inten01 = (x + 20.0) / 275.0;
        double eps = inten01 * dt_over_tau;
        if (eps > 1.0) eps = 1.0;
        lp = (R)((1.0 - eps) * (double)L + eps * (double)L);
    } else {
        lp = (R)L;
    }
    ((R *)a.lp)[sp] = lp;
    ((R *)a.base)[sp] = lp;
    float n_pos = 0.f, n_neg = 0.f, n_rate = 0.f;
    if (a.rng_mode == V2E_RNG_PHILOX)
        v2e_draw_init(a.seed, (uint32_t)clip, (uint32_t)p, &n_pos, &n_neg, &n_rate);
    float tp, tn;
    if (!a.scalar_thres) {
        if (a.rng_mode == V2E_RNG_PHILOX) {
            tp = n_pos * a.sigma_f + a.pos_mean_f;
            tn = n_neg * a.sigma_f + a.neg_mean_f;
        } else {
            tp = tp_tape[fp];
            tn = tn_tape[fp];
        }
        tp = tp < 0.01f ? 0.01f : tp;
        tn = tn < 0.01f ? 0.01f : tn;
    } else {
        tp = a.pos_mean_f;
        tn = a.neg_mean_f;
    }
    a.pos_thres[sp] = tp;
    a.neg_thres[sp] = tn;
    if (a.do_leak)
        a.noise_rate[sp] = (a.rng_mode == V2E_RNG_PHILOX) ? v2e_det_expf(a.ln10cov_f * n_rate) : nr_tape[fp];
    if (a.has_refr) a.ts_mem[sp] = 0.0f - a.refr_f;
    if (a.sc_hp) { // scidvs_highpass = zeros_like(lp); tau = SCIDVS_TAU_S * exp(normal(0, SCIDVS_TAU_COV)) (emulator.py:480-483, 719-722)
        ((R *)a.sc_hp)[sp] = (R)0;
        ((R *)a.sc_prev)[sp] = (R)0;
        if (a.rng_mode == V2E_RNG_PHILOX) a.sc_tau[sp] = 0.01f * v2e_det_expf(0.5f * v2e_draw_scidvs(a.seed, (uint32_t)clip, (uint32_t)p));
    }
}

// low_pass_filter (emulator_utils.py:69-104) of one pixel: the new lp_log_frame value
template <typename R>
__device__ __forceinline__ R lp_next(const KArgs &a, double L, double inten01, double delta_time, size_t sp)
{
    if (a.has_cutoff) { // R == double
        double tau = 1.0 / a.cutoff_two_pi;
        double dt_over_tau = delta_time / tau;
        double eps = inten01 * dt_over_tau;
        if (eps > 1.0) eps = 1.0;
        return (R)((1.0 - eps) * (double)((R *)a.lp)[sp] + eps * (double)L);
    }
    return (R)L;
}

// CSDVS: the frame's lp_log_frame BEFORE the frame is counted -- the surround's diffuser is stepped against it
// (emulator.py:707-708 sits between the low-pass and the event computation); k_count recomputes the same value
template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_cs_lp(KArgs a, const FT *__restrict__ frame, const FrameCtl *__restrict__ ctl, R *__restrict__ out)
{
    const int clip = blockIdx.y;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= a.npx) return;
    const FrameCtl c = ctl[clip];
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const double x = (double)frame[(size_t)clip * a.npx + p];
    const double L = a.log_input ? x : (double)lin_log(x);
    const double inten01 = a.use_inten ? (x + 20.0) / 275.0 : 0.0;
    out[sp] = lp_next<R>(a, L, inten01, c.t_frame - c.t_prev, sp);
}

// ----------------------------------------------------------------- k_count
template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_count(KArgs a, const FT *__restrict__ frame,
                                                 const FrameCtl *__restrict__ ctl,
                                                 const uint32_t *__restrict__ fidx_base, uint32_t fidx_off,
                                                 const float *__restrict__ leak_tape,
                                                 const float *__restrict__ shot_tape, v2e_frame_rec *rec,
                                                 const FrameCtl *__restrict__ ctl_host, FrameCtl *ctl_copy,
                                                 double t_prev_v, double t_frame_v)
{
    __shared__ int smax[BLOCK / WAVE];
    const int clip = blockIdx.y;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    // ctl == nullptr (v2e_emu_frame): the frame's times arrive by value, and workgroup 0 copies the frame's scalars from pinned
    // host memory (ctl_host) to device memory (ctl_copy) for the kernels behind this one -- no launch of its own for that
    double delta_time_v = t_frame_v - t_prev_v;
    if (ctl) delta_time_v = ctl[clip].t_frame - ctl[clip].t_prev;
    else if (ctl_host && blockIdx.x == 0 && clip == 0 && threadIdx.x < sizeof(FrameCtl) / 4)
        ((uint32_t *)ctl_copy)[threadIdx.x] = ((const uint32_t *)ctl_host)[threadIdx.x];
    const uint32_t frame_idx = (fidx_base ? *fidx_base : 0u) + fidx_off;
    int m = 0;
    if (p < a.npx) {
        const size_t sp = (size_t)clip * a.npx_pad + p;
        const size_t fp = (size_t)clip * a.npx + p;
        const double delta_time = delta_time_v;
        double x = (double)frame[fp];
        const double L = a.log_input ? x : (double)lin_log(x); // emulator.py:666
        double inten01 = a.use_inten ? (x + 20.0) / 275.0 : 0.0;
        float r = 0.f, u = 0.f;
        const bool shot_here = a.do_shot && (a.rng_mode == V2E_RNG_PHILOX || shot_tape != nullptr);
        if (a.rng_mode == V2E_RNG_PHILOX) {
            if ((a.do_leak && a.jit_f != 0.f) || a.do_shot)
                v2e_draw_frame(a.seed, (uint32_t)clip, frame_idx, (uint32_t)p, &r, &u);
        } else {
            if (a.do_leak) r = leak_tape[fp];
            if (shot_here) u = shot_tape[fp];
        }
        const float thp = a.pos_thres[sp], thn = a.neg_thres[sp];
        float delta_leak = 0.f;
        if (a.do_leak) { // emulator_utils.py:126-129, float32 left to right
            float rate = (a.leak_hz_f * a.noise_rate[sp]) * (1.0f - a.jit_f * r);
            delta_leak = ((float)delta_time * rate) * thp;
        }
        const R lpn = lp_next<R>(a, L, inten01, delta_time, sp);
        ((R *)a.lp)[sp] = lpn;
        R b = ((R *)a.base)[sp];
        if (a.do_leak) {
            b = b - (R)delta_leak;
            ((R *)a.base)[sp] = b;
        }
        R pn = (R)0.0f;
        if (a.pn_arr) { // emulator.py:694-701: float32 white noise through low_pass_filter(noise, arr, None, dt, cutoff)
            const float rn = a.rng_mode == V2E_RNG_PHILOX ? v2e_draw_pnoise(a.seed, (uint32_t)clip, frame_idx, (uint32_t)p) : a.pn_tape[fp];
            const float noise = a.pn_vrms_f * rn;                // python float * float32 tensor
            const double eps_n = delta_time / (1.0 / a.cutoff_two_pi); // emulator_utils.py:97, not clamped
            const float term2 = (float)eps_n * noise;            // eps * float32 tensor
            const double pnn = (1.0 - eps_n) * (double)((R *)a.pn_arr)[sp] + (double)term2;
            ((R *)a.pn_arr)[sp] = (R)pnn;
            pn = (R)pnn;
        }
        R photo = lpn;
        if (a.sc_hp) { // SCIDVS: nonlinear CR high-pass of the photoreceptor, amplified (emulator.py:719-725, 747; R == double)
            const R prev = frame_idx == a.sc_first_frame ? lpn : ((R *)a.sc_prev)[sp]; // first frame: clone of lp_log_frame
            R hp = ((R *)a.sc_hp)[sp];
            const float inv_tau = 1.0f / a.sc_tau[sp];                  // torch.div(1, tau): float32
            if constexpr (sizeof(R) == 8) {
                const R sh = (R)sinh((double)hp / SCIDVS_EFOLD);         // torch.sinh(v / efold)
                const R dvdt = (R)inv_tau * sh;
                hp = hp + ((lpn - prev) - (R)(delta_time * (double)dvdt));
            } else { // float32 state: every operand is a float32 tensor, Python scalars take the tensor's type; torch's
                     // vectorised float32 sinh is Sleef's sinhf_u10, restated bit for bit in v2e_detmath.h
                const float sh = v2e_sleef_sinhf((float)hp / (float)SCIDVS_EFOLD);
                const float dvdt = inv_tau * sh;
                hp = (R)((float)hp + (((float)lpn - (float)prev) - ((float)delta_time * dvdt)));
            }
            ((R *)a.sc_hp)[sp] = hp;
            ((R *)a.sc_prev)[sp] = lpn;
            photo = (R)2 * hp;                                           // SCIDVS_GAIN * scidvs_highpass
        }
        R diff = (photo + pn) - b; // photoreceptor + photoreceptor_noise_arr - base_log_frame (emulator.py:747-751)
        if (a.cs_sur) diff = ((photo + pn) - ((const R *)a.cs_sur)[sp]) - b; // c_minus_s_frame - base_log_frame (:753-754)
        if (a.dbg_diff) {
            a.dbg_lognew[sp] = L;
            a.dbg_cms[sp] = a.cs_sur ? (double)((photo + pn) - ((const R *)a.cs_sur)[sp]) : 0.0;
            a.dbg_diff[sp] = (double)diff;
        }
        R pf = diff > (R)0 ? diff : (R)0;
        R nf = (-diff) > (R)0 ? -diff : (R)0;
        R tpd = a.scalar_thres ? (R)a.pos_div : (R)thp;
        R tnd = a.scalar_thres ? (R)a.neg_div : (R)thn;
        int pc = (int)div_floor<R>(pf, tpd);
        int nc = (int)div_floor<R>(nf, tnd);
        uint32_t w = 0;
        if (pc > 0) w = (uint32_t)pc & CNT_MASK;
        else if (nc > 0) w = ((uint32_t)nc & CNT_MASK) | CNT_NEG;
        if (shot_here) {
            double shot_base = a.shot_half_rate * delta_time;
            w |= shot_bits(a, inten01, shot_base, thp, thn, u);
        }
        a.cnt[sp] = w;
        m = pc > nc ? pc : nc;
    }
    m = wave_max_i32(m);
    if ((threadIdx.x & (WAVE - 1)) == 0) smax[threadIdx.x / WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        int bm = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
        if (bm > 0) atomicMax(&rec[clip].max_events, bm);
    }
}

// ------------------------------------------------------------------ k_shot
template <typename FT>
__global__ __launch_bounds__(BLOCK) void k_shot(KArgs a, const FT *__restrict__ frame,
                                                const FrameCtl *__restrict__ ctl,
                                                const float *__restrict__ shot_tape)
{
    const int clip = blockIdx.y;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= a.npx) return;
    const FrameCtl c = ctl[clip];
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const size_t fp = (size_t)clip * a.npx + p;
    double x = (double)frame[fp];
    double inten01 = (x + 20.0) / 275.0;
    double shot_base = a.shot_half_rate * (c.t_frame - c.t_prev);
    uint32_t w = a.cnt[sp] & ~(CNT_SHOT_ON | CNT_SHOT_OFF);
    w |= shot_bits(a, inten01, shot_base, a.pos_thres[sp], a.neg_thres[sp], shot_tape[fp]);
    a.cnt[sp] = w;
}

// ------------------------------------------------------------------ k_rank
// One wave = 64 consecutive pixels.  For every iteration i < M the wave counts its
// surviving ON and OFF events (keys 2i, 2i+1); keys 2M, 2M+1 are the shot pair.
__global__ __launch_bounds__(BLOCK) void k_rank(KArgs a, const FrameCtl *__restrict__ ctl,
                                                const v2e_frame_rec *__restrict__ rec,
                                                const float *__restrict__ ts_tab, int n_ts)
{
    const int clip = blockIdx.y;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = (blockIdx.x * BLOCK + threadIdx.x) / WAVE;
    if (w >= a.nwaves) return;
    const int p = w * WAVE + lane;
    const int M = rec[clip].max_events;
    if (M > a.max_iters) return;
    const int n = M > 0 ? M : 1;
    const FrameCtl c = ctl[clip];
    const TsGen tg(c, n, ts_tab ? ts_tab + (size_t)clip * n_ts : nullptr);
    const bool use_refr = a.has_refr && (a.refr > (c.t_frame - c.t_prev) / (double)n);
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const uint32_t cw = p < a.npx ? a.cnt[sp] : 0u;
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    float tsm = (use_refr && p < a.npx) ? a.ts_mem[sp] : 0.f;
    uint32_t *hrow = a.hist + (size_t)clip * a.nkeys_cap * a.nwaves;
    bool alive = true; // some lane of the wave still has a candidate at this iteration
    for (int kb = 0; kb < 2 * M; kb += WAVE) {
        uint32_t mine = 0;
        const int i0 = kb >> 1;
        for (int ii = 0; ii < WAVE / 2 && i0 + ii < M && alive; ++ii) {
            const int i = i0 + ii;
            const bool cand = mag > i;
            if (__ballot(cand) == 0ull) { alive = false; break; }
            bool pass = cand;
            if (use_refr) {
                const float t = tg(i);
                const float pt = (cand ? 1.0f : 0.0f) * t - tsm;
                pass = pt > a.refr_f;
                if (pass) tsm = t;
            }
            const unsigned long long bo = __ballot(pass && !neg);
            const unsigned long long bf = __ballot(pass && neg);
            if (lane == 2 * ii) mine = (uint32_t)__popcll(bo);
            if (lane == 2 * ii + 1) mine = (uint32_t)__popcll(bf);
        }
        const int key = kb + lane;
        if (key < 2 * M) hrow[(size_t)key * a.nwaves + w] = mine;
    }
    const unsigned long long so = __ballot((cw & CNT_SHOT_ON) != 0);
    const unsigned long long sf = __ballot((cw & CNT_SHOT_OFF) != 0);
    if (lane == 0) hrow[(size_t)(2 * M) * a.nwaves + w] = (uint32_t)__popcll(so);
    if (lane == 1) hrow[(size_t)(2 * M + 1) * a.nwaves + w] = (uint32_t)__popcll(sf);
}

// ------------------------------------------------------------------ k_scan
// One workgroup per key: in-place exclusive scan over the per-wave counts, total to tot[].
__global__ __launch_bounds__(BLOCK) void k_scan(KArgs a, const v2e_frame_rec *__restrict__ rec)
{
    __shared__ uint32_t wsum[BLOCK / WAVE];
    const int clip = blockIdx.y;
    const int M = rec[clip].max_events;
    if (M > a.max_iters) return;
    const int nkeys = 2 * M + 2;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int ipt = (a.nwaves + BLOCK - 1) / BLOCK;
    for (int key = blockIdx.x; key < nkeys; key += gridDim.x) {
        uint32_t *row = a.hist + ((size_t)clip * a.nkeys_cap + key) * a.nwaves;
        const int b = tid * ipt, e = min(b + ipt, a.nwaves);
        uint32_t s = 0;
        for (int j = b; j < e; ++j) s += row[j];
        const uint32_t ex = wave_excl_scan_u32(s, lane);
        if (lane == WAVE - 1) wsum[wid] = ex + s;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int q = 0; q < BLOCK / WAVE; ++q) {
            if (q < wid) wbase += wsum[q];
            total += wsum[q];
        }
        uint32_t run = wbase + ex;
        for (int j = b; j < e; ++j) {
            const uint32_t v = row[j];
            row[j] = run;
            run += v;
        }
        if (tid == 0) a.tot[(size_t)clip * a.nkeys_cap + key] = total;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ k_emit
template <typename R>
__global__ __launch_bounds__(BLOCK) void k_emit(KArgs a, const FrameCtl *__restrict__ ctl, v2e_frame_rec *rec,
                                                const v2e_frame_rec *__restrict__ rec_prev,
                                                const unsigned long long *__restrict__ ev_offset0,
                                                const uint32_t *__restrict__ fidx_base, uint32_t fidx_off,
                                                const float *__restrict__ ts_tab, int n_ts,
                                                float4 *__restrict__ events, unsigned long long cap)
{
    const int clip = blockIdx.y;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = (blockIdx.x * BLOCK + threadIdx.x) / WAVE;
    if (w >= a.nwaves) return;
    const int p = w * WAVE + lane;
    const uint32_t frame_idx = (fidx_base ? *fidx_base : 0u) + fidx_off;
    const int M = rec[clip].max_events;
    unsigned long long ev0 = 0;
    if (ev_offset0) ev0 = ev_offset0[clip];
    else if (rec_prev) ev0 = rec_prev[clip].ev_offset + rec_prev[clip].n_events;
    if (M > a.max_iters) {
        if (w == 0 && lane == 0) {
            rec[clip].flags |= V2E_FLAG_ITERS_CLAMPED;
            rec[clip].ev_offset = ev0;
        }
        return;
    }
    const uint32_t *trow = a.tot + (size_t)clip * a.nkeys_cap;
    if (a.emit_guard) { // speculative launch (v2e_emu_frame): the host has not seen the totals yet
        uint32_t tsum = 0;
        for (int kb = 0; kb < 2 * M + 2; kb += WAVE) tsum += kb + lane < 2 * M + 2 ? trow[kb + lane] : 0u;
        if (ev0 + wave_sum_u32(tsum) > cap) {
            if (w == 0 && lane == 0) {
                rec[clip].flags |= V2E_FLAG_EVENTS_DROPPED;
                rec[clip].ev_offset = ev0;
            }
            return;
        }
    }
    const int n = M > 0 ? M : 1;
    const FrameCtl c = ctl[clip];
    const TsGen tg(c, n, ts_tab ? ts_tab + (size_t)clip * n_ts : nullptr);
    const bool use_refr = a.has_refr && (a.refr > (c.t_frame - c.t_prev) / (double)n);
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const bool valid = p < a.npx;
    const uint32_t cw = valid ? a.cnt[sp] : 0u;
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    float tsm = (use_refr && valid) ? a.ts_mem[sp] : 0.f;
    const uint32_t *hrow = a.hist + (size_t)clip * a.nkeys_cap * a.nwaves;
    float4 *ev = events + (size_t)clip * cap;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float fx = (float)(p % a.W), fy = (float)(p / a.W);
    const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
    uint32_t carry = 0;          // events of all earlier iterations (whole frame)
    uint32_t sum_on = 0, sum_off = 0;
    int fcount = 0;              // this pixel's surviving events
    bool dropped = false;
    bool alive = true;
    for (int kb = 0; kb < 2 * M; kb += WAVE) {
        const int key = kb + lane;
        const uint32_t t_k = key < 2 * M ? trow[key] : 0u;
        const uint32_t o_k = key < 2 * M ? hrow[(size_t)key * a.nwaves + w] : 0u;
        const uint32_t kb_k = carry + wave_excl_scan_u32(t_k, lane);
        const uint32_t chunk_total = wave_sum_u32(t_k);
        sum_on += wave_sum_u32((lane & 1) ? 0u : t_k);
        sum_off += wave_sum_u32((lane & 1) ? t_k : 0u);
        const int i0 = kb >> 1;
        for (int ii = 0; ii < WAVE / 2 && i0 + ii < M && alive; ++ii) {
            const int i = i0 + ii;
            const bool cand = mag > i;
            if (__ballot(cand) == 0ull) { alive = false; break; }
            const float t = tg(i);
            bool pass = cand;
            if (use_refr) {
                const float pt = (cand ? 1.0f : 0.0f) * t - tsm;
                pass = pt > a.refr_f;
                if (pass) tsm = t;
            }
            const unsigned long long bo = __ballot(pass && !neg);
            const unsigned long long bf = __ballot(pass && neg);
            if ((bo | bf) == 0ull) continue;
            const uint32_t it_base = lane_value(kb_k, 2 * ii);
            const uint32_t tot_on = lane_value(t_k, 2 * ii);
            const uint32_t tot_off = lane_value(t_k, 2 * ii + 1);
            const uint32_t off_on = lane_value(o_k, 2 * ii);
            const uint32_t off_off = lane_value(o_k, 2 * ii + 1);
            if (pass) {
                uint32_t cidx = neg ? tot_on + off_off + (uint32_t)__popcll(bf & lt)
                                    : off_on + (uint32_t)__popcll(bo & lt);
                if (shuf) {
                    v2e_perm_t pm;
                    v2e_perm_init(&pm, a.seed, (uint32_t)clip, frame_idx, (uint32_t)i, tot_on + tot_off);
                    cidx = v2e_perm_apply(&pm, cidx);
                }
                const unsigned long long row = ev0 + it_base + cidx;
                if (row < cap) ev[row] = make_float4(t, fx, fy, neg ? -1.0f : 1.0f);
                else dropped = true;
                ++fcount;
            }
        }
        carry += chunk_total;
    }
    // shot-noise events: after all signal events, ON block then OFF block, ts[-1], unshuffled
    const uint32_t son_tot = trow[2 * M], soff_tot = trow[2 * M + 1];
    if (a.do_shot) {
        const bool s_on = (cw & CNT_SHOT_ON) != 0, s_off = (cw & CNT_SHOT_OFF) != 0;
        const unsigned long long so = __ballot(s_on), sf = __ballot(s_off);
        if (so | sf) {
            const float tl = tg(n - 1);
            if (s_on) {
                const unsigned long long row = ev0 + carry + hrow[(size_t)(2 * M) * a.nwaves + w] + (uint32_t)__popcll(so & lt);
                if (row < cap) ev[row] = make_float4(tl, fx, fy, 1.0f);
                else dropped = true;
            }
            if (s_off) {
                const unsigned long long row = ev0 + carry + son_tot + hrow[(size_t)(2 * M + 1) * a.nwaves + w] + (uint32_t)__popcll(sf & lt);
                if (row < cap) ev[row] = make_float4(tl, fx, fy, -1.0f);
                else dropped = true;
            }
        }
    }
    // emulator.py:936-942
    if (valid) {
        const bool shot = a.do_shot && (cw & (CNT_SHOT_ON | CNT_SHOT_OFF));
        if (fcount > 0 || shot) {
            R b = ((R *)a.base)[sp];
            const float dp = (float)(neg ? 0 : fcount) * a.pos_thres[sp];
            const float dn = (float)(neg ? fcount : 0) * a.neg_thres[sp];
            b = b + (R)dp;
            b = b - (R)dn;
            if (shot) b = ((R *)a.lp)[sp];
            ((R *)a.base)[sp] = b;
        }
        if (use_refr && fcount > 0) a.ts_mem[sp] = tsm;
    }
    if (__ballot(dropped) != 0ull && lane == 0) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
    if (w == 0 && lane == 0) {
        rec[clip].n_signal = carry;
        rec[clip].n_events = carry + (a.do_shot ? son_tot + soff_tot : 0u);
        rec[clip].n_on = sum_on + (a.do_shot ? son_tot : 0u);
        rec[clip].n_off = sum_off + (a.do_shot ? soff_tot : 0u);
        rec[clip].ev_offset = ev0;
    }
}

// out[row0 + j] = in[row0 + idx[j]]  (events_curr_iter[idx], emulator.py:869)
__global__ __launch_bounds__(BLOCK) void k_permute(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                   const int32_t *__restrict__ idx, unsigned long long row0,
                                                   unsigned long long n)
{
    const unsigned long long j = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) out[row0 + j] = in[row0 + (unsigned long long)idx[j]];
}

// ------------------------------------------------------------------ v2e_emu_frame's two ends
// Everything a frame needs goes through the compute queue: a copy engine between kernels costs a cross-queue dependency at
// either end (~20 us each on this runtime), more than the whole frame's kernels.  The frame's scalars arrive as a kernel
// argument, its pixels are read by k_count straight from the pinned staging buffer, and the rows go back by a kernel writing
// pinned host memory.
struct FrameScratch { // device
    v2e_frame_rec rec[2]; // alternating between calls: a frame's last kernel zeroes the one the next frame counts into
    FrameCtl ctl;         // the frame's scalars, copied from pinned host memory by k_count's workgroup 0
};

// rows [row0, min(n_events, row0 + max_rows)) of the frame to the pinned host buffer, and the frame record with them
__global__ __launch_bounds__(256) void k_frame_rows_to_host(const float4 *__restrict__ ev, const v2e_frame_rec *__restrict__ rec,
                                                            float4 *__restrict__ out_rows, v2e_frame_rec *__restrict__ out_rec,
                                                            unsigned long long row0, unsigned long long max_rows,
                                                            v2e_frame_rec *__restrict__ rec_next)
{
    const v2e_frame_rec r = *rec;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (out_rec) *out_rec = r;
        if (rec_next) { // the record the NEXT frame's k_count takes its maximum in (nobody in this launch reads it)
            v2e_frame_rec z;
            memset(&z, 0, sizeof(z));
            *rec_next = z;
        }
    }
    if (r.flags & (V2E_FLAG_EVENTS_DROPPED | V2E_FLAG_ITERS_CLAMPED)) return;
    const unsigned long long n = r.n_events < row0 + max_rows ? r.n_events : row0 + max_rows;
    for (unsigned long long i = row0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
        out_rows[i] = ev[i];
}

#include "emu_chain.h"

} // namespace

// Zero-fill as a KERNEL node.  hipMemsetAsync captured into a hipGraph turned out unreliable for small buffers on
// this runtime (ROCm 7.0 / HIP 7.0.5): from the third launch of the same graph exec a 16..160-byte memset node wrote
// the high half of an address instead of zeros (frame-record flags, the emission offsets, the rendezvous counters).
namespace {
__global__ void k_zero_words(uint32_t *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// up to four ranges zeroed by ONE launch (the start of a run: four dependent 4-us launches were 17 us of every 300-frame step)
struct ZeroRanges { uint32_t *p[4]; unsigned long long n[4]; };
__global__ __launch_bounds__(256) void k_zero_ranges(ZeroRanges z)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long base = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (i >= base && i < base + z.n[r]) z.p[r][i - base] = 0u;
        base += z.n[r];
    }
}
} // namespace
static inline hipError_t zero_async(void *p, size_t bytes, hipStream_t s)
{
    const size_t n = (bytes + 3) / 4; // all callers pass multiples of 4
    if (n == 0) return hipSuccess;
    k_zero_words<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((uint32_t *)p, n);
    return hipGetLastError();
}

// =================================================================== host side
struct v2e_emu {
    int H, W, n_clips, max_iters, device;
    int npx, nwaves, nkeys_cap;
    long long npx_pad;
    void *lp = nullptr, *base = nullptr;
    float *ts_mem = nullptr, *pos_thres = nullptr, *neg_thres = nullptr, *noise_rate = nullptr;
    uint32_t *cnt = nullptr, *hist = nullptr, *tot = nullptr;
    void *pn_arr = nullptr;           // photoreceptor_noise_arr plane (v2e_emu_set_pnoise)
    const float *pn_tape = nullptr;
    void *sc_hp = nullptr, *sc_prev = nullptr; // SCIDVS planes (v2e_emu_set_scidvs)
    const void *cs_sur = nullptr;              // CSDVS surround plane (v2e_emu_set_csdvs)
    // CSDVS in a device-resident run (v2e_emu_set_csdvs_run): per-frame Euler parameters, scratch planes, step slots
    void *csr_scratch = nullptr, *csr_lp = nullptr;
    std::vector<double> csr_ap, csr_ah;
    std::vector<int> csr_steps;
    double csr_thr = 0.0;
    int *csr_steps_dev = nullptr;
    unsigned long long *csr_slots = nullptr;
    int csr_slots_cap = 0;
    float *sc_tau = nullptr;
    uint32_t sc_first_frame = 0;
    double *dbg_lognew = nullptr, *dbg_cms = nullptr, *dbg_diff = nullptr; // v2e_emu_set_model_state_planes
    int ngroups = 0;
    float *lut_L = nullptr;
    double *lut_I = nullptr;
    v2e_frame_rec *rec_ring = nullptr; // [RING][n_clips]
    FrameCtl *ctl_ring = nullptr;      // [RING][n_clips]
    FrameCtl *ctl_host = nullptr;      // pinned staging [RING][n_clips]
    unsigned long long *off_dev = nullptr, *off_host = nullptr; // [n_clips] explicit event offsets
    // v2e_emu_frame (one call per frame): pinned staging for the frame, the frame record + per-key totals, the event rows
    // pinned staging of a host frame (read by k_count in place), device scratch [v2e_frame_rec | FrameCtl], pinned record and
    // rows (written by k_frame_rows_to_host); *_dev: the device-side addresses of the pinned buffers
    unsigned char *fr_stage = nullptr, *fr_stage_dev = nullptr, *fr_dev = nullptr, *fr_rec_host_dev = nullptr;
    unsigned char *fr_par = nullptr, *fr_par_dev = nullptr; // the frame's FrameCtl, pinned
    int fr_flip = 0;                                        // which of the two frame records this call counts into
    float *fr_ev_host_dev = nullptr;
    float *fr_user_rows = nullptr, *fr_user_rows_dev = nullptr; // v2e_emu_frame_host_rows: the next frame's rows go here
    uint64_t fr_user_cap = 0;
    size_t fr_bytes = 0;
    unsigned char *fr_rec_host = nullptr; // v2e_frame_rec + nkeys_cap totals
    int fr_rec_keys = 0;
    float *fr_ev_host = nullptr;
    uint64_t fr_ev_cap = 0, fr_est = 0;
    unsigned long long *off_zero = nullptr; // [n_clips] zeros
    // multi-frame run
    FrameCtl *run_ctl = nullptr;       // device [run_cap][n_clips]
    FrameCtl *run_ctl_host2[2] = {nullptr, nullptr}; // pinned staging, two sets
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    int run_stage = 0;
    int run_cap = 0;
    uint32_t *run_fidx = nullptr;      // device: frame_idx0 of the current run
    uint32_t *run_fidx_host = nullptr; // pinned
    const void **run_frames = nullptr;      // device: the current run's frames (k_ahead / k_chain read the pointer from here)
    const void **run_frames_host = nullptr; // pinned, two slots
    // per-launch device time stamps of the chain kernel (v2e_emu_launch_stamps): [runs_cap][STAMP_LAUNCHES][2] wall-clock words
    unsigned long long *stamps = nullptr;
    int stamps_runs = 0;                       // runs the buffer holds (0: off)
    unsigned long long stamp_seq = 0;          // runs stamped so far
    unsigned long long **stamp_slot = nullptr;      // device: the current run's slot (nullptr: off), read by k_chain
    unsigned long long **stamp_slot_host = nullptr; // pinned, two slots
    // captured runs, keyed by everything baked into them (buffers, sizes, parameters): a caller that alternates between two
    // sets of frame / event / record buffers (the asynchronous API) replays two graphs
    struct CachedGraph { std::vector<unsigned char> key; hipGraphExec_t exec; unsigned long long used; };
    std::vector<CachedGraph> graphs;
    unsigned long long graph_clock = 0;
    void drop_graphs()
    {
        for (auto &g : graphs) hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
    unsigned long long *dbg = nullptr; // dev tool (v2e_emu_debug_timeline)
    int n_cu = 256;
    double prof_ms[4] = {0, 0, 0, 0}; // count, rank, scan, emit (use_graph == 2)
    int prof_launches = 0;
    std::vector<float> prof_chain_us; // the chain launches of the last instrumented run, one by one
    int prof_emit_batches = 0, prof_step_launches = 0;
    unsigned long long *run_off = nullptr; // [run_off_cap][n_clips] event offset at the start of every emission batch of the run
    int run_off_cap = 0;
    hipStream_t side = nullptr;        // the event writer of the chain: k_cemit
    std::vector<hipEvent_t> ev_fork, ev_join;
    // K-frames-per-launch chain (emu_chain.h); allocated on first use by v2e_emu_run
    int ch_ring_pref = 0; // ring depth in batches, chosen by the handle's first device-resident run (chain_ring_batches)
    int ch_K = 0, ch_D = 0, ch_nD = 3, ch_nwp = 0, ch_launch_cap = 0, ch_max_blocks = 0, ch_fused = -1, ch_inst = -1;
    uint32_t *ch_cnt = nullptr;     // [ch_D][n_clips][npx_pad]
    uint32_t *ch_ruleM = nullptr;   // [ch_D][n_clips]
    uint16_t *ch_wmax = nullptr;    // [2][ch_E][n_clips][ch_nwp]
    uint16_t *ch_wtot = nullptr;    // [3][ch_E][n_clips][nkeys_cap][ch_nwp]
    float *ch_tsold = nullptr;      // [ch_D][n_clips][npx_pad] (refractory runs)
    void *ch_ck = nullptr;          // refractory runs: 2 launch parities x (K / 8 - 1) checkpoints x (base 8 B, lp 8 B, ts 4 B) planes
    uint4 *ch_rec = nullptr;        // [ch_D][n_clips][npx_pad] k_ahead's per-(frame, pixel) records
    hipStream_t ahead = nullptr;    // k_ahead runs beside the chain and the emission
    std::vector<hipStream_t> spare_streams; // candidates the queue probe did not give a role (destroyed with the handle)
    hipStream_t probed_for = nullptr;       // the caller's stream the pipelined roles were last probed against
    bool probed = false;
    hipStream_t tabs = nullptr;     // the emission tables (k_ctot, k_cframe, k_coff) of batch b + 1 beside the event writer on batch b
    hipStream_t side2 = nullptr;    // the event writer of the odd batches (two batches' rows are written side by side)
    std::vector<hipEvent_t> ev_ahead, ev_chain, ev_tab;
    int ch_E = 0;                   // frames per k_ahead launch / emission batch (a multiple of ch_K)
    int last_kind = -1, last_fpl = 0, last_fpb = 0; // v2e_emu_last_pipeline
    uint32_t *ch_gM = nullptr;      // [ch_launch_cap][ch_K + 1][n_clips][ch_K]
    unsigned *ch_bar = nullptr;     // [ch_launch_cap][2 ch_K][n_clips]: one counter per rendezvous of a launch (passes + lock-step frames)
    void *ch_base2 = nullptr, *ch_lp2 = nullptr; // second set of state planes (ping-pong between launches)
    float *ch_ts2 = nullptr;
    CFrame *ch_cf = nullptr;        // [2][ch_E][n_clips]
    unsigned *ch_cdone = nullptr;   // [2][ch_E][n_clips] k_cframe's per-frame completion counters
    uint32_t *ch_cT = nullptr, *ch_ckbase = nullptr, *ch_cperm = nullptr, *ch_cpre = nullptr;
    int ch_long_batches = -1;      // large grids: 64-frame emission batches (1) or max(K, 8) (0: the ring would not fit); -1: not decided yet
    uint32_t *ch_cpre16 = nullptr; // every 16th entry of ch_cpre (CEmitArgs::cpre16)
    uint32_t *ch_cmask = nullptr; // the pull's per-(group, key) pixel ballots (CEmitArgs::cmask), three sets like the other tables
    int ch_nkeys_cap = 0;           // nkeys_cap the chain scratch was sized for
    int ch_pull = 0;                // the event writer of this configuration: k_cpull (1) or k_cemit
    // ---- two scratch sets for OVERLAPPED runs (v2e_emu_run, use_graph | 1024): everything a run's head (uploads, zero fills, first
    // records), its chain and its last emission batches touch while the run before / after it is still in flight exists twice; the
    // fields above / below name the CURRENT set, `alt` holds the other one, swap_scratch() exchanges them before a run is enqueued.
    // (Not doubled: the pixel state and its ping-pong set, the checkpoints: only the chain touches them, and chains never overlap.)
    static constexpr int NPF = 21;
    void *alt[NPF] = {};
    void **pf(int i)
    {
        void **t[NPF] = {(void **)&run_ctl, (void **)&run_fidx, (void **)&run_frames, (void **)&stamp_slot, (void **)&run_off,
                         (void **)&ch_gM, (void **)&ch_bar, (void **)&ch_rec, (void **)&ch_cnt, (void **)&ch_ruleM, (void **)&ch_tsold,
                         (void **)&ch_wmax, (void **)&ch_wtot, (void **)&ch_cf, (void **)&ch_cdone, (void **)&ch_cT, (void **)&ch_ckbase,
                         (void **)&ch_cperm, (void **)&ch_cpre, (void **)&ch_cmask, (void **)&ch_cpre16};
        return t[i];
    }
    void *&alt_of(void **field)
    {
        for (int i = 0; i < NPF; ++i) if (pf(i) == field) return alt[i];
        static void *none = nullptr;
        return none;
    }
    int scratch_par = 0;
    void swap_scratch()
    {
        for (int i = 0; i < NPF; ++i) std::swap(*pf(i), alt[i]);
        scratch_par ^= 1;
    }
    // what is sized by the chain configuration (K, E, keys), in both sets
    void free_chain_scratch()
    {
        for (int i = 5; i < NPF; ++i) { // from ch_gM on (run_ctl / run_fidx / run_frames / stamp_slot / run_off have their own sizes)
            hipFree(*pf(i)); hipFree(alt[i]);
            *pf(i) = nullptr; alt[i] = nullptr;
        }
        hipFree(ch_ck); ch_ck = nullptr;
    }
    // overlapped runs in flight: events of the pieces
    hipEvent_t ev_main_done2[2] = {nullptr, nullptr}, ev_tail_done2[2] = {nullptr, nullptr};
    bool piece_pending[2] = {false, false}; // set `par` has a run whose pieces may still be executing
    const void *last_events = nullptr, *last_recs = nullptr; // buffers of the overlapped run enqueued last
    hipEvent_t ev_user = nullptr;
    void *recs_host[2] = {nullptr, nullptr}; // pinned: the records of the last pipelined run per scratch set (v2e_emu_run_recs)
    size_t recs_host_cap[2] = {0, 0}, recs_host_n[2] = {0, 0};
    int last_ticket = -1; // the scratch set of the run enqueued last if it went out in pieces, else -1 (v2e_emu_run_ticket)
    // every piece of every overlapped run has completed (host-blocking; before scratch is freed or re-sized)
    void sync_runs()
    {
        for (int q = 0; q < 2; ++q)
            if (piece_pending[q]) { if (ev_tail_done2[q]) hipEventSynchronize(ev_tail_done2[q]); if (ev_main_done2[q]) hipEventSynchronize(ev_main_done2[q]); piece_pending[q] = false; }
    }
    // `s` waits for every piece of every overlapped run enqueued so far (no host block)
    int join_runs(hipStream_t s)
    {
        for (int q = 0; q < 2; ++q)
            if (piece_pending[q]) {
                if (hipStreamWaitEvent(s, ev_main_done2[q], 0) != hipSuccess || hipStreamWaitEvent(s, ev_tail_done2[q], 0) != hipSuccess) return -1;
            }
        return 0;
    }
    int occ_cache[32];              // workgroups of a k_chain instantiation a CU holds (-1: not queried yet)
};

static thread_local char g_err[512] = "";

void v2e_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// pinned host staging -> device (v2e_emu_run): the run's frame times, from which the frame scalars (FrameCtl: the reference's Python
// doubles, the per-iteration-count timestamp tables, the refractory switch) are computed HERE, one thread per frame -- the same IEEE
// operations as on the host (make_ctl is one function for both; device float64 / float32 division is correctly rounded, and
// -ffp-contract=off holds for this file): 64 divisions per frame were ~100 us of host time per 300-frame run, on the path that
// bounds the pipelined loop -- and its first frame index, frames' address and stamp slot
static __global__ __launch_bounds__(BLOCK) void k_upload_ctl(const double *__restrict__ tp, const double *__restrict__ tf, FrameCtl *__restrict__ dst, size_t nct,
                                                      double cutoff_hz, double shot_rate_hz, double refr_s,
                                                      const uint32_t *__restrict__ fsrc, uint32_t *__restrict__ fdst,
                                                      const void *const *__restrict__ psrc, const void **__restrict__ pdst,
                                                      unsigned long long *const *__restrict__ ssrc, unsigned long long **__restrict__ sdst)
{
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < nct; i += (size_t)gridDim.x * BLOCK) dst[i] = make_ctl(tp[i], tf[i], cutoff_hz, shot_rate_hz, refr_s);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *fdst = *fsrc; *pdst = *psrc; *sdst = *ssrc; }
    // the run's slot of per-launch time stamps (v2e_emu_launch_stamps; nullptr: off) starts out zero
    unsigned long long *slot = *ssrc;
    if (slot && blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * STAMP_LAUNCHES; i += BLOCK) slot[i] = 0ull;
}

static KArgs make_kargs(const v2e_emu *h, const v2e_emu_params *p)
{
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.W = h->W; a.npx = h->npx; a.nwaves = h->nwaves; a.nkeys_cap = h->nkeys_cap; a.max_iters = h->max_iters;
    a.npx_pad = h->npx_pad;
    a.scalar_thres = p->scalar_thres; a.rng_mode = p->rng_mode; a.shuffle = p->shuffle;
    a.do_leak = p->leak_rate_hz > 0; a.do_shot = p->shot_noise_rate_hz > 0;
    a.has_cutoff = p->cutoff_hz > 0; a.has_refr = p->refractory_period_s > 0;
    a.use_inten = a.has_cutoff || a.do_shot;
    a.log_input = p->log_input != 0;
    a.cutoff_two_pi = M_PI * 2 * p->cutoff_hz;
    a.pos_div = p->pos_thres_scalar; a.neg_div = p->neg_thres_scalar;
    a.pos_nom_f = (float)p->pos_thres_nominal; a.neg_nom_f = (float)p->neg_thres_nominal;
    a.pos_pre_scalar = p->pos_pre_scalar; a.neg_pre_scalar = p->neg_pre_scalar;
    a.leak_hz_f = (float)p->leak_rate_hz; a.jit_f = (float)p->leak_jitter_fraction;
    a.sigma_f = (float)p->sigma_thres;
    a.pos_mean_f = (float)p->pos_thres_scalar; a.neg_mean_f = (float)p->neg_thres_scalar;
    a.ln10cov_f = (float)(log(10.0) * p->noise_rate_cov_decades);
    a.shot_half_rate = p->shot_noise_rate_hz / 2;
    a.inten_slope = p->shot_noise_inten_factor - 1;
    a.refr = p->refractory_period_s; a.refr_f = (float)p->refractory_period_s;
    a.seed = p->seed;
    a.lp = h->lp; a.base = h->base; a.ts_mem = h->ts_mem;
    a.pos_thres = h->pos_thres; a.neg_thres = h->neg_thres; a.noise_rate = h->noise_rate;
    a.cnt = h->cnt; a.hist = h->hist; a.tot = h->tot;
    a.lut_L = h->lut_L; a.lut_I = h->lut_I;
    if (p->photoreceptor_noise) { a.pn_arr = h->pn_arr; a.pn_tape = h->pn_tape; a.pn_vrms_f = (float)p->photoreceptor_noise_vrms; }
    a.cs_sur = h->cs_sur;
    if (h->sc_hp) { a.sc_hp = h->sc_hp; a.sc_prev = h->sc_prev; a.sc_tau = h->sc_tau; a.sc_first_frame = h->sc_first_frame; }
    a.dbg_lognew = h->dbg_lognew; a.dbg_cms = h->dbg_cms; a.dbg_diff = h->dbg_diff;
    return a;
}

static int check_params(const v2e_emu *h, const v2e_emu_params *p)
{
    V2E_REQUIRE(h && p, "null handle/params");
    V2E_REQUIRE(h->lp && h->base && h->pos_thres && h->neg_thres, "state not bound (v2e_emu_bind_state)");
    V2E_REQUIRE((p->f64_state != 0) == (p->cutoff_hz > 0 || p->log_input != 0), "f64_state must equal (cutoff_hz > 0 || log_input)");
    V2E_REQUIRE(!(p->leak_rate_hz > 0) || h->noise_rate, "leak enabled but noise_rate plane not bound");
    V2E_REQUIRE(!(p->refractory_period_s > 0) || h->ts_mem, "refractory enabled but ts_mem plane not bound");
    return 0;
}

extern "C" {

const char *v2e_last_error(void) { return g_err; }
int v2e_version(void) { return 100; }

int64_t v2e_emu_npx_pad(int H, int W)
{
    int64_t npx = (int64_t)H * W;
    return (npx + 255) / 256 * 256;
}

static int alloc_iter_scratch(v2e_emu *h, int max_iters)
{
    if (h->hist) { V2E_HIP(hipFree(h->hist)); h->hist = nullptr; }
    if (h->tot) { V2E_HIP(hipFree(h->tot)); h->tot = nullptr; }
    h->max_iters = max_iters;
    h->nkeys_cap = 2 * max_iters + 2;
    V2E_HIP(hipMalloc(&h->hist, sizeof(uint32_t) * (size_t)h->n_clips * h->nkeys_cap * h->nwaves));
    V2E_HIP(hipMalloc(&h->tot, sizeof(uint32_t) * (size_t)h->n_clips * h->nkeys_cap));
    h->drop_graphs();
    return 0;
}

int v2e_emu_create(int H, int W, int n_clips, int max_iters, int device, v2e_emu **out)
{
    V2E_REQUIRE(out && H > 0 && W > 0 && n_clips > 0 && max_iters > 0, "bad create args");
    V2E_REQUIRE((int64_t)H * W < (1ll << 31) - 256, "sensor too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        v2e_set_error("no HIP device visible");
        return V2E_ENODEV;
    }
    V2E_REQUIRE(device >= 0 && device < ndev, "bad device index");
    V2E_HIP(hipSetDevice(device));
    v2e_emu *h = new v2e_emu();
    h->H = H; h->W = W; h->n_clips = n_clips; h->device = device;
    h->npx = H * W;
    h->npx_pad = v2e_emu_npx_pad(H, W);
    h->nwaves = (h->npx + WAVE - 1) / WAVE;
    h->ngroups = (h->npx + BLOCK - 1) / BLOCK;
    for (int &v : h->occ_cache) v = -1;
    V2E_HIP(hipMalloc(&h->cnt, sizeof(uint32_t) * (size_t)n_clips * h->npx_pad));
    V2E_HIP(hipMalloc(&h->lut_L, sizeof(float) * 256));
    V2E_HIP(hipMalloc(&h->lut_I, sizeof(double) * 256));
    k_lut<<<1, 256>>>(h->lut_L, h->lut_I);
    V2E_HIP(hipDeviceSynchronize());
    {
        int ncu = 0;
        V2E_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        h->n_cu = ncu;
    }
    V2E_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    int rc = alloc_iter_scratch(h, max_iters);
    if (rc) return rc;
    V2E_HIP(hipMalloc(&h->rec_ring, sizeof(v2e_frame_rec) * RING * n_clips));
    V2E_HIP(hipMemset(h->rec_ring, 0, sizeof(v2e_frame_rec) * RING * n_clips));
    V2E_HIP(hipMalloc(&h->ctl_ring, sizeof(FrameCtl) * RING * n_clips));
    V2E_HIP(hipHostMalloc(&h->ctl_host, sizeof(FrameCtl) * RING * n_clips));
    V2E_HIP(hipMalloc(&h->off_dev, sizeof(unsigned long long) * n_clips));
    V2E_HIP(hipMalloc(&h->off_zero, sizeof(unsigned long long) * n_clips));
    V2E_HIP(hipMemset(h->off_zero, 0, sizeof(unsigned long long) * n_clips));
    V2E_HIP(hipHostMalloc(&h->off_host, sizeof(unsigned long long) * n_clips));
    V2E_HIP(hipMalloc(&h->run_fidx, sizeof(uint32_t)));
    V2E_HIP(hipHostMalloc(&h->run_fidx_host, 2 * sizeof(uint32_t)));
    V2E_HIP(hipMalloc(&h->run_frames, sizeof(void *)));
    V2E_HIP(hipHostMalloc(&h->run_frames_host, 2 * sizeof(void *)));
    V2E_HIP(hipMalloc(&h->stamp_slot, sizeof(void *)));
    V2E_HIP(hipMemset(h->stamp_slot, 0, sizeof(void *)));
    V2E_HIP(hipHostMalloc(&h->stamp_slot_host, 2 * sizeof(void *)));
    h->stamp_slot_host[0] = h->stamp_slot_host[1] = nullptr;
    // the second scratch set's copies of the run's device variables (overlapped runs)
    V2E_HIP(hipMalloc(&h->alt_of((void **)&h->run_fidx), sizeof(uint32_t)));
    V2E_HIP(hipMalloc(&h->alt_of((void **)&h->run_frames), sizeof(void *)));
    V2E_HIP(hipMalloc(&h->alt_of((void **)&h->stamp_slot), sizeof(void *)));
    V2E_HIP(hipMemset(h->alt_of((void **)&h->stamp_slot), 0, sizeof(void *)));
    V2E_HIP(hipDeviceSynchronize()); // the fills above are ordered on the default stream only; the caller's stream may be any
    *out = h;
    return 0;
}

int v2e_emu_destroy(v2e_emu *h)
{
    if (!h) return 0;
    hipSetDevice(h->device);
    h->sync_runs();
    hipDeviceSynchronize();
    h->drop_graphs();
    for (int i = 0; i < v2e_emu::NPF; ++i) hipFree(h->alt[i]);
    if (h->ev_user) hipEventDestroy(h->ev_user);
    for (int q = 0; q < 2; ++q) if (h->recs_host[q]) hipHostFree(h->recs_host[q]);
    for (int q = 0; q < 2; ++q) {
        if (h->ev_main_done2[q]) hipEventDestroy(h->ev_main_done2[q]);
        if (h->ev_tail_done2[q]) hipEventDestroy(h->ev_tail_done2[q]);
    }

    hipFree(h->cnt); hipFree(h->hist); hipFree(h->tot); hipFree(h->rec_ring); hipFree(h->ctl_ring);
    hipFree(h->lut_L); hipFree(h->lut_I); hipFree(h->run_off);
    for (hipEvent_t e : h->ev_fork) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_join) hipEventDestroy(e);
    if (h->side) hipStreamDestroy(h->side);
    hipFree(h->ch_cnt); hipFree(h->ch_ruleM); hipFree(h->ch_wmax); hipFree(h->ch_wtot); hipFree(h->ch_tsold); hipFree(h->ch_ck);
    hipFree(h->ch_gM); hipFree(h->ch_bar); hipFree(h->ch_rec); hipFree(h->csr_slots);
    for (hipEvent_t e : h->ev_ahead) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_chain) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_tab) hipEventDestroy(e);
    for (hipStream_t q : h->spare_streams) hipStreamDestroy(q);
    if (h->ahead) hipStreamDestroy(h->ahead);
    if (h->tabs) hipStreamDestroy(h->tabs);
    if (h->side2) hipStreamDestroy(h->side2);
    hipFree(h->ch_base2); hipFree(h->ch_lp2); hipFree(h->ch_ts2); hipFree(h->ch_cf); hipFree(h->ch_cT); hipFree(h->ch_ckbase);
    hipFree(h->ch_cperm); hipFree(h->ch_cpre); hipFree(h->ch_cdone); hipFree(h->ch_cmask); hipFree(h->ch_cpre16);
    if (h->ctl_host) hipHostFree(h->ctl_host);
    hipFree(h->off_dev);
    if (h->off_host) hipHostFree(h->off_host);
    if (h->run_ctl) hipFree(h->run_ctl);
    for (int q = 0; q < 2; ++q) { if (h->run_ctl_host2[q]) hipHostFree(h->run_ctl_host2[q]); if (h->ev_stage[q]) hipEventDestroy(h->ev_stage[q]); }
    hipFree(h->run_fidx);
    if (h->run_fidx_host) hipHostFree(h->run_fidx_host);
    hipFree(h->run_frames);
    if (h->run_frames_host) hipHostFree(h->run_frames_host);
    hipFree(h->stamp_slot); hipFree(h->stamps);
    if (h->stamp_slot_host) hipHostFree(h->stamp_slot_host);
    hipFree(h->fr_dev); hipFree(h->off_zero);
    if (h->fr_stage) hipHostFree(h->fr_stage);
    if (h->fr_par) hipHostFree(h->fr_par);
    if (h->fr_rec_host) hipHostFree(h->fr_rec_host);
    if (h->fr_ev_host) hipHostFree(h->fr_ev_host);
    delete h;
    return 0;
}

int v2e_emu_bind_state(v2e_emu *h, void *lp, void *base, float *ts_mem, float *pos_thres, float *neg_thres,
                       float *noise_rate)
{
    V2E_REQUIRE(h && lp && base && pos_thres && neg_thres, "bind_state: null plane");
    h->lp = lp; h->base = base; h->ts_mem = ts_mem;
    h->pos_thres = pos_thres; h->neg_thres = neg_thres; h->noise_rate = noise_rate;
    h->drop_graphs();
    return 0;
}

#define DISPATCH_FT(dtype, ...)                                          \
    switch (dtype) {                                                     \
    case V2E_DT_U8: { typedef uint8_t FT; __VA_ARGS__; } break;          \
    case V2E_DT_F32: { typedef float FT; __VA_ARGS__; } break;           \
    case V2E_DT_F64: { typedef double FT; __VA_ARGS__; } break;          \
    default: v2e_set_error("bad frame dtype %d", dtype); return V2E_EINVAL; \
    }

int v2e_emu_init_state(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype, double t_frame,
                       const float *thres_pos, const float *thres_neg, const float *noise_rate, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(frame, "null frame");
    if (p->rng_mode == V2E_RNG_TAPE) {
        V2E_REQUIRE(p->scalar_thres || (thres_pos && thres_neg), "tape mode needs threshold draws");
        V2E_REQUIRE(!(p->leak_rate_hz > 0) || noise_rate, "tape mode needs noise_rate values");
    }
    V2E_HIP(hipSetDevice(h->device));
    KArgs a = make_kargs(h, p);
    dim3 grid(v2e_cdiv(h->npx, BLOCK), h->n_clips);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_FT(dtype, {
        if (p->f64_state) k_init<double, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, t_frame, thres_pos, thres_neg, noise_rate);
        else k_init<float, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, t_frame, thres_pos, thres_neg, noise_rate);
    });
    V2E_HIP(hipGetLastError());
    return 0;
}

static int stage_ctl(v2e_emu *h, const v2e_emu_params *p, uint32_t frame_idx, const double *t_prev, const double *t_frame, hipStream_t s)
{
    const int slot = frame_idx % RING;
    FrameCtl *hc = h->ctl_host + (size_t)slot * h->n_clips;
    for (int c = 0; c < h->n_clips; ++c) hc[c] = make_ctl(t_prev[c], t_frame[c], p->cutoff_hz, p->shot_noise_rate_hz, p->refractory_period_s);
    V2E_HIP(hipMemcpyAsync(h->ctl_ring + (size_t)slot * h->n_clips, hc, sizeof(FrameCtl) * h->n_clips,
                           hipMemcpyHostToDevice, s));
    return 0;
}

static int launch_count(v2e_emu *h, const KArgs &a, int f64_state, const void *frame, int dtype, const FrameCtl *ctl,
                        const uint32_t *fidx_base, uint32_t fidx_off, const float *leak, const float *shot,
                        v2e_frame_rec *rec, hipStream_t s, const FrameCtl *ctl_host = nullptr, FrameCtl *ctl_copy = nullptr,
                        double t_prev_v = 0.0, double t_frame_v = 0.0)
{
    dim3 grid(v2e_cdiv(h->npx, BLOCK), h->n_clips);
    DISPATCH_FT(dtype, {
        if (f64_state) k_count<double, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, ctl, fidx_base, fidx_off, leak, shot, rec, ctl_host, ctl_copy, t_prev_v, t_frame_v);
        else k_count<float, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, ctl, fidx_base, fidx_off, leak, shot, rec, ctl_host, ctl_copy, t_prev_v, t_frame_v);
    });
    return 0;
}

int v2e_emu_set_scidvs(v2e_emu *h, void *highpass, void *previous_photo, float *tau, uint32_t first_frame_idx)
{
    V2E_REQUIRE(h && ((highpass && previous_photo && tau) || (!highpass && !previous_photo && !tau)), "set_scidvs: all three planes or none");
    h->sc_hp = highpass; h->sc_prev = previous_photo; h->sc_tau = tau; h->sc_first_frame = first_frame_idx;
    h->drop_graphs();
    return 0;
}

int v2e_emu_set_csdvs(v2e_emu *h, const void *surround)
{
    V2E_REQUIRE(h, "null");
    h->cs_sur = surround;
    h->drop_graphs();
    return 0;
}

int v2e_emu_lp_preview(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype, const double *t_prev, const double *t_frame,
                       uint32_t frame_idx, void *lp_out, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(frame && t_prev && t_frame && lp_out, "null");
    V2E_REQUIRE(dtype == V2E_DT_U8 || dtype == V2E_DT_F32 || dtype == V2E_DT_F64, "bad frame dtype");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = stage_ctl(h, p, frame_idx, t_prev, t_frame, s); // the frame's own slot: v2e_emu_count stages the same values again
    if (rc) return rc;
    const FrameCtl *ctl = h->ctl_ring + (size_t)(frame_idx % RING) * h->n_clips;
    KArgs a = make_kargs(h, p);
    dim3 grid(v2e_cdiv(h->npx, BLOCK), h->n_clips);
    DISPATCH_FT(dtype, {
        if (p->f64_state) k_cs_lp<double, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, ctl, (double *)lp_out);
        else k_cs_lp<float, FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, ctl, (float *)lp_out);
    });
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_emu_set_csdvs_run(v2e_emu *h, void *h_scratch, void *lp_scratch, const double *alpha_p, const double *alpha_h,
                          const int *num_steps, int n_frames, double max_change_to_stop, int *steps_taken_dev)
{
    V2E_REQUIRE(h, "null handle");
    h->csr_ap.clear(); h->csr_ah.clear(); h->csr_steps.clear();
    if (n_frames <= 0) return 0;
    V2E_REQUIRE(h_scratch && lp_scratch && alpha_p && alpha_h && num_steps && steps_taken_dev, "null");
    int mx = 0;
    for (int f = 0; f < n_frames; ++f) {
        V2E_REQUIRE(num_steps[f] >= 0 && num_steps[f] <= 65536, "CSDVS steps per frame out of range for a device-resident run");
        mx = num_steps[f] > mx ? num_steps[f] : mx;
    }
    V2E_HIP(hipSetDevice(h->device));
    if (mx + 1 > h->csr_slots_cap) {
        V2E_HIP(hipDeviceSynchronize());
        if (h->csr_slots) V2E_HIP(hipFree(h->csr_slots));
        h->csr_slots_cap = mx + 1;
        V2E_HIP(hipMalloc((void **)&h->csr_slots, sizeof(unsigned long long) * (size_t)h->csr_slots_cap));
    }
    h->csr_scratch = h_scratch; h->csr_lp = lp_scratch; h->csr_thr = max_change_to_stop; h->csr_steps_dev = steps_taken_dev;
    h->csr_ap.assign(alpha_p, alpha_p + n_frames); h->csr_ah.assign(alpha_h, alpha_h + n_frames);
    h->csr_steps.assign(num_steps, num_steps + n_frames);
    return 0;
}

int v2e_emu_set_model_state_planes(v2e_emu *h, double *log_new_frame, double *c_minus_s_frame, double *diff_frame)
{
    V2E_REQUIRE(h, "null handle");
    V2E_REQUIRE((log_new_frame && c_minus_s_frame && diff_frame) || (!log_new_frame && !c_minus_s_frame && !diff_frame),
                "model-state planes: all three or none");
    h->dbg_lognew = log_new_frame; h->dbg_cms = c_minus_s_frame; h->dbg_diff = diff_frame;
    return 0;
}

int v2e_emu_set_pnoise(v2e_emu *h, void *pn_arr, const float *randn_tape)
{
    V2E_REQUIRE(h, "null");
    h->pn_arr = pn_arr;
    h->pn_tape = randn_tape;
    return 0;
}

int v2e_emu_count(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype, const double *t_prev,
                  const double *t_frame, uint32_t frame_idx, const float *leak_randn, const float *shot_rand,
                  void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(frame && t_prev && t_frame, "null frame/time");
    if (p->rng_mode == V2E_RNG_TAPE) V2E_REQUIRE(!(p->leak_rate_hz > 0) || leak_randn, "tape mode needs leak_randn");
    if (p->photoreceptor_noise) {
        V2E_REQUIRE(h->pn_arr && p->f64_state && p->cutoff_hz > 0, "photoreceptor noise needs its plane (v2e_emu_set_pnoise) and a cutoff");
        V2E_REQUIRE(!(p->shot_noise_rate_hz > 0), "photoreceptor noise replaces shot events: pass shot_noise_rate_hz = 0");
        V2E_REQUIRE(p->rng_mode != V2E_RNG_TAPE || h->pn_tape, "tape mode needs the photoreceptor-noise draws");
    }
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    rc = stage_ctl(h, p, frame_idx, t_prev, t_frame, s);
    if (rc) return rc;
    const int slot = frame_idx % RING;
    v2e_frame_rec *rec = h->rec_ring + (size_t)slot * h->n_clips;
    V2E_HIP(zero_async(rec, sizeof(v2e_frame_rec) * h->n_clips, s));
    KArgs a = make_kargs(h, p);
    rc = launch_count(h, a, p->f64_state, frame, dtype, h->ctl_ring + (size_t)slot * h->n_clips, nullptr, frame_idx,
                      leak_randn, shot_rand, rec, s);
    if (rc) return rc;
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_emu_shot(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype, uint32_t frame_idx,
                 const float *shot_rand, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(frame && shot_rand, "null frame/shot_rand");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int slot = frame_idx % RING;
    KArgs a = make_kargs(h, p);
    dim3 grid(v2e_cdiv(h->npx, BLOCK), h->n_clips);
    DISPATCH_FT(dtype, { k_shot<FT><<<grid, BLOCK, 0, s>>>(a, (const FT *)frame, h->ctl_ring + (size_t)slot * h->n_clips, shot_rand); });
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_emu_read_rec(v2e_emu *h, uint32_t frame_idx, v2e_frame_rec *recs_host, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    V2E_REQUIRE(h && recs_host, "null");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int slot = frame_idx % RING;
    V2E_HIP(hipMemcpyAsync(recs_host, h->rec_ring + (size_t)slot * h->n_clips, sizeof(v2e_frame_rec) * h->n_clips,
                           hipMemcpyDeviceToHost, s));
    V2E_HIP(hipStreamSynchronize(s));
    return 0;
}

int v2e_emu_reserve_iters(v2e_emu *h, int max_events, void *stream)
{
    V2E_REQUIRE(h, "null");
    if (max_events <= h->max_iters) return 0;
    V2E_HIP(hipSetDevice(h->device));
    V2E_HIP(hipStreamSynchronize((hipStream_t)stream));
    int want = h->max_iters;
    while (want < max_events) want *= 2;
    return alloc_iter_scratch(h, want);
}

int v2e_emu_rank(v2e_emu *h, const v2e_emu_params *p, uint32_t frame_idx, const float *ts_table, int n_ts,
                 void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int slot = frame_idx % RING;
    KArgs a = make_kargs(h, p);
    const FrameCtl *ctl = h->ctl_ring + (size_t)slot * h->n_clips;
    v2e_frame_rec *rec = h->rec_ring + (size_t)slot * h->n_clips;
    dim3 grid(v2e_cdiv((int64_t)h->nwaves * WAVE, BLOCK), h->n_clips);
    k_rank<<<grid, BLOCK, 0, s>>>(a, ctl, rec, ts_table, n_ts);
    k_scan<<<dim3(SCAN_BLOCKS, h->n_clips), BLOCK, 0, s>>>(a, rec);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_emu_emit(v2e_emu *h, const v2e_emu_params *p, uint32_t frame_idx, const float *ts_table, int n_ts,
                 float *events, uint64_t cap, const uint64_t *ev_offset0, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(events || cap == 0, "null events");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int slot = frame_idx % RING;
    KArgs a = make_kargs(h, p);
    const FrameCtl *ctl = h->ctl_ring + (size_t)slot * h->n_clips;
    v2e_frame_rec *rec = h->rec_ring + (size_t)slot * h->n_clips;
    unsigned long long *d_off = nullptr;
    if (ev_offset0) { // host array -> pinned staging -> device (frame-at-a-time API syncs every frame)
        for (int c = 0; c < h->n_clips; ++c) h->off_host[c] = ev_offset0[c];
        d_off = h->off_dev;
        V2E_HIP(hipMemcpyAsync(d_off, h->off_host, sizeof(unsigned long long) * h->n_clips, hipMemcpyHostToDevice, s));
    }
    const v2e_frame_rec *rec_prev = h->rec_ring + (size_t)((frame_idx + RING - 1) % RING) * h->n_clips;
    dim3 grid(v2e_cdiv((int64_t)h->nwaves * WAVE, BLOCK), h->n_clips);
    if (p->f64_state) k_emit<double><<<grid, BLOCK, 0, s>>>(a, ctl, rec, rec_prev, d_off, nullptr, frame_idx, ts_table, n_ts, (float4 *)events, cap);
    else k_emit<float><<<grid, BLOCK, 0, s>>>(a, ctl, rec, rec_prev, d_off, nullptr, frame_idx, ts_table, n_ts, (float4 *)events, cap);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_emu_read_iter_counts(v2e_emu *h, uint32_t frame_idx, int n_iters, uint32_t *counts_host, void *stream)
{
    (void)frame_idx;
    V2E_REQUIRE(h && counts_host && n_iters >= 0 && n_iters <= h->max_iters, "bad read_iter_counts args");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int nk = 2 * (n_iters + 1);
    for (int c = 0; c < h->n_clips; ++c)
        V2E_HIP(hipMemcpyAsync(counts_host + (size_t)c * nk, h->tot + (size_t)c * h->nkeys_cap, sizeof(uint32_t) * nk,
                               hipMemcpyDeviceToHost, s));
    V2E_HIP(hipStreamSynchronize(s));
    return 0;
}

int v2e_emu_permute(v2e_emu *h, const float *events_in, float *events_out, const int32_t *idx, uint64_t row0,
                    uint64_t n, void *stream)
{
    V2E_REQUIRE(h && events_in && events_out && idx, "null");
    if (n == 0) return 0;
    V2E_HIP(hipSetDevice(h->device));
    k_permute<<<v2e_cdiv((int64_t)n, BLOCK), BLOCK, 0, (hipStream_t)stream>>>((const float4 *)events_in, (float4 *)events_out,
                                                                              idx, row0, n);
    V2E_HIP(hipGetLastError());
    return 0;
}

// One frame of the frame-at-a-time API in ONE call (Philox mode, one clip): what a v2e.py caller of generate_events pays
// per frame used to be five C calls with three stream synchronisations between them.  Here: frame -> pinned staging ->
// device, count / rank / scan enqueued back to back, ONE read-back of the frame record and the per-key totals (which
// gives M and the event count), emit, ONE read-back of exactly that many rows into pinned memory.
// Returns 0, or 1 when the frame needs more iteration scratch than the handle has (M > max_iters; the caller grows it
// with v2e_emu_reserve_iters and finishes the frame with v2e_emu_rank / v2e_emu_emit: the count is done and stays valid),
// or 2 when the rows do not fit `cap` (same: nothing was emitted).
int v2e_emu_frame(v2e_emu *h, const v2e_emu_params *p, const void *frame, int frame_on_host, int dtype, double t_prev, double t_frame,
                  uint32_t frame_idx, float *events_dev, uint64_t cap, uint32_t *out8, const float **events_host, void *stream)
{
    if (h && (h->piece_pending[0] || h->piece_pending[1]) && h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(frame && out8 && events_host && events_dev, "null");
    V2E_REQUIRE(h->n_clips == 1 && p->rng_mode == V2E_RNG_PHILOX && !p->photoreceptor_noise, "v2e_emu_frame: one clip, Philox mode, no photoreceptor noise");
    V2E_REQUIRE(dtype == V2E_DT_U8 || dtype == V2E_DT_F32 || dtype == V2E_DT_F64, "bad frame dtype");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == V2E_DT_U8 ? 1 : (dtype == V2E_DT_F32 ? 4 : 8);
    const size_t fbytes = esz * (size_t)h->npx;
    const auto tp0 = std::chrono::steady_clock::now();
    // (see k_frame_begin: no copy engine on the way in or out)
    if (frame_on_host && fbytes > h->fr_bytes) {
        V2E_HIP(hipStreamSynchronize(s));
        if (h->fr_stage) V2E_HIP(hipHostFree(h->fr_stage));
        h->fr_stage = nullptr; h->fr_bytes = 0;
        V2E_HIP(hipHostMalloc((void **)&h->fr_stage, fbytes, hipHostMallocMapped));
        V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_stage_dev, h->fr_stage, 0));
        h->fr_bytes = fbytes;
    }
    if (!h->fr_dev) {
        V2E_HIP(hipMalloc((void **)&h->fr_dev, sizeof(FrameScratch)));
        V2E_HIP(hipMemset(h->fr_dev, 0, sizeof(FrameScratch))); // both records zero: the first frame counts into a clean one
        V2E_HIP(hipDeviceSynchronize());                        // (ordered on the default stream only; once per handle)
        V2E_HIP(hipHostMalloc((void **)&h->fr_par, sizeof(FrameCtl), hipHostMallocMapped));
        V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_par_dev, h->fr_par, 0));
        h->fr_flip = 0;
    }
    if (h->fr_rec_keys != h->nkeys_cap) {
        if (h->fr_rec_host) V2E_HIP(hipHostFree(h->fr_rec_host));
        V2E_HIP(hipHostMalloc((void **)&h->fr_rec_host, sizeof(v2e_frame_rec) + sizeof(uint32_t) * h->nkeys_cap, hipHostMallocMapped));
        V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_rec_host_dev, h->fr_rec_host, 0));
        h->fr_rec_keys = h->nkeys_cap;
    }
    const void *frame_dev = frame;
    if (frame_on_host) {
        memcpy(h->fr_stage, frame, fbytes); // (the previous frame's kernels have completed: every call ends synchronised)
        frame_dev = h->fr_stage_dev;
    }
    // destination of the rows: the caller's pinned buffer for this frame (v2e_emu_frame_host_rows; one frame only), else the
    // handle's own
    float *user_rows = h->fr_user_rows, *user_rows_dev = h->fr_user_rows_dev;
    const uint64_t user_cap = h->fr_user_cap;
    h->fr_user_rows = h->fr_user_rows_dev = nullptr; h->fr_user_cap = 0;
    uint64_t est = std::min<uint64_t>(std::max<uint64_t>(h->fr_est, 1024), cap);
    if (user_rows) est = std::min<uint64_t>(est, user_cap);
    if (!user_rows && est > h->fr_ev_cap) {
        V2E_HIP(hipStreamSynchronize(s));
        if (h->fr_ev_host) V2E_HIP(hipHostFree(h->fr_ev_host));
        h->fr_ev_host = nullptr; h->fr_ev_cap = 0;
        const uint64_t want = std::max<uint64_t>(2 * est, 1u << 16);
        V2E_HIP(hipHostMalloc((void **)&h->fr_ev_host, sizeof(float) * 4 * want, hipHostMallocMapped));
        V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_ev_host_dev, h->fr_ev_host, 0));
        h->fr_ev_cap = want;
    }
    // Five launches, no copy engine, no host step between them: the frame's times go to k_count by value, its workgroup 0 copies
    // the frame's scalars from pinned host memory to the device for the kernels behind it; the record a frame counts into was
    // zeroed by the frame before (two records alternate); the event writer goes out before the host has seen the totals
    // (k_emit checks them against cap itself), and with it the copy of the rows the frame is expected to have: one
    // synchronisation per frame unless the estimate was short.  (Replaying the launches as a hipGraph was measured: the host
    // spends 10 us instead of 19 us enqueueing, the graph's start latency leaves the frame's wall time where it was.)
    const FrameCtl ctl_host = make_ctl(t_prev, t_frame, p->cutoff_hz, p->shot_noise_rate_hz, p->refractory_period_s);
    *(FrameCtl *)h->fr_par = ctl_host;
    FrameScratch *sc = (FrameScratch *)h->fr_dev;
    v2e_frame_rec *rec = &sc->rec[h->fr_flip], *rec_next = &sc->rec[h->fr_flip ^ 1];
    h->fr_flip ^= 1;
    FrameCtl *ctl = &sc->ctl;
    KArgs a = make_kargs(h, p);
    a.emit_guard = 1;
    v2e_frame_rec *rh = (v2e_frame_rec *)h->fr_rec_host;
    rc = launch_count(h, a, p->f64_state, frame_dev, dtype, nullptr, nullptr, frame_idx, nullptr, nullptr, rec, s,
                      (const FrameCtl *)h->fr_par_dev, ctl, t_prev, t_frame);
    if (rc) return rc;
    {
        dim3 gridw(v2e_cdiv((int64_t)h->nwaves * WAVE, BLOCK), 1);
        k_rank<<<gridw, BLOCK, 0, s>>>(a, ctl, rec, nullptr, 0);
        k_scan<<<dim3(SCAN_BLOCKS, 1), BLOCK, 0, s>>>(a, rec);
        if (p->f64_state) k_emit<double><<<gridw, BLOCK, 0, s>>>(a, ctl, rec, nullptr, h->off_zero, nullptr, frame_idx, nullptr, 0, (float4 *)events_dev, cap);
        else k_emit<float><<<gridw, BLOCK, 0, s>>>(a, ctl, rec, nullptr, h->off_zero, nullptr, frame_idx, nullptr, 0, (float4 *)events_dev, cap);
        k_frame_rows_to_host<<<256, 256, 0, s>>>((const float4 *)events_dev, rec, (float4 *)(user_rows ? user_rows_dev : h->fr_ev_host_dev),
                                                 (v2e_frame_rec *)h->fr_rec_host_dev, 0ull, est, rec_next);
    }
    V2E_HIP(hipGetLastError());
    constexpr bool timing = false; // (round 2-4 dev switch V2E_AMD_FRAME_TIMING: where a frame's host time goes; removed, the block below is dead code the compiler drops)
    static double t_enq = 0, t_sync = 0; static int t_n = 0;
    const auto tp1 = std::chrono::steady_clock::now();
    V2E_HIP(hipStreamSynchronize(s));
    if (timing) {
        const auto tp2 = std::chrono::steady_clock::now();
        t_enq += std::chrono::duration<double, std::micro>(tp1 - tp0).count();
        t_sync += std::chrono::duration<double, std::micro>(tp2 - tp1).count();
        if (++t_n == 100) { fprintf(stderr, "v2e_emu_frame: enqueue %.1f us, synchronise %.1f us per frame\n", t_enq / 100, t_sync / 100); t_enq = t_sync = 0; t_n = 0; }
    }
    const int M = rh->max_events;
    memset(out8, 0, sizeof(uint32_t) * 8);
    out8[4] = (uint32_t)M;
    *events_host = nullptr;
    if (M > h->max_iters || (rh->flags & V2E_FLAG_EVENTS_DROPPED)) {
        // not emitted (nothing of the state touched): hand the counted frame to the step-wise entry points, which read
        // the ring's record and scalars
        const int slot = frame_idx % RING;
        V2E_HIP(hipMemcpyAsync(h->rec_ring + slot, rec, sizeof(v2e_frame_rec), hipMemcpyDeviceToDevice, s));
        V2E_HIP(zero_async(&h->rec_ring[slot].flags, sizeof(uint32_t), s));
        V2E_HIP(hipMemcpyAsync(h->ctl_ring + slot, ctl, sizeof(FrameCtl), hipMemcpyDeviceToDevice, s));
        h->ctl_host[slot] = ctl_host;
        if (M > h->max_iters) { V2E_HIP(hipStreamSynchronize(s)); return 1; }
        uint32_t *th = (uint32_t *)(h->fr_rec_host + sizeof(v2e_frame_rec));
        V2E_HIP(hipMemcpyAsync(th, h->tot, sizeof(uint32_t) * (2 * M + 2), hipMemcpyDeviceToHost, s));
        V2E_HIP(hipStreamSynchronize(s));
        uint64_t n_on = 0, n_off = 0, n_sig = 0;
        for (int k = 0; k < 2 * M + 2; ++k) {
            ((k & 1) ? n_off : n_on) += th[k];
            if (k < 2 * M) n_sig += th[k];
        }
        out8[0] = (uint32_t)(n_on + n_off); out8[1] = (uint32_t)n_on; out8[2] = (uint32_t)n_off; out8[3] = (uint32_t)n_sig;
        return 2;
    }
    const uint64_t n = rh->n_events;
    out8[0] = (uint32_t)n; out8[1] = rh->n_on; out8[2] = rh->n_off; out8[3] = rh->n_signal;
    h->fr_est = n + n / 4 + 256;
    if (user_rows && n > user_cap) { // the caller's buffer cannot hold the frame: all rows into the handle's own
        if (n > h->fr_ev_cap) {
            if (h->fr_ev_host) V2E_HIP(hipHostFree(h->fr_ev_host));
            h->fr_ev_host = nullptr; h->fr_ev_cap = 0;
            const uint64_t want = std::max<uint64_t>(2 * n, 1u << 16);
            V2E_HIP(hipHostMalloc((void **)&h->fr_ev_host, sizeof(float) * 4 * want, hipHostMallocMapped));
            V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_ev_host_dev, h->fr_ev_host, 0));
            h->fr_ev_cap = want;
        }
        k_frame_rows_to_host<<<(unsigned)std::min<uint64_t>(v2e_cdiv((int64_t)n, 256), 4096), 256, 0, s>>>(
            (const float4 *)events_dev, rec, (float4 *)h->fr_ev_host_dev, nullptr, 0ull, n, nullptr);
        V2E_HIP(hipGetLastError());
        V2E_HIP(hipStreamSynchronize(s));
        *events_host = h->fr_ev_host;
        return 0;
    }
    if (user_rows) {
        if (n > est) {
            k_frame_rows_to_host<<<(unsigned)std::min<uint64_t>(v2e_cdiv((int64_t)(n - est), 256), 4096), 256, 0, s>>>(
                (const float4 *)events_dev, rec, (float4 *)user_rows_dev, nullptr, est, n - est, nullptr);
            V2E_HIP(hipGetLastError());
            V2E_HIP(hipStreamSynchronize(s));
        }
        if (n > 0) *events_host = user_rows;
        return 0;
    }
    if (n > est) { // the estimate was short: the rest of the rows
        if (n > h->fr_ev_cap) {
            float *grown = nullptr;
            const uint64_t want = std::max<uint64_t>(2 * n, 1u << 16);
            V2E_HIP(hipHostMalloc((void **)&grown, sizeof(float) * 4 * want, hipHostMallocMapped));
            memcpy(grown, h->fr_ev_host, sizeof(float) * 4 * est);
            V2E_HIP(hipHostFree(h->fr_ev_host));
            h->fr_ev_host = grown; h->fr_ev_cap = want;
            V2E_HIP(hipHostGetDevicePointer((void **)&h->fr_ev_host_dev, h->fr_ev_host, 0));
        }
        k_frame_rows_to_host<<<(unsigned)std::min<uint64_t>(v2e_cdiv((int64_t)(n - est), 256), 4096), 256, 0, s>>>(
            (const float4 *)events_dev, rec, (float4 *)h->fr_ev_host_dev, nullptr, est, n - est, nullptr);
        V2E_HIP(hipGetLastError());
        V2E_HIP(hipStreamSynchronize(s));
    }
    if (n > 0) *events_host = h->fr_ev_host;
    return 0;
}

int v2e_emu_frame_host_rows(v2e_emu *h, float *pinned_rows, uint64_t cap_rows)
{
    V2E_REQUIRE(h, "null");
    h->fr_user_rows = h->fr_user_rows_dev = nullptr; h->fr_user_cap = 0;
    if (!pinned_rows || cap_rows == 0) return 0;
    V2E_HIP(hipSetDevice(h->device));
    void *dev = nullptr;
    V2E_HIP(hipHostGetDevicePointer(&dev, pinned_rows, 0)); // fails for memory that is not pinned (hipHostMalloc / hipHostRegister)
    h->fr_user_rows = pinned_rows; h->fr_user_rows_dev = (float *)dev; h->fr_user_cap = cap_rows;
    return 0;
}

// Enqueue the whole multi-frame sequence on stream s (no host sync).
static int enqueue_run(v2e_emu *h, const v2e_emu_params *p, const KArgs &a, const void *frames, int dtype, int n_frames,
                       float *events, uint64_t cap, v2e_frame_rec *recs, hipStream_t s, hipEvent_t *evs = nullptr)
{
#define V2E_MARK(i) do { if (evs) V2E_HIP(hipEventRecord(evs[(i)], s)); } while (0)
    const size_t esz = dtype == V2E_DT_U8 ? 1 : (dtype == V2E_DT_F32 ? 4 : 8);
    V2E_HIP(zero_async(recs, sizeof(v2e_frame_rec) * (size_t)n_frames * h->n_clips, s));
    dim3 gridw(v2e_cdiv((int64_t)h->nwaves * WAVE, BLOCK), h->n_clips);
    for (int f = 0; f < n_frames; ++f) {
        const void *fr = (const char *)frames + (size_t)f * h->n_clips * h->npx * esz;
        const FrameCtl *ctl = h->run_ctl + (size_t)f * h->n_clips;
        v2e_frame_rec *rec = recs + (size_t)f * h->n_clips;
        const v2e_frame_rec *rec_prev = f > 0 ? rec - h->n_clips : nullptr;
        V2E_MARK(4 * f + 0);
        if (h->cs_sur) { // emulator.py:707-708: the surround is stepped against the coming frame's lp_log_frame, on the stream
            dim3 gridp(v2e_cdiv(h->npx, BLOCK), h->n_clips);
            DISPATCH_FT(dtype, {
                if (p->f64_state) k_cs_lp<double, FT><<<gridp, BLOCK, 0, s>>>(a, (const FT *)fr, ctl, (double *)h->csr_lp);
                else k_cs_lp<float, FT><<<gridp, BLOCK, 0, s>>>(a, (const FT *)fr, ctl, (float *)h->csr_lp);
            });
            int rcs = v2e_csdvs_enqueue_frame(h->csr_lp, const_cast<void *>(h->cs_sur), h->csr_scratch, h->H, h->W, p->f64_state,
                                              h->csr_ap[f], h->csr_ah[f], h->csr_steps[f], h->csr_thr, h->csr_slots, h->csr_steps_dev + f, s);
            if (rcs) return rcs;
        }
        int rc = launch_count(h, a, p->f64_state, fr, dtype, ctl, h->run_fidx, (uint32_t)f, nullptr, nullptr, rec, s);
        if (rc) return rc;
        V2E_MARK(4 * f + 1);
        k_rank<<<gridw, BLOCK, 0, s>>>(a, ctl, rec, nullptr, 0);
        V2E_MARK(4 * f + 2);
        k_scan<<<dim3(SCAN_BLOCKS, h->n_clips), BLOCK, 0, s>>>(a, rec);
        V2E_MARK(4 * f + 3);
        if (p->f64_state) k_emit<double><<<gridw, BLOCK, 0, s>>>(a, ctl, rec, rec_prev, nullptr, h->run_fidx, (uint32_t)f, nullptr, 0, (float4 *)events, cap);
        else k_emit<float><<<gridw, BLOCK, 0, s>>>(a, ctl, rec, rec_prev, nullptr, h->run_fidx, (uint32_t)f, nullptr, 0, (float4 *)events, cap);
    }
    V2E_MARK(4 * n_frames);
#undef V2E_MARK
    V2E_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------ launch scheduling of the chain pipeline
// The run is a DAG over three logical streams (chain, k_ahead + emission tables, event writer) with cross edges between
// all of them.  It is either enqueued on three HIP streams with events, or built as an explicit hipGraph (kernel nodes
// with their dependency lists).  Stream capture is NOT used for it: hipStreamEndCapture of a capture with edges between
// two forked streams segfaults on this runtime (ROCm 7.0 / HIP 7.0.5; round 2 had met the same with a fourth stream).
enum { ST_MAIN = 0, ST_AHEAD = 1, ST_SIDE = 2, ST_TAB = 3, ST_SIDE2 = 4, ST_COUNT = 5 };
enum { EV_FORK = 0, EV_JOIN = 1, EV_AHEAD = 2, EV_CHAIN = 3, EV_TAB = 4, EV_KINDS = 5 };

// dev tool (scripts/step_stamps.py): sequence-numbered device time stamps from inside a run's graph: buf[0] counts, buf[1 + i] = time
struct Sched {
    v2e_emu *h;
    hipStream_t st[ST_COUNT];
    hipGraph_t graph = nullptr; // non-null: build nodes instead of launching
    std::vector<hipGraphNode_t> pos[ST_COUNT];                  // what the next node of a logical stream depends on
    std::vector<std::vector<hipGraphNode_t>> evdeps;     // graph mode: the dependencies an event stands for
    int ev_cap = 0;

    hipEvent_t real_event(int kind, int idx) const
    {
        switch (kind) {
        case EV_FORK: return h->ev_fork[idx];
        case EV_JOIN: return h->ev_join[idx];
        case EV_AHEAD: return h->ev_ahead[idx];
        case EV_CHAIN: return h->ev_chain[idx];
        default: return h->ev_tab[idx];
        }
    }
    static void merge(std::vector<hipGraphNode_t> &dst, const std::vector<hipGraphNode_t> &src)
    {
        for (hipGraphNode_t n : src) {
            bool dup = false;
            for (hipGraphNode_t d : dst) dup = dup || d == n;
            if (!dup) dst.push_back(n);
        }
    }
    int kernel(int s, const void *fn, dim3 grid, dim3 block, size_t lds, void **args)
    {
        if (!graph) {
            V2E_HIP(hipLaunchKernel(fn, grid, block, args, lds, st[s]));
            return 0;
        }
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = const_cast<void *>(fn);
        kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = (unsigned)lds;
        kp.kernelParams = args; kp.extra = nullptr;
        hipGraphNode_t node;
        V2E_HIP(hipGraphAddKernelNode(&node, graph, pos[s].empty() ? nullptr : pos[s].data(), pos[s].size(), &kp));
        pos[s].assign(1, node);
        return 0;
    }
    int zero(int s, void *ptr, size_t bytes)
    {
        size_t n = (bytes + 3) / 4;
        if (n == 0) return 0;
        uint32_t *p32 = (uint32_t *)ptr;
        void *args[] = {(void *)&p32, (void *)&n};
        return kernel(s, (const void *)k_zero_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, args);
    }
    int zero4(int s, void *const ptr[4], const size_t bytes[4])
    {
        ZeroRanges z;
        unsigned long long tot = 0;
        for (int r = 0; r < 4; ++r) { z.p[r] = (uint32_t *)ptr[r]; z.n[r] = ptr[r] ? (bytes[r] + 3) / 4 : 0; tot += z.n[r]; }
        if (tot == 0) return 0;
        void *args[] = {(void *)&z};
        return kernel(s, (const void *)k_zero_ranges, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, args);
    }
    int record(int kind, int idx, int s)
    {
        if (!graph) { V2E_HIP(hipEventRecord(real_event(kind, idx), st[s])); return 0; }
        evdeps[(size_t)kind * ev_cap + idx] = pos[s];
        return 0;
    }
    int wait(int s, int kind, int idx)
    {
        if (!graph) { V2E_HIP(hipStreamWaitEvent(st[s], real_event(kind, idx), 0)); return 0; }
        merge(pos[s], evdeps[(size_t)kind * ev_cap + idx]);
        return 0;
    }
};

// ------------------------------------------------------------------ K frames per launch (emu_chain.h)
static int chain_egroups(const v2e_emu *h) { return (h->npx + GROUP_PX - 1) / GROUP_PX; }
static bool chain_small_grid(const v2e_emu *h) { return (long long)h->ngroups * h->n_clips <= 2ll * h->n_cu; }

// Batches in the ring of frame slots.  Three is the minimum (batch b is read by its emission while the chain is in batch b + 1 and
// k_ahead fills batch b + 2); the chain then waits for the emission of batch b - 3 where the ring wraps.  One hipGraph per run runs
// SLOWER with a deeper ring (round 3: 8.5 Gev/s with 3 batches, 7.3 with 5 or 7; round 6: 787 -> 904 us per step with 5), 64 clips
// and 1280x720 do not care (14.5-14.6 / 11.9-12.1 Gev/s either way) -- but PIPELINED runs of a small grid, whose k_ahead runs a whole
// run ahead and whose event rows have a stream of their own, gain from never waiting inside a 300-frame run: ring 3 681-685 us per
// step, 4 668-671, 5 662-664, 6 657-673, 8 671 (profiles/r06_emulator_experiments.txt item 21).  A handle keeps the depth its FIRST
// device-resident run chose (no re-allocation when a caller mixes modes); V2E_AMD_CHAIN_RING overrides.
static int chain_ring_batches(v2e_emu *h, bool pipelined)
{
    if (const char *ev = getenv("V2E_AMD_CHAIN_RING")) { const int v = atoi(ev); if (v >= 3 && v <= 16) return v; }
    if (h->ch_ring_pref == 0) h->ch_ring_pref = (pipelined && chain_small_grid(h) && h->n_clips == 1) ? 5 : 3;
    return h->ch_ring_pref;
}

// records built inside the chain (large grids) or by k_ahead (small grids); V2E_AMD_CHAIN_FUSED=0/1 overrides (dev / tests)
static bool chain_fused_records(const v2e_emu *h, int dtype, int nD)
{
    if (dtype != V2E_DT_U8) return true; // k_ahead's record path is instantiated for uint8 frames only
    // k_chain reads k_ahead's records and writes the count words through buffer resources (32-bit offsets): the record ring of
    // nD batches of up to 64 frames must stay below 4 GB (1280x720: 2.8 GB with three batches; beyond that the chain builds the records)
    if ((size_t)nD * 64 * h->n_clips * h->npx_pad * sizeof(uint4) >= (1ull << 32)) return true;
    const char *fe = getenv("V2E_AMD_CHAIN_FUSED"); // read per call: tests switch it per emulator instance
    const int fused_env = fe ? atoi(fe) : -1;
    return fused_env >= 0 ? fused_env != 0 : !chain_small_grid(h);
}

static size_t chain_dyn_lds(bool fused) { return fused ? 0 : (size_t)CHAIN_SUB * BLOCK * sizeof(uint4); }

// allon: every run-time feature switch of the frame loop is on (cutoff, leak, shot noise, refractory period: the v2e CLI
// defaults) -> the instantiation without those tests; built for float64 state and uint8 frames (what that configuration has)
static const void *chain_fn(bool f64, int dtype, bool fused, bool allon = false)
{
    if (allon && f64 && dtype == V2E_DT_U8)
        return fused ? (const void *)k_chain<double, uint8_t, true, true> : (const void *)k_chain<double, uint8_t, false, true>;
    if (!fused) return f64 ? (const void *)k_chain<double, uint8_t, false> : (const void *)k_chain<float, uint8_t, false>;
    switch (dtype) {
    case V2E_DT_U8: return f64 ? (const void *)k_chain<double, uint8_t, true> : (const void *)k_chain<float, uint8_t, true>;
    case V2E_DT_F32: return f64 ? (const void *)k_chain<double, float, true> : (const void *)k_chain<float, float, true>;
    default: return f64 ? (const void *)k_chain<double, double, true> : (const void *)k_chain<float, double, true>;
    }
}

// workgroups of the k_chain instantiation that will run which a CU holds (the redo rendezvous needs a clip's workgroups
// co-resident); the occupancy API can over-report by one per CU (MI355X guide), hence the margin.  Queried once per
// (handle = device, instantiation).
static int chain_blocks_per_cu(v2e_emu *h, bool f64, int dtype, bool fused, bool allon = false)
{
    // (the instantiation that will be launched: the ALLON one -- every feature switch on, the v2e CLI defaults -- needs half the
    // registers of the general fused one, so twice the clips of a multi-clip run fit beside one another)
    allon = allon && f64 && dtype == V2E_DT_U8;
    const int key = (f64 ? 1 : 0) | ((dtype & 3) << 1) | (fused ? 8 : 0) | (allon ? 16 : 0);
    if (h->occ_cache[key] >= 0) return h->occ_cache[key];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_fn(f64, dtype, fused, allon), BLOCK, chain_dyn_lds(fused)) != hipSuccess) per_cu = 0;
    if (per_cu > 2) per_cu = std::min(per_cu - 1, 6); // LDS-bound counts (<= 2) are exact
    h->occ_cache[key] = per_cu;
    return per_cu;
}

// Frames per chain launch.  The launch boundary (~4 us) and the prologue are paid once per K frames; a redo (rule-on frame:
// ~1 % of the frames of the benchmark clip) repeats a launch from a checkpoint.  Small grids (bounded by latency) and large
// grids without a refractory period: 32 (1280x720 noisy: 16 frames 6.24, 32 frames 6.49 Gev/s); large grids where a redo is
// possible: 8.  A clip whose workgroups cannot all be resident cannot hold the redo rendezvous: one frame per launch, where
// the one frame a redo fixes is the launch's last and nothing is left to verify (|128 asks for that too: clips on which the
// rule is active on most frames would redo most launches).
static int chain_frames_per_launch(const v2e_emu *h, bool has_refr, int use_graph, int max_blocks)
{
    // (without a refractory period nothing is speculated and nothing redone: the longest launch the kernel supports -- 1280x720 noisy,
    //  pipelined, round 6: 64 frames 13.6-14.1 Gev/s, 32 13.0-13.5, 16 12.6-13.0)
    int K = !has_refr ? CHAIN_K_MAX : (chain_small_grid(h) ? 32 : 8);
    if (const char *ev = getenv("V2E_AMD_CHAIN_K")) { const int v = atoi(ev); if (v >= 1 && v <= CHAIN_K_MAX) K = v; }
    if (has_refr && ((use_graph & 128) || h->ngroups > max_blocks)) K = 1;
    return K;
}

// The launch schedule of a run, as plain data (exported for the CPU tests): per chain launch the frames it advances
// [f0, f0 + nf) and validates [pf0, pf0 + pnf), the emission batch whose completion frees the ring slots it overwrites,
// the k_ahead batch it needs, the k_ahead batch enqueued behind it, the emission batch that is final once it is enqueued.
struct ChainLaunch { int f0, nf, pf0, pnf, wait_join, wait_ahead, ahead_next, emit_batch; };
static std::vector<ChainLaunch> chain_plan(int n_frames, int K, int E, int nD, bool has_refr, bool fused)
{
    const int m = E / K;
    const int nB = (n_frames + K - 1) / K;           // chain launches with frames
    const int nL = has_refr ? nB + 1 : nB;           // + the tail launch that validates the last K frames
    const int nEB = (n_frames + E - 1) / E;          // batches of the parallel kernels
    std::vector<ChainLaunch> plan;
    for (int L = 0; L < nL; ++L) {
        ChainLaunch c;
        const bool tail = L >= nB;
        c.f0 = tail ? n_frames : L * K;
        c.nf = tail ? 0 : std::min((L + 1) * K, n_frames) - L * K;
        c.pf0 = (L - 1) * K;
        c.pnf = (has_refr && L > 0) ? std::min(L * K, n_frames) - (L - 1) * K : 0;
        const bool first = !tail && L % m == 0; // first launch of batch L / m
        c.wait_join = (first && L / m >= nD) ? L / m - nD : -1;
        c.wait_ahead = (first && !fused) ? L / m : -1;
        c.ahead_next = (first && !fused && L / m + 2 < nEB) ? L / m + 2 : -1;
        const int fin = has_refr ? L - 1 : L; // what is final now: with a refractory period the launch just validated
        c.emit_batch = (fin >= 0 && ((fin + 1) % m == 0 || fin == nB - 1)) ? fin / m : -1;
        plan.push_back(c);
    }
    return plan;
}

// scratch of the chain pipeline: everything a captured run must not allocate
// The event writer: k_cpull (a thread per output row; round 5, the default) or k_cemit (a wave per pixel group pushing its rows through
// the bijection).  The pull needs 32 bytes of pixel ballots per (frame of the three table sets, group, key) -- 281 MB at 346x260 with
// the default max_iters = 64, 18 GB for 64 clips: when that would take more than a quarter of the free device memory the push writer
// stays (same rows, bit for bit; V2E_AMD_EMIT_PULL=0 forces it: tests/test_emulator_gpu.py runs both).
static bool emit_pull(size_t table_bytes)
{
    if (const char *e = getenv("V2E_AMD_EMIT_PULL")) { if (atoi(e) == 0) return false; } // (read when a handle's tables are allocated)
    // an absolute budget as well as a share of what is free NOW (round-5 advisor: a quarter of momentarily free memory alone let the
    // choice depend on who allocated first, and 18 GB of tables for 64 clips could starve a SloMo engine built later): 24 GB per
    // scratch set by default, V2E_AMD_PULL_BUDGET_MB overrides; v2e_emu_event_writer says which writer a handle got
    size_t budget = (size_t)24 << 30;
    if (const char *e = getenv("V2E_AMD_PULL_BUDGET_MB")) { const long long v = atoll(e); if (v >= 0) budget = (size_t)v << 20; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return table_bytes <= free_b / 4 && table_bytes <= budget;
}

static int chain_alloc(v2e_emu *h, const v2e_emu_params *p, int dtype, int n_frames, int use_graph)
{
    const bool has_refr = p->refractory_period_s > 0;
    const int nD = chain_ring_batches(h, (use_graph & 1024) != 0 && (use_graph & 3) == 0);
    const bool fused = chain_fused_records(h, dtype, nD);
    const int inst = (p->f64_state ? 1 : 0) | ((dtype & 3) << 1) | (fused ? 8 : 0);
    const bool allon_p = p->cutoff_hz > 0 && p->leak_rate_hz > 0 && p->shot_noise_rate_hz > 0 && has_refr;
    const int max_blocks = chain_blocks_per_cu(h, p->f64_state != 0, dtype, fused, allon_p) * h->n_cu;
    const int K = chain_frames_per_launch(h, has_refr, use_graph, max_blocks);
    // frames per k_ahead launch and per emission batch: a multiple of K, 64 frames (the emission kernels are bound by per-wave latency
    // and by the launch gaps between them, and this runtime runs the captured graph's chain and emission kernels one after the
    // other: fewer, larger batches -- 346x260, round 3: 64 frames 8.7 Gev/s, 32 frames 7.7; 1280x720 noisy, round 5 with the pull
    // writer: 32 frames 9.57 / 9.61, 64 frames 10.70 / 10.71 Gev/s -- until round 5 large grids took max(K, 8) for their ring's
    // memory: 64 frames of 1280x720 are 0.7 GB of count words in a ring of three batches).  Grids whose ring would take more than an
    // eighth of the free device memory, and multi-clip runs whose short batch holds 64 (frame, clip) pairs anyway, keep the short batches.
    int m = std::max(1, 64 / K);
    if (!chain_small_grid(h)) {
        if (h->ch_long_batches < 0) { // decided once per handle (the answer must not change with what the handle itself allocates)
            size_t free_b = 0, total_b = 0;
            const size_t ring = (size_t)3 * 64 * h->n_clips * h->npx_pad * (sizeof(uint32_t) + sizeof(float) + sizeof(uint4));
            h->ch_long_batches = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ring <= free_b / 8) ? 1 : 0;
        }
        // (64 clips of 346x260 have 2 048 (frame, clip) pairs in a 32-frame batch already: 11.64 Gev/s with 32 frames, 11.37 with 64)
        const int m_short = std::max(1, std::max(K, 8) / K);
        if (!h->ch_long_batches || (long long)m_short * K * h->n_clips >= 64) m = m_short;
    }
    if (const char *ev = getenv("V2E_AMD_CHAIN_M")) { const int v = atoi(ev); if (v >= 1 && v <= 64) m = v; }
    m = std::max(1, std::min(m, 64 / K)); // k_cemit sums a batch's per-frame event counts one frame per lane
    const int E = m * K;
    if (h->ch_K != K || h->ch_E != E || h->ch_nkeys_cap != h->nkeys_cap || h->ch_fused != (int)fused || h->ch_nD != nD) {
        // a new configuration: everything sized by it goes, in BOTH scratch sets (overlapped runs alternate between two, see
        // v2e_emu::swap_scratch); what the run needs is allocated below, for the set that is current
        h->sync_runs();
        h->free_chain_scratch();
        h->ch_launch_cap = 0;
        h->ch_K = K;
        h->ch_E = E;
        h->ch_fused = fused;
        // (ring of frame slots: chain_ring_batches)
        h->ch_nD = nD;
        h->ch_D = nD * E;
        h->ch_nwp = (chain_egroups(h) + 15) / 16 * 16; // emission groups (one wave each), padded to the 16 a lane of k_cframe takes
        h->ch_nkeys_cap = h->nkeys_cap;
        // The event writer (decided once per configuration, for both scratch sets): the pull needs 32 bytes per (group, key) of the
        // three table sets -- only the keys a group has events of are ever written or read (no clearing)
        const size_t mask_bytes = 3 * sizeof(uint32_t) * 2 * GPX * E * (size_t)h->n_clips * h->nkeys_cap * h->ch_nwp;
        // (beyond 2 M groups = 33 M pixels the two-level search's coarse rows would not fit the 64 KB of LDS a launch gets by default)
        h->ch_pull = (h->ch_nwp / 16 <= 8192 && emit_pull(mask_bytes)) ? 1 : 0;
        h->drop_graphs();
    }
    const size_t nc = (size_t)h->n_clips;
    if (!h->ch_cnt) {
        V2E_HIP(hipMalloc(&h->ch_cnt, sizeof(uint32_t) * h->ch_D * nc * h->npx_pad));
        V2E_HIP(hipMalloc(&h->ch_ruleM, sizeof(uint32_t) * h->ch_D * nc));
        V2E_HIP(hipMemset(h->ch_ruleM, 0, sizeof(uint32_t) * h->ch_D * nc));
        // three sets of emission tables in rotation: the tables of batches b + 1, b + 2 are built while k_cemit(b) reads its own
        V2E_HIP(hipMalloc(&h->ch_wmax, 3 * sizeof(uint16_t) * E * nc * h->ch_nwp));
        V2E_HIP(hipMemset(h->ch_wmax, 0, 3 * sizeof(uint16_t) * E * nc * h->ch_nwp));
        V2E_HIP(hipMalloc(&h->ch_wtot, sizeof(uint16_t) * 3 * E * nc * h->nkeys_cap * h->ch_nwp));
        V2E_HIP(hipMemset(h->ch_wtot, 0, sizeof(uint16_t) * 3 * E * nc * h->nkeys_cap * h->ch_nwp));
        V2E_HIP(hipMalloc(&h->ch_cf, 3 * sizeof(CFrame) * E * nc));
        V2E_HIP(hipMemset(h->ch_cf, 0, 3 * sizeof(CFrame) * E * nc));
        V2E_HIP(hipMalloc(&h->ch_cdone, 3 * sizeof(unsigned) * E * nc));
        V2E_HIP(hipMemset(h->ch_cdone, 0, 3 * sizeof(unsigned) * E * nc));
        V2E_HIP(hipMalloc(&h->ch_cT, 3 * sizeof(uint32_t) * E * nc * h->nkeys_cap));
        V2E_HIP(hipMalloc(&h->ch_ckbase, 3 * sizeof(uint32_t) * E * nc * h->nkeys_cap));
        V2E_HIP(hipMalloc(&h->ch_cperm, 3 * sizeof(uint32_t) * E * nc * h->max_iters * 8));
        V2E_HIP(hipMalloc(&h->ch_cpre, 3 * sizeof(uint32_t) * E * nc * h->nkeys_cap * h->ch_nwp));
        if (h->ch_pull) {
            V2E_HIP(hipMalloc(&h->ch_cmask, 3 * sizeof(uint32_t) * 2 * GPX * E * nc * h->nkeys_cap * h->ch_nwp));
            V2E_HIP(hipMalloc(&h->ch_cpre16, 3 * sizeof(uint32_t) * E * nc * h->nkeys_cap * (h->ch_nwp / 16)));
        }
        if (!fused) { // (zero once: k_ahead never writes a slot's padding [npx, npx_pad), and k_chain's lanes beyond the frame read it)
            V2E_HIP(hipMalloc(&h->ch_rec, sizeof(uint4) * (size_t)h->ch_D * nc * h->npx_pad));
            V2E_HIP(hipMemset(h->ch_rec, 0, sizeof(uint4) * (size_t)h->ch_D * nc * h->npx_pad));
        }
        // The fills above are ordered on the DEFAULT stream only, and a pipelined run's head (its k_ahead records, its tables) goes out on
        // streams of the handle that do not wait for it: enqueued behind a run still executing on the default stream, the record ring's
        // fill once ran AFTER the next run's k_ahead had written its records (round 6: wrong events in the first run on a freshly
        // allocated scratch set, only with another run in flight).  Allocation is rare: wait for the device.
        V2E_HIP(hipDeviceSynchronize());
        h->drop_graphs();
    }
    h->ch_max_blocks = max_blocks;
    h->ch_inst = inst;
    const int n_launch = (n_frames + K - 1) / K + 1;
    if (n_launch + 2 > h->run_off_cap) {
        h->sync_runs();
        hipFree(h->run_off); hipFree(h->alt_of((void **)&h->run_off));
        h->run_off = nullptr; h->alt_of((void **)&h->run_off) = nullptr;
        h->run_off_cap = n_launch + 2;
        h->drop_graphs();
    }
    if (!h->run_off) V2E_HIP(hipMalloc(&h->run_off, sizeof(unsigned long long) * (size_t)h->run_off_cap * h->n_clips));
    if (!h->ahead) V2E_HIP(hipStreamCreateWithFlags(&h->ahead, hipStreamNonBlocking));
    if (!h->tabs) V2E_HIP(hipStreamCreateWithFlags(&h->tabs, hipStreamNonBlocking));
    if (!h->side2) V2E_HIP(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
    auto grow = [&](std::vector<hipEvent_t> &v, size_t n) -> int {
        while (v.size() < n) {
            hipEvent_t e;
            V2E_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v.push_back(e);
        }
        return 0;
    };
    if (grow(h->ev_ahead, n_launch + 1) || grow(h->ev_chain, n_launch + 1) || grow(h->ev_fork, n_launch + 1) ||
        grow(h->ev_join, n_launch + 1) || grow(h->ev_tab, n_launch + 1)) return V2E_EHIP;
    if (has_refr) {
        if (!h->ch_tsold) V2E_HIP(hipMalloc(&h->ch_tsold, sizeof(float) * (size_t)h->ch_D * h->n_clips * h->npx_pad));
        if (!h->ch_ck && K > CHAIN_SUB) V2E_HIP(hipMalloc(&h->ch_ck, (size_t)2 * (K / CHAIN_SUB - 1 + (K % CHAIN_SUB ? 1 : 0)) * 20 * h->n_clips * h->npx_pad)); // see ChainArgs::ckc_base
        if (!h->ch_base2) {
            const size_t n = (size_t)h->n_clips * h->npx_pad;
            V2E_HIP(hipMalloc(&h->ch_base2, sizeof(double) * n));
            V2E_HIP(hipMalloc(&h->ch_lp2, sizeof(double) * n));
            V2E_HIP(hipMalloc(&h->ch_ts2, sizeof(float) * n));
            V2E_HIP(hipMemset(h->ch_base2, 0, sizeof(double) * n));
            V2E_HIP(hipMemset(h->ch_lp2, 0, sizeof(double) * n));
            V2E_HIP(hipMemset(h->ch_ts2, 0, sizeof(float) * n));
            V2E_HIP(hipDeviceSynchronize()); // (as above)
        }
        if (n_launch > h->ch_launch_cap) {
            h->sync_runs();
            hipFree(h->ch_gM); hipFree(h->ch_bar); hipFree(h->alt_of((void **)&h->ch_gM)); hipFree(h->alt_of((void **)&h->ch_bar));
            h->ch_gM = nullptr; h->ch_bar = nullptr; h->alt_of((void **)&h->ch_gM) = nullptr; h->alt_of((void **)&h->ch_bar) = nullptr;
            h->ch_launch_cap = n_launch;
            h->drop_graphs();
        }
        if (!h->ch_gM) {
            V2E_HIP(hipMalloc(&h->ch_gM, sizeof(uint32_t) * (size_t)h->ch_launch_cap * (K + 1) * h->n_clips * K));
            V2E_HIP(hipMalloc(&h->ch_bar, sizeof(unsigned) * (size_t)h->ch_launch_cap * 2 * K * h->n_clips));
            h->drop_graphs();
        }
    }
    return 0;
}

// can this run go through k_chain?  (otherwise: the count / rank / scan / emit kernels, one frame at a time)
static bool chain_eligible(const v2e_emu *h, const v2e_emu_params *p, int dtype)
{
    if (h->max_iters > CHAIN_MAX_ITERS) return false;
    if (dtype == V2E_DT_F64 && p->log_input) return false; // the frame record carries the lin-log value as float32
    if (p->photoreceptor_noise) return false;              // one more state plane and normal per pixel
    if (h->sc_hp) return false;                            // SCIDVS: two more state planes
    if (h->cs_sur) return false;                           // CSDVS: the surround is stepped between frames
    return true;
}

static int enqueue_run_chain(v2e_emu *h, const v2e_emu_params *p, const KArgs &a_in, const void *frames, int dtype, int n_frames,
                             float *events, uint64_t cap, v2e_frame_rec *recs, hipStream_t s, hipGraph_t graph = nullptr,
                             std::vector<hipEvent_t> *ev_main = nullptr, std::vector<hipEvent_t> *ev_side = nullptr,
                             bool capturing = false, bool pipelined = false)
{
    // Three logical streams: the chain (k_chain, K frames per launch); k_ahead and the emission tables (k_ctot, k_cframe);
    // the event writer k_cemit; the last two in batches of E = m K frames (batch b = frames [b E, (b + 1) E) = chain
    // launches [b m, (b + 1) m)).  The emission of a batch is a pipeline of its own: tables of batch b + 1 beside the rows of b.
    //   k_ahead(b)  before chain launch b m; overwrites the records of batch b - nD, last read by launch (b - nD + 1) m (its redo)
    //   tables(b)   once batch b is final: after the launch that validated its last K frames (or, without a refractory
    //               period, after its last launch); k_cemit(b) after tables(b) and k_cemit(b - 1) (running event offset)
    //   chain launch b m overwrites the ring slots of batch b - nD: after k_cemit(b - nD)   (nD = ch_D / E batches in the ring)
    const size_t esz = dtype == V2E_DT_U8 ? 1 : (dtype == V2E_DT_F32 ? 4 : 8);
    const bool has_refr = p->refractory_period_s > 0;
    const int K = h->ch_K, E = h->ch_E, m = E / K, D = h->ch_D, NC = h->n_clips, nD = h->ch_nD;
    const bool fused_rec = h->ch_fused != 0;
    const std::vector<ChainLaunch> plan = chain_plan(n_frames, K, E, nD, has_refr, fused_rec);
    const int nL = (int)plan.size();
    const int nB = (n_frames + K - 1) / K;
    const int nEB = (n_frames + E - 1) / E;
    KArgs a = a_in; // kernel arguments are passed by address (hipLaunchKernel / kernel nodes copy them at the call)
    Sched sc;
    sc.h = h; sc.st[ST_MAIN] = s; sc.st[ST_AHEAD] = h->ahead; sc.st[ST_SIDE] = h->side; sc.st[ST_TAB] = h->tabs; sc.st[ST_SIDE2] = h->side2; sc.graph = graph;
    if (ev_main) // instrumented run: every kernel on the one stream, so that a launch's HIP events bracket that kernel running alone --
        for (int q = 0; q < ST_COUNT; ++q) sc.st[q] = s; // which is how the captured graph of the timed runs executes on this runtime
    if (graph) {
        sc.ev_cap = nL + 2;
        sc.evdeps.assign((size_t)EV_KINDS * sc.ev_cap, std::vector<hipGraphNode_t>());
    }
    auto mark = [&](std::vector<hipEvent_t> *v, hipStream_t stq) -> int { // instrumented runs (never graphs)
        if (!v || graph) return 0;
        hipEvent_t e;
        V2E_HIP(hipEventCreate(&e));
        v->push_back(e);
        V2E_HIP(hipEventRecord(e, stq));
        return 0;
    };
    // clips resident at once: with a refractory period every workgroup of a clip must be resident for the redo rendezvous
    // (K = 1 has no rendezvous: all clips at once)
    int gy = NC;
    if (has_refr && K > 1) gy = std::max(1, std::min(NC, h->ch_max_blocks / std::max(h->ngroups, 1)));
    dim3 grid(h->ngroups, gy);
    constexpr bool no_emit = false;
    const bool no_emit_run = false;
    constexpr int chain_prio = 3; // wave priority of the chain (measured against 0 in round 3)
    // occupancy cap of the kernels that run BESIDE the chain (k_ahead, k_ctot, k_cemit): dynamic LDS they do not use, so that a
    // CU holds at most floor(160 KB / pad) of their workgroups (0: no cap)
    constexpr int side_pad = 0; // (an LDS pad capping the side kernels' occupancy was measured in round 3 and bought nothing)
    constexpr int bar_light = 1; // the fence-free rendezvous (round 4; the fenced one of round 3 is still in clip_barrier)
    // lock-step frames in redo passes (emu_chain.h): measured round 4 with the fence-free rendezvous, A/B in one process each:
    // a launch whose rule-on frames come in a run 109-112 -> 89-92 us, a launch with a single rule-on frame 54-60 -> 68-71 us
    // (the extra rendezvous is ~10 us inside the frame loop: it also drains the record prefetch), 10.4-10.6 against 10.8-11.0
    // Gev/s on the benchmark clip: off by default, kept behind V2E_AMD_LOCKSTEP=1 (GPU parity suite green with it on)
    constexpr int lockstep = 0;
    // Under stream capture only edges between the origin stream and a forked stream are safe (edges between two forked
    // streams crash hipStreamEndCapture on this runtime): tables and rows then share the side stream.
    constexpr int tab_env = ST_TAB; // plain streams: tables on a stream of their own
    constexpr bool tabs_on_main = false;
    // pipelined plain launches (v2e_emu_run, mode 0 | 1024): four streams in all
    // -- the caller's (the chain), k_ahead's, the emission tables' (h->side) and the event rows' (h->side2): with tables and rows on one
    // stream that stream was busy ~500 of a step's 720 us and the chain waited for it where the ring of frame slots wraps (device
    // stamps: 57 us before the first launch of batch 3; with the rows on their own stream 19, 717-732 -> 699-713 us per step).
    // (The tables on the k_ahead stream instead: the next run's head queues behind this run's last tables, 290 us between two runs'
    // chains.)
    constexpr bool pipe_rows_own = true;
    const int tab_stream = capturing ? (tabs_on_main ? ST_MAIN : ST_SIDE) : (pipelined ? ST_SIDE : tab_env);
    constexpr int NSET = 3; // emission table sets (chain_alloc sizes them)
    constexpr bool one_row_stream = false;
    // b: the emission's slot (event indices, table set b % NSET, event offsets run_off[b] -> run_off[b + 1]); frames [ef0, ef0 + enE).
    // A batch is normally slot b = frames [b E, (b + 1) E); the run's LAST batch may be emitted in two pieces (slots nEB - 1, nEB).
    // (Round 5 measured the tables of the run's LAST pieces on the chain's stream, beside the rows of the batch before on the side
    // stream: 12.0 -> 11.5 Gev/s -- they delay the tail launch, which the last rows wait for.  Not kept.)
    auto launch_emission = [&](int b, int ef0, int enE) -> int { // EV_FORK[b] has been recorded on the chain stream
        CEmitArgs ea;
        memset(&ea, 0, sizeof(ea));
        ea.ctl = h->run_ctl; ea.recs = recs; ea.fidx_base = h->run_fidx;
        ea.f0 = ef0; ea.nE = enE; ea.D = D; ea.n_clips = NC; ea.slot0 = ef0 % D;
        ea.nwp = h->ch_nwp; ea.nwaves = chain_egroups(h); ea.E = E; // emission groups (GROUP_PX pixels, one wave each)
        ea.cnt = h->ch_cnt; ea.tsold = has_refr ? h->ch_tsold : nullptr; ea.ruleM = has_refr ? h->ch_ruleM : nullptr;
        const size_t set = (size_t)(b % NSET) * E * NC; // table set of this batch
        ea.wmax = h->ch_wmax + set * h->ch_nwp; ea.wtot = h->ch_wtot + set * h->nkeys_cap * h->ch_nwp;
        ea.cf = h->ch_cf + set; ea.cT = h->ch_cT + set * h->nkeys_cap; ea.ckbase = h->ch_ckbase + set * h->nkeys_cap;
        ea.cperm = h->ch_cperm + set * h->max_iters * 8; ea.cpre = h->ch_cpre + set * h->nkeys_cap * h->ch_nwp;
        ea.cdone = h->ch_cdone + set;
        const bool pull = h->ch_cmask != nullptr;
        ea.cmask = pull ? h->ch_cmask + set * h->ch_nwp * h->nkeys_cap * 2 * GPX : nullptr;
        // k_cpull: small frames search a whole prefix row in LDS (346x260: 1.4 KB per key), a workgroup per ~2048 pixels' worth of
        // a frame's rows; large ones search every 16th entry there and one line from L2, and take more, smaller workgroups (their
        // batches are few frames: the launch needs the parallelism)
        bool two = h->ch_nwp > 1024;
        if (const char *ev = getenv("V2E_AMD_PULL_TWO_LEVEL")) two = atoi(ev) != 0;
        ea.cpre16 = pull && two ? h->ch_cpre16 + set * h->nkeys_cap * (h->ch_nwp / 16) : nullptr;
        ea.wpf = std::max(4, std::min(two ? 128 : 64, h->npx / 2048)); // (1280x720 noisy, round 5: 450 workgroups per frame 9.28, 128: 9.68, push 9.41 Gev/s)
        ea.p2 = two ? 16 : 64;
        while (ea.p2 < (two ? h->ch_nwp / 16 : ea.nwaves)) ea.p2 *= 2;
        ea.events = (float4 *)events; ea.cap = cap;
        ea.off_in = h->run_off + (size_t)b * NC;
        ea.off_out = h->run_off + (size_t)(b + 1) * NC;
        // event records of k_cemit: 256 x ich per group (wave) and pass.  The chain's workgroups need their LDS (5 KB per frame)
        // on every CU: 4 iterations per pass (measured at 346x260: 3 / 6 / 10 per pass 10.47 / 10.30 / 10.31 Gev/s) keep an emission workgroup of four groups at 16 KB
        constexpr int ich_env = 0;
        ea.ich = (ich_env >= 1 && ich_env <= 31) ? ich_env : 4;
        ea.capw = GROUP_PX * ea.ich;
        const bool rows_own = pipelined && pipe_rows_own;
        // (one row stream: the rows carry the running event offset forward themselves -- no k_coff launch)
        ea.coff_in_cemit = (tab_stream == ST_SIDE || tab_stream == ST_MAIN || one_row_stream || rows_own) ? 1 : 0;
        // frames per workgroup (measured at 346x260, 32-frame batches: k_ctot 4 frames 13 us, 32 frames 37 us; k_cemit 1 frame
        // 34 us, 8 frames 45 us -- these kernels are bound by the latency of a wave's dependent loads, not by wave dispatch:
        // more, shorter waves win)
        constexpr int zpw_env = 0;
        (void)zpw_env;
        ea.zpw_tot = CTOT_ZF;
        ea.zpw_emit = 1; // (k_cemit: one frame per workgroup; several per workgroup measured slower and cost 25 % more instructions)
        const int REC_LDS = (ea.capw + WAVE) * 4 * (BLOCK / WAVE);
        const int egx = (chain_egroups(h) + BLOCK / WAVE - 1) / (BLOCK / WAVE); // workgroups of four emission groups
        void *args[] = {(void *)&a, (void *)&ea};
        // tables on a stream of their own (NSET table sets rotate: tables(b + 1) are built while k_cemit(b) reads those of b;
        // k_cemit(b - NSET), which read this set last, is waited for), rows on the side stream
        if (tab_stream != ST_MAIN && sc.wait(tab_stream, EV_FORK, b)) return V2E_EHIP;
        if (b >= NSET && (tab_stream != ST_SIDE || rows_own) && sc.wait(tab_stream, EV_JOIN, b - NSET)) return V2E_EHIP;
        if (!no_emit) {
            if (sc.kernel(tab_stream, (const void *)k_ctot, dim3(egx, NC, (ea.nE + CTOT_ZF - 1) / CTOT_ZF), dim3(BLOCK), (size_t)side_pad, args)) return V2E_EHIP;
            // a key row of a small grid is a couple of steps of one wave: one workgroup per frame; of a large grid (beyond a million
            // pixels) a segmented scan by a workgroup of its own.  (1280x720 has 3 600 groups since the groups are 256 pixels: k_cframe1,
            // 45 us per 32 frames; round 5 measured k_cframe there: 9.74 / 9.51 -> 8.89 / 9.02 Gev/s.)
            if (h->ch_nwp <= 4096) {
                if (sc.kernel(tab_stream, (const void *)k_cframe1, dim3(1, NC, ea.nE), dim3(CFRAME_THREADS), 0, args)) return V2E_EHIP;
            } else if (sc.kernel(tab_stream, (const void *)k_cframe, dim3(std::min(h->nkeys_cap, CFRAME_ROWS), NC, ea.nE), dim3(CFRAME_THREADS), 0, args)) {
                return V2E_EHIP;
            }
        }
        // rows: two batches side by side on two streams (the rows of a batch depend on nothing but its tables; the batch's
        // event offset then comes from k_coff behind the tables); on one stream with the tables the rows carry it forward
        const int row_stream = rows_own ? ST_SIDE2 : ((tab_stream == ST_SIDE || tab_stream == ST_MAIN || one_row_stream || pipelined) ? ST_SIDE : ((b & 1) ? ST_SIDE2 : ST_SIDE));
        if (!no_emit && !ea.coff_in_cemit) {
            void *oargs[] = {(void *)&ea};
            if (sc.kernel(tab_stream, (const void *)k_coff, dim3(NC), dim3(WAVE), 0, oargs)) return V2E_EHIP;
        }
        if (sc.record(EV_TAB, b, tab_stream)) return V2E_EHIP;
        if (row_stream != tab_stream && sc.wait(row_stream, EV_TAB, b)) return V2E_EHIP;
        if (mark(ev_side, sc.st[row_stream])) return V2E_EHIP;
        if (pull) {
            if (sc.kernel(row_stream, ea.cpre16 ? (const void *)k_cpull<true> : (const void *)k_cpull<false>, dim3(8 * ea.wpf * ((ea.nE + 7) / 8), NC), dim3(BLOCK), (size_t)(2 * ea.p2 * 4), args)) return V2E_EHIP;
        } else if (!no_emit && sc.kernel(row_stream, (const void *)k_cemit, dim3(egx, NC, ea.nE), dim3(BLOCK), (size_t)std::max(REC_LDS, side_pad), args)) return V2E_EHIP;
        if (mark(ev_side, sc.st[row_stream])) return V2E_EHIP;
        return sc.record(EV_JOIN, b, row_stream);
    };
    // Batch 0 of a run is split at the first chain launch's frames: the chain starts after the records of K frames, not of E
    // (its first launch idled through the whole first k_ahead: ~15 us of every 300-frame step); the rest of the batch follows on
    // the same stream and is waited for by the batch's other launches (event slot nL of EV_AHEAD: the batches use 0 .. nEB - 1).
    constexpr bool split_first_ahead = true;
    // (pipelined runs: the head runs beside the run before, the chain never waits for it -- one k_ahead launch, one event and one wait
    // fewer: profiles/r06_emulator_experiments.txt item 14)
    const bool split0 = split_first_ahead && m > 1 && n_frames > K && !pipelined;
    // Pipelined runs whose ring holds the whole run (n_frames <= D: 300 frames in five batches of 64): ONE k_ahead launch for the run, right
    // behind its upload and zero fills, and ONE wait of the chain for all of it -- in steady state it ran beside the run before.  Every
    // wait and every record on the chain's stream costs its next launch ~7 us even when the event has long completed (gap before a
    // batch's first launch 13-16 us, before its second 6-7: profiles/r06_emulator_experiments.txt item 23).
    const bool ahead_whole = pipelined && !fused_rec && n_frames <= D;
    auto launch_ahead = [&](int b) -> int {
        if (!ahead_whole && b >= nD && sc.wait(ST_AHEAD, EV_CHAIN, (b - nD + 1) * m)) return V2E_EHIP; // records of batch b - nD: last read by that launch's redo
        // frame pairs touched by a launch: at most nf / 2 + 1 (the device knows the run's first frame index, the host
        // does not when it builds a graph: one extra pair covers either alignment; threads of a pair outside the range return)
        const int b0 = ahead_whole ? 0 : b * E, b1 = ahead_whole ? n_frames : std::min((b + 1) * E, n_frames);
        const int cut = (b == 0 && split0) ? std::min(K, b1) : b1;
        for (int part = 0; part < 2; ++part) {
            const int f0 = part == 0 ? b0 : cut, f1 = part == 0 ? cut : b1;
            if (f1 <= f0) continue;
            AheadArgs aa;
            memset(&aa, 0, sizeof(aa));
            aa.frames_pp = h->run_frames; aa.frame_stride = (unsigned long long)NC * h->npx * esz;
            aa.ctl = h->run_ctl; aa.fidx_base = h->run_fidx;
            aa.f0 = f0; aa.nf = f1 - f0; aa.D = D; aa.n_clips = NC; aa.slot0 = f0 % D;
            aa.rec = h->ch_rec;
            // frame pairs per thread: a thread's set-up (thresholds, two divisions, the seed's ten Philox round keys) is paid once per
            // thread, so a thread takes several pairs -- as many as leave ~4 workgroups per CU (346x260, 64 frames: 3 z-blocks of 11 pairs;
            // round 6, A/B x 3 in one session: 1 pair 683-689 us per step, 4: 668-676, 11: 665-670, 16: 663-671; k_ahead's instructions
            // per 64-pixel wave-frame 133 + 81 -> 108 + 43), at most 16
            {
                const int pairs = aa.nf / 2 + 1;
                const long long wg_xy = (long long)h->ngroups * NC;
                int zb = (int)std::min<long long>(pairs, std::max<long long>((4ll * h->n_cu + wg_xy - 1) / wg_xy, (pairs + 15) / 16));
                aa.ppt = (pairs + zb - 1) / zb;
            }
            void *args[] = {(void *)&a, (void *)&aa};
            if (sc.kernel(ST_AHEAD, (const void *)k_ahead<uint8_t>, dim3(h->ngroups, NC, (aa.nf / 2 + 1 + aa.ppt - 1) / aa.ppt), dim3(BLOCK), (size_t)side_pad, args)) return V2E_EHIP;
            if (sc.record(EV_AHEAD, part == 0 ? b : nL, ST_AHEAD)) return V2E_EHIP;
        }
        return 0;
    };
    // state planes: X[0] the caller's (bound) planes, X[1] the engine's second set; launch L reads X[L % 2], writes X[(L + 1) % 2]
    void *xb[2] = {h->base, has_refr ? h->ch_base2 : h->base}, *xl[2] = {h->lp, has_refr ? h->ch_lp2 : h->lp};
    float *xt[2] = {h->ts_mem, has_refr ? h->ch_ts2 : h->ts_mem};
    // the run's uploads (frame times, first frame index) and the zero fills precede everything
    constexpr bool no_side_fork = true; // under capture the side stream joins at its first batch
    // The zero fill precedes the fork.  (Round 5 tried the fork first -- k_ahead touches nothing the fill clears -- and lost 10 %:
    // 12.0 -> 10.8 Gev/s, A/B x 3 in one session; the enqueue order of a capture decides which branches this runtime overlaps,
    // profiles/r03_graph_scheduling.txt.)
    {   // one launch: the run's records, batch 0's event offset, and (refractory runs) the rule-on maxima and rendezvous counters
        void *const zp[4] = {recs, h->run_off, has_refr ? (void *)h->ch_gM : nullptr, has_refr ? (void *)h->ch_bar : nullptr};
        const size_t zb[4] = {sizeof(v2e_frame_rec) * (size_t)n_frames * NC, sizeof(unsigned long long) * NC,
                              sizeof(uint32_t) * (size_t)nL * (K + 1) * NC * K, sizeof(unsigned) * (size_t)nL * 2 * K * NC};
        if (sc.zero4(pipelined ? ST_AHEAD : ST_MAIN, zp, zb)) return V2E_EHIP;
    }
    // (pipelined: no forks -- the k_ahead stream carries the run's upload and zero fills itself and runs ahead of the chain, the
    // emission stream waits for the chain batch by batch)
    if (!pipelined) {
    if (sc.record(EV_FORK, nL, ST_MAIN)) return V2E_EHIP;
    if (!(capturing && no_side_fork) && sc.wait(ST_SIDE, EV_FORK, nL)) return V2E_EHIP;
    if (!fused_rec && sc.wait(ST_AHEAD, EV_FORK, nL)) return V2E_EHIP;
    if (!capturing && (sc.wait(ST_TAB, EV_FORK, nL) || sc.wait(ST_SIDE2, EV_FORK, nL))) return V2E_EHIP;
    }
    int last_ahead = -1; // the last k_ahead batch this piece enqueued (joined at the end)
    if (ahead_whole && launch_ahead(0)) return V2E_EHIP; // (all of the run's records: before the one event the chain waits for)
    if (pipelined && (sc.record(EV_FORK, nL, ST_AHEAD) || sc.wait(ST_MAIN, EV_FORK, nL))) return V2E_EHIP; // the upload and the zero fills, for the chain
    for (int b = 0; b < std::min(nEB, 2) && !fused_rec && !ahead_whole; ++b) {
        if (launch_ahead(b)) return V2E_EHIP;
        last_ahead = b;
    }
    // last batch in two pieces: only where the last launch is not the first of its batch (else there is nothing to emit early)
    constexpr bool split_tail_env = true;
    const int tail_f0 = (nB - 1) * K;                          // first frame of the last chain launch with frames
    // (pipelined runs: the last emission finishes beside the next run's chain -- one piece, six API calls fewer)
    const bool split_tail = split_tail_env && has_refr && m > 1 && nB >= 2 && (nB - 1) % m != 0 && !no_emit_run && !pipelined;
    int last_emitted = -1; // the last emission slot this piece enqueued (joined at the end)
    constexpr int allon_env = 1;
    const bool allon = allon_env && a_in.has_cutoff && a_in.do_leak && a_in.do_shot && a_in.has_refr;
    const void *kfn = chain_fn(p->f64_state != 0, dtype, fused_rec, allon);
    for (int L = 0; L < nL; ++L) {
        const ChainLaunch &pl = plan[L];
        const bool tail = pl.nf == 0;
        ChainArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.frames_pp = h->run_frames; ca.frame_stride = (unsigned long long)NC * h->npx * esz; ca.fidx_base = h->run_fidx;
        ca.ctl = h->run_ctl;
        ca.f0 = pl.f0; ca.nf = pl.nf; ca.pf0 = pl.pf0; ca.pnf = pl.pnf;
        ca.D = D; ca.n_clips = NC; ca.K = K; ca.ngroups = h->ngroups;
        ca.slot_f0 = pl.f0 % D; ca.slot_pf0 = pl.pf0 % D;
        ca.cnt = h->ch_cnt; ca.ruleM = h->ch_ruleM; ca.tsold = has_refr ? h->ch_tsold : nullptr;
        ca.rec = h->ch_rec;
        if (has_refr) {
            ca.gM_cur = h->ch_gM + (size_t)L * (K + 1) * NC * K;
            ca.gM_prev = h->ch_gM + (size_t)(L > 0 ? L - 1 : 0) * (K + 1) * NC * K;
            ca.bar_prev = h->ch_bar + (size_t)(L > 0 ? L - 1 : 0) * 2 * K * NC;
        }
        const int in = L % 2, out = tail ? 0 : (L + 1) % 2, pin = (L + 1) % 2;
        ca.base_in = xb[in]; ca.lp_in = xl[in]; ca.ts_in = xt[in];
        ca.base_fix = xb[in]; ca.lp_fix = xl[in]; ca.ts_fix = xt[in];
        ca.base_out = xb[out]; ca.lp_out = xl[out]; ca.ts_out = xt[out];
        ca.base_pin = xb[pin]; ca.lp_pin = xl[pin]; ca.ts_pin = xt[pin];
        constexpr bool no_ckpt = false;
        if (has_refr && K > CHAIN_SUB && h->ch_ck && !no_ckpt) {
            const size_t plane = (size_t)(K / CHAIN_SUB - 1 + (K % CHAIN_SUB ? 1 : 0)) * NC * h->npx_pad; // elements per (parity, quantity): a checkpoint before frames 8, 16, ...
            auto ck = [&](int par, char *&bp, char *&lpp, float *&tp) {
                char *q = (char *)h->ch_ck + (size_t)par * plane * 20;
                bp = q; lpp = q + plane * 8; tp = (float *)(q + plane * 16);
            };
            char *b0, *l0, *b1, *l1; float *t0, *t1;
            ck(L % 2, b0, l0, t0);
            ck((L + 1) % 2, b1, l1, t1);
            ca.ckc_base = b0; ca.ckc_lp = l0; ca.ckc_ts = t0;
            ca.ckp_base = b1; ca.ckp_lp = l1; ca.ckp_ts = t1;
        }
        ca.recs = recs;
        ca.store_out = tail && in != 0;
        ca.prio = chain_prio;
        ca.bar_light = bar_light;
        ca.lockstep = lockstep;
        ca.dbg = (h->dbg && L == nB / 2) ? h->dbg : nullptr;
        ca.stamp_pp = h->stamp_slot; ca.lidx = std::min(L, STAMP_LAUNCHES - 1);
        if (pl.wait_join >= 0 && sc.wait(ST_MAIN, EV_JOIN, pl.wait_join)) return V2E_EHIP; // ring slots: read by k_cemit of that batch
        // (records of batches 0 and 1 without the head piece in this enqueue: the caller ordered this piece behind the head's stream)
        if (!ahead_whole && pl.wait_ahead >= 0 && sc.wait(ST_MAIN, EV_AHEAD, pl.wait_ahead)) return V2E_EHIP;
        if (split0 && !fused_rec && !tail && L >= 1 && L < m && sc.wait(ST_MAIN, EV_AHEAD, nL)) return V2E_EHIP; // the rest of batch 0
        if (mark(ev_main, s)) return V2E_EHIP; // instrumented runs: an event before and after every chain launch
        void *args[] = {(void *)&a, (void *)&ca};
        if (sc.kernel(ST_MAIN, kfn, grid, dim3(BLOCK), chain_dyn_lds(fused_rec), args)) return V2E_EHIP;
        if (mark(ev_main, s)) return V2E_EHIP;
        if (!ahead_whole && !fused_rec && L % m == 0 && sc.record(EV_CHAIN, L, ST_MAIN)) return V2E_EHIP; // (what a later k_ahead batch waits for where the ring wraps)
        // Enqueue order of the two side branches: k_ahead first.  It decides how this runtime executes the captured graph:
        // with the emission enqueued first the chain's next launch runs BEHIND the emission kernels (measured, profiles/
        // r03_graph_scheduling.txt), with k_ahead first it runs beside them.
        constexpr bool ahead_first = true;
        if (ahead_first && !ahead_whole && pl.ahead_next >= 0) { if (launch_ahead(pl.ahead_next)) return V2E_EHIP; last_ahead = pl.ahead_next; }
        if (split_tail && L == nB - 1 && tail_f0 > (nEB - 1) * E) {
            if (sc.record(EV_FORK, nEB - 1, ST_MAIN)) return V2E_EHIP;
            if (launch_emission(nEB - 1, (nEB - 1) * E, tail_f0 - (nEB - 1) * E)) return V2E_EHIP;
            last_emitted = nEB - 1;
        }
        if (pl.emit_batch >= 0 && split_tail && pl.emit_batch == nEB - 1 && tail_f0 > (nEB - 1) * E) {
            if (sc.record(EV_FORK, nEB, ST_MAIN)) return V2E_EHIP;
            if (launch_emission(nEB, tail_f0, n_frames - tail_f0)) return V2E_EHIP;
            last_emitted = nEB;
        } else if (pl.emit_batch >= 0) {
            // (emitting the batches in PAIRS -- one record on the chain's stream per 128 frames -- was measured in round 6: gaps inside a run
            //  54 -> 46 us, but the bunched emission costs the chain's launches more than that: 614-633 -> 628-641 us per step)
            if (sc.record(EV_FORK, pl.emit_batch, ST_MAIN)) return V2E_EHIP;
            if (launch_emission(pl.emit_batch, pl.emit_batch * E, std::min((pl.emit_batch + 1) * E, n_frames) - pl.emit_batch * E)) return V2E_EHIP;
            last_emitted = pl.emit_batch;
        }
        if (!ahead_first && !ahead_whole && pl.ahead_next >= 0) { if (launch_ahead(pl.ahead_next)) return V2E_EHIP; last_ahead = pl.ahead_next; }
    }
    if (!graph && !pipelined) { // join: the piece is complete on `s` (a graph is complete when all its nodes are)
        if (last_emitted >= 0 && sc.wait(ST_MAIN, EV_JOIN, last_emitted)) return V2E_EHIP;
        if (!capturing && last_emitted >= 0 && sc.wait(ST_MAIN, EV_TAB, last_emitted)) return V2E_EHIP;
        if (!capturing && last_emitted >= 1 && sc.wait(ST_MAIN, EV_JOIN, last_emitted - 1)) return V2E_EHIP;
        if (!fused_rec && last_ahead >= 0 && sc.wait(ST_MAIN, EV_AHEAD, last_ahead)) return V2E_EHIP;
    }
    V2E_HIP(hipGetLastError());
    return 0;
}

// ---- which of the handle's streams share a HARDWARE queue with the caller's (pipelined runs)
// This runtime serves every stream of a process from four in-order hardware queues (more make every cross-queue dependency cost
// 20-50 us: profiles/r06_emulator_experiments.txt item 5), so of the five streams a pipelined run uses at least two share one -- which
// two is decided by how many streams the process happened to have when the handle created its own.  When the chain's stream shares
// its queue with the rows' or the tables' stream, their kernels take turns with the chain's: 1280x720 noisy 10.3 instead of 13.7 Gev/s
// (experiment 32: the same leg behind 0 .. 6 other live streams).  So the roles are given out by MEASUREMENT, once per handle and
// caller's stream: a kernel that spins for 100 us on one stream and a kernel that reads the clock on the other -- on one hardware queue
// the second starts when the first has ended, on two it starts at once.  Roles: k_ahead, tables, rows on three queues that are not the
// chain's; the stream left over (`tabs`, idle in pipelined runs) takes whatever remains.
namespace {
__global__ void k_probe_spin(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16); // (bounded by the clock: 100 MHz ticks)
    out[0] = wall_clock64();
}
__global__ void k_probe_stamp(unsigned long long *out) { out[1] = wall_clock64(); }
} // namespace

// 1: b's kernel waited for a's (one hardware queue, or a == b); 0: it ran beside it; < 0: HIP error
static int streams_share_queue(hipStream_t a, hipStream_t b, unsigned long long *dev)
{
    if (a == b) return 1;
    int shared = 1;
    for (int trial = 0; trial < 2 && shared; ++trial) { // (a late enqueue of the second kernel looks like sharing: one clean trial decides)
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        k_probe_spin<<<1, 64, 0, a>>>(dev, 10000ull);
        k_probe_stamp<<<1, 64, 0, b>>>(dev);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        unsigned long long t[2] = {0, 0};
        if (hipMemcpy(t, dev, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        shared = t[1] >= t[0] ? 1 : 0;
    }
    return shared;
}

static int assign_pipelined_streams(v2e_emu *h, hipStream_t s)
{
    if (getenv("V2E_AMD_NO_QUEUE_PROBE")) return 0;
    h->sync_runs();
    V2E_HIP(hipDeviceSynchronize());
    unsigned long long *dev = nullptr;
    V2E_HIP(hipMalloc(&dev, 2 * sizeof(unsigned long long)));
    std::vector<hipStream_t> cand = {h->ahead, h->side, h->side2, h->tabs};
    for (hipStream_t q : h->spare_streams) cand.push_back(q);
    h->spare_streams.clear();
    while (cand.size() < 8) { // (streams take the least-loaded hardware queue when they are created: eight cover all four)
        hipStream_t q = nullptr;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) break;
        cand.push_back(q);
    }
    std::vector<char> used(cand.size(), 0);
    bool failed = false;
    auto pick = [&](std::initializer_list<hipStream_t> avoid) -> int {
        for (size_t i = 0; i < cand.size(); ++i) {
            if (used[i]) continue;
            bool ok = true;
            for (hipStream_t a : avoid) {
                const int sh = streams_share_queue(a, cand[i], dev);
                if (sh < 0) failed = true;
                if (sh != 0) { ok = false; break; }
            }
            if (ok) { used[i] = 1; return (int)i; }
        }
        return -1;
    };
    int ia = pick({s});
    int it = ia >= 0 ? pick({s, cand[ia]}) : -1;
    int ir = it >= 0 ? pick({s, cand[ia], cand[it]}) : -1;
    if (it >= 0 && ir < 0) ir = pick({s, cand[ia]}); // (three free queues were not found: the rows share the tables' rather than the chain's)
    if (getenv("V2E_AMD_PROBE_DEBUG")) { // dev: the candidates' relation to the caller's stream and to each other
        fprintf(stderr, "queue probe: ahead %d tables %d rows %d failed %d | shares caller:", ia, it, ir, (int)failed);
        for (size_t i = 0; i < cand.size(); ++i) fprintf(stderr, " %d", streams_share_queue(s, cand[i], dev));
        fprintf(stderr, " | with candidate 0:");
        for (size_t i = 0; i < cand.size(); ++i) fprintf(stderr, " %d", streams_share_queue(cand[0], cand[i], dev));
        fprintf(stderr, "\n");
    }
    hipFree(dev);
    if (!failed && ia >= 0 && it >= 0 && ir >= 0) {
        h->ahead = cand[ia]; h->side = cand[it]; h->side2 = cand[ir];
        bool have_tabs = false;
        for (size_t i = 0; i < cand.size(); ++i) {
            if (used[i]) continue;
            if (!have_tabs) { h->tabs = cand[i]; have_tabs = true; }
            else h->spare_streams.push_back(cand[i]);
        }
        h->drop_graphs(); // (captured with the old roles)
    } else { // leave the roles as they were; the new candidates are kept for the handle's destruction
        for (size_t i = 4; i < cand.size(); ++i) h->spare_streams.push_back(cand[i]);
    }
    h->probed = true;
    h->probed_for = s;
    return 0;
}

int v2e_emu_run(v2e_emu *h, const v2e_emu_params *p, const void *frames, int dtype, int n_frames, const double *t_prev,
                const double *t_frame, uint32_t frame_idx0, float *events, uint64_t cap, v2e_frame_rec *recs_dev,
                int use_graph, void *stream)
{
    int rc = check_params(h, p);
    if (rc) return rc;
    V2E_REQUIRE(p->rng_mode == V2E_RNG_PHILOX, "v2e_emu_run is the device-resident Philox path");
    V2E_REQUIRE(frames && t_prev && t_frame && events && recs_dev && n_frames > 0, "bad run args");
    V2E_REQUIRE(dtype == V2E_DT_U8 || dtype == V2E_DT_F32 || dtype == V2E_DT_F64, "bad frame dtype");
    V2E_REQUIRE(!h->cs_sur || ((int)h->csr_steps.size() == n_frames && h->n_clips == 1),
                "a run with a CSDVS surround needs v2e_emu_set_csdvs_run for exactly these frames (one clip)");
    V2E_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (n_frames > h->run_cap) {
        h->sync_runs();
        V2E_HIP(hipStreamSynchronize(s));
        if (h->run_ctl) V2E_HIP(hipFree(h->run_ctl));
        hipFree(h->alt_of((void **)&h->run_ctl));
        h->run_ctl = nullptr; h->alt_of((void **)&h->run_ctl) = nullptr;
        for (int q = 0; q < 2; ++q) if (h->run_ctl_host2[q]) V2E_HIP(hipHostFree(h->run_ctl_host2[q]));
        h->run_cap = n_frames;
        for (int q = 0; q < 2; ++q) V2E_HIP(hipHostMalloc(&h->run_ctl_host2[q], sizeof(FrameCtl) * (size_t)h->run_cap * h->n_clips));
        h->drop_graphs();
    }
    KArgs a = make_kargs(h, p);
    const int mode = use_graph & 3;          // 0 plain launches, 1 hipGraph, 2 instrumented
    // K frames per launch with the state in registers (emu_chain.h) wherever it can run: |256 insists on it; |16 selects the
    // count / rank / scan / emit kernels one frame at a time (kept for A/B; also what carries the photoreceptor-noise plane,
    // emulator.py:694-703, float64 log-encoded frames and more than CHAIN_MAX_ITERS events per pixel and frame)
    const bool chain_ok = chain_eligible(h, p, dtype);
    V2E_REQUIRE(!(use_graph & 256) || chain_ok, "k_chain cannot run this configuration (max_iters, photoreceptor noise or float64 log frames)");
    const bool chain = chain_ok && ((use_graph & 256) != 0 || !(use_graph & 16));
    const bool legacy = !chain;
    // mode 0 | 1024: PIPELINED runs (round 6) -- plain launches on four streams, no graph, no join at the run's end.  The chain of run
    // n + 1 needs the chain of run n and nothing else of it, but a run begins with an upload, zero fills and its first records (k_ahead)
    // and ends with the emission of its last frames: measured with the device stamps (v2e_emu_launch_stamps), 145-165 us of every
    // 820 us step of the benchmark passed between the last chain launch of one run and the first of the next (a graph's end and start
    // cost two cross-queue hops of 20-50 us each on this runtime, and everything of a graph must be complete before the next one
    // starts).  Pipelined: the caller's stream carries the chain's launches and nothing else, run after run in one hardware queue
    // (33 us between two runs' chains); the handle's k_ahead stream carries the run's upload, zero fills and records and runs AHEAD of
    // the chain, beside the run before; the handle's emission stream follows the chain batch by batch and finishes beside the run
    // after.  What two runs in flight would share exists twice (v2e_emu::swap_scratch): such a run takes the set the run before it did not.
    const bool pipelined = (use_graph & 1024) != 0 && mode == 0 && chain;
    if (pipelined) h->swap_scratch();
    h->last_ticket = pipelined ? h->scratch_par : -1;
    if (!h->run_ctl) { V2E_HIP(hipMalloc(&h->run_ctl, sizeof(FrameCtl) * (size_t)h->run_cap * h->n_clips)); h->drop_graphs(); }
    if (chain) {
        rc = chain_alloc(h, p, dtype, n_frames, use_graph);
        if (rc) return rc;
    }
    if (pipelined && (!h->probed || h->probed_for != s)) {
        rc = assign_pipelined_streams(h, s);
        if (rc) return rc;
    }
    const int par = h->scratch_par;
    hipStream_t s_up = s; // the stream of the run's upload
    if (pipelined) {
        for (int q = 0; q < 2; ++q) {
            if (!h->ev_main_done2[q]) V2E_HIP(hipEventCreateWithFlags(&h->ev_main_done2[q], hipEventDisableTiming));
            if (!h->ev_tail_done2[q]) V2E_HIP(hipEventCreateWithFlags(&h->ev_tail_done2[q], hipEventDisableTiming));
        }
        s_up = h->ahead; // (created by chain_alloc)
        // the k_ahead stream reads the frames: it is ordered behind what the caller's stream holds now -- unless the caller vouches
        // that the frames are resident (|2048: nothing enqueued on `stream` still writes them); otherwise it would wait for the chain
        // of the run before, which is on that stream, and the run's head would not overlap it
        if (!(use_graph & 2048)) {
            if (!h->ev_user) V2E_HIP(hipEventCreateWithFlags(&h->ev_user, hipEventDisableTiming));
            V2E_HIP(hipEventRecord(h->ev_user, s));
            V2E_HIP(hipStreamWaitEvent(s_up, h->ev_user, 0));
        }
        // this scratch set's last user (two runs ago): its chain and its emission have finished
        if (h->piece_pending[par]) {
            V2E_HIP(hipStreamWaitEvent(s_up, h->ev_main_done2[par], 0));
            V2E_HIP(hipStreamWaitEvent(s_up, h->ev_tail_done2[par], 0));
        }
        if (h->piece_pending[par ^ 1] && (events == h->last_events || (const void *)recs_dev == h->last_recs))
            V2E_HIP(hipStreamWaitEvent(s_up, h->ev_tail_done2[par ^ 1], 0));
    } else if (h->join_runs(s)) { // everything else runs whole on the caller's stream, behind every piece still in flight
        v2e_set_error("hipStreamWaitEvent failed");
        return V2E_EHIP;
    }
    // two pinned staging sets, each guarded by the event of the upload that last read it: the host prepares run n + 1
    // while run n executes
    const int sq = h->run_stage ^= 1;
    if (!h->ev_stage[sq]) V2E_HIP(hipEventCreateWithFlags(&h->ev_stage[sq], hipEventDisableTiming));
    else V2E_HIP(hipEventSynchronize(h->ev_stage[sq]));
    FrameCtl *ctl_host = h->run_ctl_host2[sq];
    uint32_t *fidx_host = h->run_fidx_host + sq;
    const size_t nct = (size_t)n_frames * h->n_clips;
    double *const tp_host = (double *)ctl_host, *const tf_host = tp_host + nct; // (the staging set holds the frame times)
    memcpy(tp_host, t_prev, sizeof(double) * nct);
    memcpy(tf_host, t_frame, sizeof(double) * nct);
    *fidx_host = frame_idx0;
    const void **frames_host = h->run_frames_host + sq; // the chain pipeline reads the frames' address from a device variable:
    *frames_host = frames;                              // runs over different frame buffers replay the same captured graph
    // The run's frame scalars, first frame index and frames' address go up through a kernel that reads the pinned staging set
    // (round 5; before: three hipMemcpyAsync).  A/B x 3 in one session on the benchmark loop: 11.9 -> 12.15 Gev/s -- in the rocprofv3
    // timeline the copy-engine upload started 40-90 us behind the command before it, the kernel 10 us.
    {
        static_assert(sizeof(FrameCtl) >= 2 * sizeof(double), "the staging set holds two doubles per frame");
        unsigned long long **slot_host = h->stamp_slot_host + sq;
        *slot_host = h->stamps_runs > 0 ? h->stamps + (size_t)(h->stamp_seq++ % (unsigned long long)h->stamps_runs) * 2 * STAMP_LAUNCHES : nullptr;
        k_upload_ctl<<<(unsigned)std::min<size_t>((nct + BLOCK - 1) / BLOCK, 1024), BLOCK, 0, s_up>>>(tp_host, tf_host, h->run_ctl, nct, p->cutoff_hz, p->shot_noise_rate_hz,
                                                                                                   p->refractory_period_s, fidx_host, h->run_fidx, frames_host, h->run_frames,
                                                                                                   slot_host, h->stamp_slot);
        V2E_HIP(hipGetLastError());
    }
    V2E_HIP(hipEventRecord(h->ev_stage[sq], s_up));
    h->last_kind = legacy ? 0 : (h->ch_fused ? 4 : 3);
    h->last_fpl = chain ? h->ch_K : 1;
    h->last_fpb = chain ? h->ch_E : 1;
    if (mode == 0) {
        if (legacy) return enqueue_run(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s, nullptr);
        if (!pipelined) return enqueue_run_chain(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s);
        rc = enqueue_run_chain(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s, nullptr, nullptr, nullptr, false, true);
        if (rc) return rc;
        V2E_HIP(hipEventRecord(h->ev_main_done2[par], s));          // the run's chain (the pixel state)
        hipStream_t rows_stream = h->side2;
        // the run's records to pinned host memory behind its last rows (v2e_emu_run_recs): what result() reads -- no copy of the
        // caller's own, no stream of the caller's to synchronise
        const size_t rbytes = sizeof(v2e_frame_rec) * (size_t)n_frames * h->n_clips;
        if (rbytes > h->recs_host_cap[par]) {
            if (h->recs_host[par]) V2E_HIP(hipHostFree(h->recs_host[par]));
            h->recs_host[par] = nullptr;
            V2E_HIP(hipHostMalloc(&h->recs_host[par], rbytes));
            h->recs_host_cap[par] = rbytes;
        }
        V2E_HIP(hipMemcpyAsync(h->recs_host[par], recs_dev, rbytes, hipMemcpyDeviceToHost, rows_stream));
        h->recs_host_n[par] = (size_t)n_frames * h->n_clips;
        V2E_HIP(hipEventRecord(h->ev_tail_done2[par], rows_stream)); // its last event rows (and the records' copy)
        h->piece_pending[par] = true;
        h->last_events = events; h->last_recs = recs_dev;
        return 0;
    }
    if (mode == 2 && chain) { // instrumented: chain time from events on `s`, emission batches from events on the side stream
        std::vector<hipEvent_t> em, es;
        rc = enqueue_run_chain(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s, nullptr, &em, &es);
        if (rc == 0) {
            V2E_HIP(hipStreamSynchronize(s));
            for (int k = 0; k < 4; ++k) h->prof_ms[k] = 0.0;
            float ms = 0.f;
            V2E_HIP(hipEventElapsedTime(&ms, em.front(), em.back()));
            h->prof_ms[0] = ms; // first launch's start to last launch's end: the chain's launch-to-launch period x launches
            h->prof_chain_us.clear();
            for (size_t i = 0; i + 1 < em.size(); i += 2) { // the kernels alone
                V2E_HIP(hipEventElapsedTime(&ms, em[i], em[i + 1]));
                h->prof_ms[1] += ms;
                h->prof_chain_us.push_back(ms * 1e3f);
            }
            h->prof_step_launches = (int)(em.size() / 2);
            for (size_t i = 0; i + 1 < es.size(); i += 2) {
                V2E_HIP(hipEventElapsedTime(&ms, es[i], es[i + 1]));
                h->prof_ms[3] += ms;
            }
            h->prof_launches = n_frames;
            h->prof_emit_batches = (int)(es.size() / 2);
        }
        for (hipEvent_t e : em) hipEventDestroy(e);
        for (hipEvent_t e : es) hipEventDestroy(e);
        return rc;
    }
    if (mode == 2) { // instrumented: a hipEvent before every launch; blocking
        const int ne = 4 * n_frames + 4;
        std::vector<hipEvent_t> evs(ne);
        for (int i = 0; i < ne; ++i) V2E_HIP(hipEventCreate(&evs[i]));
        rc = enqueue_run(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s, evs.data());
        if (rc == 0) {
            V2E_HIP(hipStreamSynchronize(s));
            for (int k = 0; k < 4; ++k) h->prof_ms[k] = 0.0;
            for (int i = 0; i < 4 * n_frames; ++i) {
                float ms = 0.f;
                V2E_HIP(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
                h->prof_ms[i & 3] += ms;
            }
            h->prof_launches = n_frames;
            h->prof_emit_batches = 0;
            h->prof_step_launches = n_frames;
        }
        for (int i = 0; i < ne; ++i) hipEventDestroy(evs[i]);
        return rc;
    }

    // graph path: everything baked into the graph is part of the cache key
    std::vector<unsigned char> key;
    auto push = [&key](const void *ptr, size_t n) { const unsigned char *b = (const unsigned char *)ptr; key.insert(key.end(), b, b + n); };
    push(&a, sizeof(a)); push(&dtype, sizeof(dtype)); push(&n_frames, sizeof(n_frames));
    if (legacy) push(&frames, sizeof(frames)); // the chain pipeline reads the frames' address from h->run_frames
    push(&events, sizeof(events)); push(&cap, sizeof(cap)); push(&recs_dev, sizeof(recs_dev));
    int f64 = p->f64_state; push(&f64, sizeof(f64));
    int lg = legacy ? 1 : 3; push(&lg, sizeof(lg)); push(&h->dbg, sizeof(h->dbg));
    push(&h->run_ctl, sizeof(h->run_ctl)); push(&h->run_fidx, sizeof(h->run_fidx)); push(&h->run_off, sizeof(h->run_off)); // (per scratch set)
    { const int mk = h->stamps_runs > 0 ? 1 : 0; push(&mk, sizeof(mk)); }
    if (h->cs_sur) { // everything the diffuser's launches bake in
        push(&h->cs_sur, sizeof(h->cs_sur)); push(&h->csr_scratch, sizeof(void *)); push(&h->csr_lp, sizeof(void *));
        push(&h->csr_steps_dev, sizeof(void *)); push(&h->csr_slots, sizeof(void *)); push(&h->csr_thr, sizeof(double));
        push(h->csr_ap.data(), sizeof(double) * h->csr_ap.size()); push(h->csr_ah.data(), sizeof(double) * h->csr_ah.size());
        push(h->csr_steps.data(), sizeof(int) * h->csr_steps.size());
    }
    if (chain) {
        push(&h->ch_K, sizeof(h->ch_K)); push(&h->ch_E, sizeof(h->ch_E)); push(&h->ch_fused, sizeof(h->ch_fused)); push(&h->ch_nD, sizeof(h->ch_nD));
        push(&h->ch_max_blocks, sizeof(h->ch_max_blocks));
        push(&h->ch_gM, sizeof(h->ch_gM)); push(&h->ch_tsold, sizeof(h->ch_tsold)); push(&h->ch_ck, sizeof(h->ch_ck)); push(&h->ch_cnt, sizeof(h->ch_cnt));
    }
    auto get_exec = [&](int phase, hipGraphExec_t *out) -> int { // the cached graph of the run
        std::vector<unsigned char> k2 = key;
        k2.push_back((unsigned char)phase);
        for (auto &cg : h->graphs)
            if (cg.key == k2) { *out = cg.exec; cg.used = ++h->graph_clock; return 0; }
        if (h->graphs.size() >= 24) { // drop the least recently used one (two scratch sets x the caller's buffer sets x the pieces of a run)
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].used < h->graphs[lru].used) lru = i;
            hipGraphExecDestroy(h->graphs[lru].exec);
            h->graphs.erase(h->graphs.begin() + lru);
        }
        hipGraph_t g = nullptr;
        int rc2 = 0;
        if (legacy) { // one stream: plain stream capture
            hipStream_t cs;
            V2E_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            V2E_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            rc2 = enqueue_run(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, cs, nullptr);
            hipError_t e = hipStreamEndCapture(cs, &g);
            hipStreamDestroy(cs);
            if (rc2) { if (g) hipGraphDestroy(g); return rc2; }
            V2E_HIP(e);
        } else if (getenv("V2E_AMD_GRAPH_EXPLICIT")) { // dev: the graph node by node (see Sched); this runtime then runs it serially
            V2E_HIP(hipGraphCreate(&g, 0));
            rc2 = enqueue_run_chain(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, s, g);
            if (rc2) { hipGraphDestroy(g); return rc2; }
        } else { // stream capture, with edges between the origin and the forked streams only
            hipStream_t cs;
            V2E_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            V2E_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            rc2 = enqueue_run_chain(h, p, a, frames, dtype, n_frames, events, cap, recs_dev, cs, nullptr, nullptr, nullptr, true);
            hipError_t e = hipStreamEndCapture(cs, &g);
            hipStreamDestroy(cs);
            if (rc2) { if (g) hipGraphDestroy(g); return rc2; }
            V2E_HIP(e);
        }
        hipGraphExec_t exec = nullptr;
        V2E_HIP(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        V2E_HIP(hipGraphDestroy(g));
        h->graphs.push_back({k2, exec, ++h->graph_clock});
        *out = exec;
        return 0;
    };
    hipGraphExec_t exec = nullptr;
    rc = get_exec(0, &exec);
    if (rc) return rc;
    V2E_HIP(hipGraphLaunch(exec, s));
    return 0;
}

// host-blocking: every piece of the LAST overlapped run that used scratch set `ticket` (v2e_emu_run_ticket right after that run) is complete
int v2e_emu_run_wait(v2e_emu *h, int ticket)
{
    V2E_REQUIRE(h && (ticket == 0 || ticket == 1), "bad ticket");
    if (h->piece_pending[ticket]) {
        V2E_HIP(hipEventSynchronize(h->ev_main_done2[ticket]));
        V2E_HIP(hipEventSynchronize(h->ev_tail_done2[ticket]));
    }
    return 0;
}
int v2e_emu_run_ticket(v2e_emu *h) { return h ? h->last_ticket : -1; }
int v2e_emu_event_writer(v2e_emu *h) { return (h && h->ch_K > 0) ? (h->ch_pull ? 1 : 0) : -1; }
const v2e_frame_rec *v2e_emu_run_recs(v2e_emu *h, int ticket, uint64_t *n_recs)
{
    if (!h || (ticket != 0 && ticket != 1)) return nullptr;
    if (n_recs) *n_recs = h->recs_host_n[ticket];
    return (const v2e_frame_rec *)h->recs_host[ticket];
}

// `stream` waits for every piece of every overlapped run (v2e_emu_run, |1024) enqueued on this handle so far
int v2e_emu_run_join(v2e_emu *h, void *stream)
{
    V2E_REQUIRE(h, "null");
    V2E_HIP(hipSetDevice(h->device));
    if (h->join_runs((hipStream_t)stream)) { v2e_set_error("hipStreamWaitEvent failed"); return V2E_EHIP; }
    return 0;
}

int v2e_emu_launch_stamps(v2e_emu *h, int runs, uint64_t *out_ns, int cap_runs, int *n_runs, int *launches_per_run)
{
    V2E_REQUIRE(h && runs >= 0, "bad args");
    V2E_HIP(hipSetDevice(h->device));
    if (out_ns || n_runs) { // read back (blocking): the last min(stamp_seq, stamps_runs) runs, oldest first
        V2E_HIP(hipDeviceSynchronize());
        const unsigned long long have = std::min<unsigned long long>(h->stamp_seq, (unsigned long long)h->stamps_runs);
        const int n = (int)std::min<unsigned long long>(have, (unsigned long long)std::max(cap_runs, 0));
        if (n_runs) *n_runs = n;
        if (launches_per_run) *launches_per_run = STAMP_LAUNCHES;
        if (out_ns && n > 0) {
            std::vector<unsigned long long> tmp((size_t)h->stamps_runs * 2 * STAMP_LAUNCHES);
            V2E_HIP(hipMemcpy(tmp.data(), h->stamps, tmp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int r = 0; r < n; ++r) {
                const unsigned long long seq = h->stamp_seq - (unsigned long long)n + (unsigned long long)r;
                const unsigned long long *src = tmp.data() + (size_t)(seq % (unsigned long long)h->stamps_runs) * 2 * STAMP_LAUNCHES;
                for (int l = 0; l < STAMP_LAUNCHES; ++l) { // wall clock: 100 MHz (s_memrealtime); the start is kept complemented
                    const unsigned long long a = src[2 * l], b = src[2 * l + 1];
                    out_ns[((size_t)r * STAMP_LAUNCHES + l) * 2] = a ? (~a) * 10ull : 0ull;
                    out_ns[((size_t)r * STAMP_LAUNCHES + l) * 2 + 1] = b * 10ull;
                }
            }
        }
    }
    if (runs != h->stamps_runs) {
        V2E_HIP(hipDeviceSynchronize());
        hipFree(h->stamps);
        h->stamps = nullptr;
        h->stamps_runs = 0;
        h->stamp_seq = 0;
        if (runs > 0) {
            V2E_HIP(hipMalloc(&h->stamps, sizeof(unsigned long long) * (size_t)runs * 2 * STAMP_LAUNCHES));
            V2E_HIP(hipMemset(h->stamps, 0, sizeof(unsigned long long) * (size_t)runs * 2 * STAMP_LAUNCHES));
            V2E_HIP(hipDeviceSynchronize()); // (the fill is ordered on the default stream only: see chain_alloc)
            h->stamps_runs = runs;
        }
    }
    return 0;
}

// dev tool (not part of the public header): enable / read back the in-kernel timeline of one mid-run k_chain launch
int v2e_emu_debug_timeline(v2e_emu *h, unsigned long long *out_host /* [ngroups][16] or NULL to enable */, int *ngroups)
{
    V2E_REQUIRE(h, "null");
    V2E_HIP(hipSetDevice(h->device));
    if (!h->dbg) {
        V2E_HIP(hipMalloc(&h->dbg, sizeof(unsigned long long) * 16 * h->ngroups));
        V2E_HIP(hipMemset(h->dbg, 0, sizeof(unsigned long long) * 16 * h->ngroups));
        h->drop_graphs();
    }
    if (ngroups) *ngroups = h->ngroups;
    if (out_host) {
        V2E_HIP(hipDeviceSynchronize());
        V2E_HIP(hipMemcpy(out_host, h->dbg, sizeof(unsigned long long) * 16 * h->ngroups, hipMemcpyDeviceToHost));
    }
    return 0;
}

// dev tool (not part of the public header; scripts/chain_rounds.py): the rule-on rows of the last run's chain launches,
// [launches][K + 1 rows][n_clips][K] -- row 0 what a launch's own pass flagged, row r what the r-th redo pass on it flagged
int v2e_emu_debug_chain_rows(v2e_emu *h, uint32_t *out, size_t cap_words, int *launches, int *K)
{
    V2E_REQUIRE(h && launches && K, "null");
    *launches = h->ch_launch_cap; *K = h->ch_K;
    if (!h->ch_gM || !out) return 0;
    V2E_HIP(hipSetDevice(h->device));
    V2E_HIP(hipDeviceSynchronize());
    const size_t n = (size_t)h->ch_launch_cap * (h->ch_K + 1) * h->n_clips * h->ch_K;
    V2E_HIP(hipMemcpy(out, h->ch_gM, sizeof(uint32_t) * std::min(n, cap_words), hipMemcpyDeviceToHost));
    return 0;
}

int v2e_emu_last_profile(v2e_emu *h, double *ms_count, double *ms_rank, double *ms_scan, double *ms_emit, int *launches)
{
    V2E_REQUIRE(h && ms_count && ms_rank && ms_scan && ms_emit && launches, "null");
    *ms_count = h->prof_ms[0]; *ms_rank = h->prof_ms[1]; *ms_scan = h->prof_ms[2]; *ms_emit = h->prof_ms[3];
    *launches = h->prof_launches;
    return 0;
}

int v2e_emu_chain_plan(int n_frames, int frames_per_launch, int frames_per_batch, int ring_batches, int has_refractory,
                       int fused_records, int32_t *out, int cap)
{
    V2E_REQUIRE(n_frames > 0 && frames_per_launch >= 1 && frames_per_launch <= CHAIN_K_MAX && frames_per_batch >= frames_per_launch &&
                frames_per_batch % frames_per_launch == 0 && ring_batches >= 3, "bad plan arguments");
    const std::vector<ChainLaunch> plan = chain_plan(n_frames, frames_per_launch, frames_per_batch, ring_batches, has_refractory != 0,
                                                     fused_records != 0);
    if (out) {
        V2E_REQUIRE(cap >= (int)plan.size(), "plan buffer too small");
        for (size_t i = 0; i < plan.size(); ++i) {
            const ChainLaunch &c = plan[i];
            const int32_t row[8] = {c.f0, c.nf, c.pf0, c.pnf, c.wait_join, c.wait_ahead, c.ahead_next, c.emit_batch};
            memcpy(out + 8 * i, row, sizeof(row));
        }
    }
    return (int)plan.size();
}

int v2e_emu_last_pipeline(v2e_emu *h, int *kind, int *frames_per_launch, int *frames_per_batch)
{
    V2E_REQUIRE(h && kind && frames_per_launch && frames_per_batch, "null");
    *kind = h->last_kind; *frames_per_launch = h->last_fpl; *frames_per_batch = h->last_fpb;
    return 0;
}

int v2e_emu_last_profile_launches(v2e_emu *h, float *us, int cap, int *n)
{
    V2E_REQUIRE(h && n, "null");
    *n = (int)h->prof_chain_us.size();
    for (int i = 0; us && i < cap && i < *n; ++i) us[i] = h->prof_chain_us[i];
    return 0;
}

int v2e_emu_last_profile_pipe(v2e_emu *h, int *emit_batches, int *frames_per_batch, int *step_launches)
{
    V2E_REQUIRE(h && emit_batches && frames_per_batch, "null");
    *emit_batches = h->prof_emit_batches;
    *frames_per_batch = h->last_fpb;
    if (step_launches) *step_launches = h->prof_step_launches;
    return 0;
}

} // extern "C"
