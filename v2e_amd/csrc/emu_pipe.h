// emu_pipe.h -- decoupled device-resident pipeline of the DVS pixel model (included by emu.hip).
//
// What the next frame needs from the previous one is only the pixel's own base_log_frame /
// timestamp_mem update (emulator.py:936-942), and that needs, beyond the pixel's own counts, one
// global number: the frame's max count M (it fixes the timestamps, hence the refractory test).
// Where an event lands in the output list -- the prefix over all pixels, the per-iteration shuffle
// -- is needed by nobody downstream.  So the frame-to-frame dependency chain is kept to
//
//   k_step(f)  = finalise(f-1) [M(f-1) from the published workgroup maxima; refractory pass count;
//                base / ts_mem update] + count(f) [lin-log, low-pass, leak, floor-divide, shot
//                noise, packed count word to a ring slot, workgroup max]
//
// one launch per frame, and everything about the event list runs behind it on a second stream,
// E frames per launch, overlapping the chain:
//
//   k_tot_multi(f0..f0+E)   per-workgroup (iteration, polarity) totals, refractory rule applied
//   [k_scan2_multi          large grids: prefixes over workgroups]
//   k_emit_multi(f0..f0+E)  offsets, timestamps, shuffle, float4 rows, frame records
//
// (scripts/ubench_gridbar.hip: an in-kernel grid rendezvous costs 11-15 us on MI355X, a dependent
// launch 3.5 us, so launch boundaries are the grid synchronisation.)  A ring of D = 2E frame slots
// holds what the emission side reads: count words, workgroup maxima, and -- for frames on which
// the refractory rule was active -- the ts_mem plane as it was before the update.
#pragma once

// frames per emission launch E (v2e_emu::pipe_E, chosen at create time from the ring's footprint) and
// ring slots D = 2E: the emission of batch b overlaps the steps of batch b+1
constexpr int PIPE_E_MAX = 32;
#ifndef WT_EVENTS
#define WT_EVENTS 1
#endif

// The run's frame-index base lives in device memory (a captured graph is replayed with a new base), and the
// Philox draw needs it first thing.  As a compiler-scheduled scalar load it ends up a dependent round trip in
// front of every vector load; as a vector load it retires in order with them, so the draw could not overlap
// their latency.  Hence by hand: issue the s_load early, wait for it (lgkmcnt) only where the draw starts.
__device__ __forceinline__ uint32_t sload_u32_issue(const uint32_t *ptr)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(ptr) : "memory");
    return v; // NOT valid until sload_wait(v)
}
__device__ __forceinline__ uint32_t sload_wait(uint32_t v)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
    return v;
}

struct StepArgs {
    const void *frame;       // frame f (count)
    const FrameCtl *ctl_c;   // times of frame f
    const FrameCtl *ctl_e;   // times of frame f-1
    const uint32_t *fidx_base;
    uint32_t fidx_c;
    int do_final, do_count, ngroups;
    const uint32_t *cnt_e;   // [n_clips][npx_pad] count words of frame f-1
    uint32_t *cnt_c;         //                    ... of frame f
    const int *gmax_e;       // [n_clips][ngroups] workgroup maxima of frame f-1
    int *gmax_c;
    float *tsold_e;          // [n_clips][npx_pad] slot of frame f-1 (refractory frames only) or nullptr
    unsigned long long *dbg;
};

#define V2E_STAMP_S(i) do { if (sa.dbg && tid == 0) sa.dbg[(size_t)g * 16 + (i)] = wall_clock64(); } while (0)

template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_step(KArgs a, StepArgs sa)
{
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ float s_lutL[256];
    __shared__ double s_lutI[256];
    constexpr bool U8 = sizeof(FT) == 1;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, g = blockIdx.x;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = (size_t)clip * a.npx_pad + p;

    __builtin_amdgcn_s_setprio(3); // the dependency chain outranks the emission waves sharing the SIMD
    const uint32_t fbase_pending = sload_u32_issue(sa.fidx_base);
    V2E_STAMP_S(0);
    // ------------------------------------------------------------ every load of the launch, back to back
    // (vmcnt retires in order: a wait on any of them is a wait on all earlier ones, so nothing below may
    // consume a loaded value before the last load has been issued -- hence registers, not loops/LDS, here)
    R b = (R)0, lp_old = (R)0;
    float thp = 1.f, thn = 1.f, nr = 0.f, tsm = 0.f;
    uint32_t cw_e = 0;
    FT px = (FT)0;
    if (valid) {
        b = ((R *)a.base)[sp];
        thp = a.pos_thres[sp];
        thn = a.neg_thres[sp];
        if (a.has_cutoff || (sa.do_final && a.do_shot)) lp_old = ((R *)a.lp)[sp];
        if (a.do_leak && sa.do_count) nr = a.noise_rate[sp];
        if (a.has_refr && sa.do_final) tsm = a.ts_mem[sp];
        if (sa.do_final) cw_e = sa.cnt_e[sp];
        if (sa.do_count) px = ((const FT *)sa.frame)[(size_t)clip * a.npx + p];
    }
    float lutL_r = 0.f;
    double lutI_r = 0.0;
    if (U8 && sa.do_count) {
        lutL_r = a.lut_L[tid];
        lutI_r = a.lut_I[tid];
    }
    int gm0 = 0, gm1 = 0, gm2 = 0, gm3 = 0; // workgroup maxima of frame f-1: four per thread cover 1024 workgroups
    const int *gmv = sa.gmax_e + (size_t)clip * sa.ngroups;
    if (sa.do_final) {
        if (tid < sa.ngroups) gm0 = gmv[tid];
        if (tid + BLOCK < sa.ngroups) gm1 = gmv[tid + BLOCK];
        if (tid + 2 * BLOCK < sa.ngroups) gm2 = gmv[tid + 2 * BLOCK];
        if (tid + 3 * BLOCK < sa.ngroups) gm3 = gmv[tid + 3 * BLOCK];
    }
    float tab_start = 0.f, tab_step = 0.f, tab_end = 0.f;
    uint32_t refr_mask = 0;
    if (sa.do_final && a.has_refr) {
        const FrameCtl *ce = sa.ctl_e + clip;
        tab_start = ce->ts_start[lane & 31];
        tab_step = ce->ts_stepf[lane & 31];
        refr_mask = ce->refr_mask;
        tab_end = ce->ts_end;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- arithmetic that depends on no memory but fbase, done while those loads are in flight
    const uint32_t fbase = sload_wait(fbase_pending);
    float rng_r = 0.f, rng_u = 0.f;
    if (sa.do_count && valid && ((a.do_leak && a.jit_f != 0.f) || a.do_shot))
        v2e_draw_frame(a.seed, (uint32_t)clip, fbase + sa.fidx_c, (uint32_t)p, &rng_r, &rng_u);
    V2E_STAMP_S(5);
    if (U8 && sa.do_count) { // published by the barriers of block_max_finish / the one below
        s_lutL[tid] = lutL_r;
        s_lutI[tid] = lutI_r;
    }
    int gm_part = max(max(gm0, gm1), max(gm2, gm3));
    if (sa.do_final)
        for (int k = tid + 4 * BLOCK; k < sa.ngroups; k += BLOCK) gm_part = max(gm_part, gmv[k]);
    bool b_dirty = false;

    // ------------------------------------------------------------ finalise(f-1): emulator.py:830-842, 936-942
    if (sa.do_final) {
        if (sa.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); V2E_STAMP_S(6); }
        const int M = block_max_finish(gm_part, s_red, lane, wave); // its barriers also publish the LUT
        V2E_STAMP_S(1);
        if (M <= a.max_iters) { // beyond max_iters the run is flagged by the emission side and discarded
            const int n = M > 0 ? M : 1;
            const uint32_t cw = cw_e;
            const int mag = (int)(cw & CNT_MASK);
            const bool neg = (cw & CNT_NEG) != 0;
            int fcount = mag;
            if (a.has_refr) {
                bool use_refr;
                TsGen tg(0.f, 0.f, 0.f, n);
                if (n <= 32) {
                    tg = TsGen(__uint_as_float(lane_value(__float_as_uint(tab_start), n - 1)), tab_end,
                               __uint_as_float(lane_value(__float_as_uint(tab_step), n - 1)), n);
                    use_refr = (refr_mask >> (n - 1)) & 1u;
                } else {
                    const FrameCtl c = sa.ctl_e[clip];
                    tg = TsGen(c, n, nullptr);
                    use_refr = a.refr > (c.t_frame - c.t_prev) / (double)n;
                }
                if (use_refr) {
                    // the emission side re-derives which iterations passed from ts_mem as it was
                    if (valid) sa.tsold_e[sp] = tsm;
                    fcount = 0;
                    for (int i = 0; i < mag; ++i) {
                        const float t = tg(i);
                        const float pt = 1.0f * t - tsm;
                        if (pt > a.refr_f) { tsm = t; ++fcount; }
                    }
                    if (valid && fcount > 0) a.ts_mem[sp] = tsm;
                }
            }
            if (valid) {
                const bool shot = a.do_shot && (cw & (CNT_SHOT_ON | CNT_SHOT_OFF));
                if (fcount > 0 || shot) {
                    const float dp = (float)(neg ? 0 : fcount) * thp;
                    const float dn = (float)(neg ? fcount : 0) * thn;
                    b = b + (R)dp;
                    b = b - (R)dn;
                    if (shot) b = lp_old;
                    b_dirty = true;
                }
            }
        }
    } else if (U8 && sa.do_count) {
        __syncthreads(); // LUT visible
    }
    V2E_STAMP_S(2);

    // ------------------------------------------------------------ count(f)
    if (sa.do_count) {
        const FrameCtl c = sa.ctl_c[clip];
        const double delta_time = c.t_frame - c.t_prev;
        int m = 0;
        uint32_t cw = 0;
        if (valid) {
            double L; // lin-log of the frame, or the frame itself when it is log-encoded already (emulator.py:666)
            double inten01;
            if (U8) {
                L = a.log_input ? (double)px : (double)s_lutL[(int)px];
                inten01 = s_lutI[(int)px];
            } else {
                const double x = (double)px;
                L = a.log_input ? x : (double)lin_log(x);
                inten01 = a.use_inten ? (x + 20.0) / 275.0 : 0.0;
            }
            const float r = rng_r, u = rng_u;
            R lpn;
            if (a.has_cutoff) {
                double eps = inten01 * c.dt_over_tau;
                if (eps > 1.0) eps = 1.0;
                lpn = (R)((1.0 - eps) * (double)lp_old + eps * (double)L);
            } else {
                lpn = (R)L;
            }
            ((R *)a.lp)[sp] = lpn;
            if (a.do_leak) { // emulator_utils.py:126-129, float32 left to right
                const float rate = (a.leak_hz_f * nr) * (1.0f - a.jit_f * r);
                const float delta_leak = ((float)delta_time * rate) * thp;
                b = b - (R)delta_leak;
                b_dirty = true;
            }
            if (b_dirty) ((R *)a.base)[sp] = b;
            const R diff = (lpn + (R)0.0f) - b;
            const R pf = diff > (R)0 ? diff : (R)0;
            const R nf = (-diff) > (R)0 ? -diff : (R)0;
            const R tpd = a.scalar_thres ? (R)a.pos_div : (R)thp;
            const R tnd = a.scalar_thres ? (R)a.neg_div : (R)thn;
            // diff has one sign, so one of pf/nf is zero and floor(0/thr) = 0: one exact floor
            // division serves both torch.div(..., rounding_mode='floor') calls (emulator_utils.py:154-157)
            const bool is_pos = diff > (R)0;
            const int q = (int)floor_div_pos<R>(is_pos ? pf : nf, is_pos ? tpd : tnd);
            const int pc = is_pos ? q : 0, nc = is_pos ? 0 : q;
            if (pc > 0) cw = (uint32_t)pc & CNT_MASK;
            else if (nc > 0) cw = ((uint32_t)nc & CNT_MASK) | CNT_NEG;
            if (a.do_shot) cw |= shot_bits(a, inten01, c.shot_base, thp, thn, u);
            sa.cnt_c[sp] = cw;
            m = pc > nc ? pc : nc;
        }
        V2E_STAMP_S(3);
        m = wave_max_i32(m);
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        if (tid == 0) sa.gmax_c[(size_t)clip * sa.ngroups + g] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    } else if (valid && b_dirty) {
        ((R *)a.base)[sp] = b;
    }
    V2E_STAMP_S(4);
}

// ------------------------------------------------------------------ emission side
struct EmitArgs {
    const FrameCtl *ctl;     // [n_frames][n_clips] of the run
    v2e_frame_rec *recs;     // [n_frames][n_clips]
    const uint32_t *fidx_base;
    int f0, nE, D;           // run-relative frames f0 .. f0+nE-1, frame f0+z in grid plane z; ring slot = frame % D
    int ngroups, ngp, n_clips;
    const uint32_t *cnt;     // ring bases (slot = frame % D)
    const int *gmax;
    const float *tsold;
    uint16_t *gtT;           // [D][n_clips][nkeys_cap][ngp] per-workgroup key totals, key-major u16
    int *rowext;             // [D][n_clips][ngroups] iterations up to which this workgroup's rows are non-zero
    uint32_t *nw;            // [D][n_clips][ngroups] events of the workgroup in the slot's frame
    uint32_t *pre32, *tot32; // large grids: [E][n_clips][nkeys_cap][ngp] / [E][n_clips][nkeys_cap]
    float4 *events;
    unsigned long long cap;
    const unsigned long long *off_in; // [n_clips] event offset at the start of this batch
    unsigned long long *off_out;      // [n_clips] ... of the next one
    // small grids (ngp == 512): per-frame tables built once by k_frame_multi instead of by every wave
    struct FrameTable *ftab;          // [E][n_clips]
    uint32_t *pre512;                 // [E][n_clips][FT_KEYS][512] exclusive prefix over workgroups per key
    int capw;                         // event records per wave that fit the dynamic LDS of k_emit2_multi
};

constexpr int FT_KEYS = 64; // keys covered by the frame table: shot pair + iterations 0..30
struct FrameTable {
    int M;                      // max count of the frame
    uint32_t n_events;          // all events of the frame (sum of the workgroups' counts)
    uint32_t big;               // M > 31 (or > max_iters): the frame goes through k_emit_multi instead
    uint32_t pad[5];
    uint32_t T[FT_KEYS];        // events per key over all workgroups
    uint32_t kbase[FT_KEYS];    // first row of the key's iteration within the frame (prefix over signal keys)
    uint32_t perm[32][8];       // per iteration: shuffle round keys k0..k3, sh, a, amask, n
};

// Timestamps and refractory switch of one frame once its max count n is known (block-uniform).
// The per-n tables of the frame's FrameCtl are fetched one entry per lane BEFORE n is known
// (FrameTab), then broadcast: no memory round trip and no float64 division after the reduction.
struct FrameTab {
    float start, step, end;
    uint32_t refr_mask;
    __device__ __forceinline__ FrameTab(const FrameCtl *c, int lane)
        : start(c->ts_start[lane & 31]), step(c->ts_stepf[lane & 31]), end(c->ts_end), refr_mask(c->refr_mask) {}
};

__device__ __forceinline__ TsGen frame_tsgen(const KArgs &a, const FrameCtl *c, const FrameTab &ft, int n, bool &use_refr)
{
    if (n <= 32) {
        use_refr = a.has_refr && ((ft.refr_mask >> (n - 1)) & 1u);
        return TsGen(__uint_as_float(lane_value(__float_as_uint(ft.start), n - 1)), ft.end,
                     __uint_as_float(lane_value(__float_as_uint(ft.step), n - 1)), n);
    }
    const double t_prev = c->t_prev, t_frame = c->t_frame;
    use_refr = a.has_refr && a.refr > (t_frame - t_prev) / (double)n;
    FrameCtl cc;
    cc.t_prev = t_prev; cc.t_frame = t_frame;
    return TsGen(cc, n, nullptr);
}

// ------------------------------------------------------------------ two frames per launch
// The launch-to-launch gap (~3.9 us) is longer than k_step itself (~2.5 us), so k_step2 takes two frames per
// launch: after the exact finalise of the previous launch's last frame it counts frame c0, finalises it
// SPECULATIVELY -- assuming the refractory rule is off for c0, which is all that needs the global max M(c0):
// then the pixel's pass count is its own count -- and counts frame c1.  The next launch knows M(c0) and
// validates: if the rule was in fact on (refractory_period_s > dt / M, about 1 % of the frames of the
// benchmark clip), it restores base_log_frame from the checkpoint taken before the speculation, finalises c0
// exactly, re-counts c1, republishes the workgroup maxima and runs ONE in-kernel grid rendezvous (clip_barrier,
// 11-15 us, bounded spin) to learn the corrected M(c1).  lp_log_frame never depends on the speculation.
// Needs every workgroup of the grid co-resident (checked by the host; small grids only).
struct Step2Args {
    const void *frame0, *frame1;     // frames c0, c1 (nullptr: absent)
    const FrameCtl *ctl_c0, *ctl_c1; // their times
    const FrameCtl *ctl_e1, *ctl_e2; // times of e1 (exact finalise; = c0 - 1) and e2 (validate; = c0 - 2)
    const uint32_t *fidx_base;
    uint32_t fidx_c0, fidx_c1, fidx_e1; // run-relative frame indices
    int has_e1, has_e2, ngroups;
    const uint32_t *cnt_e2;
    uint32_t *cnt_e1, *cnt_c0, *cnt_c1;
    const int *gmax_e2;
    int *gmax_e1, *gmax_c0, *gmax_c1;
    float *tsold_e2, *tsold_e1;
    const void *bck_e2, *lpn_e2;     // checkpoint of e2: base before its finalise, lp after it
    void *bck_c0, *lpn_c0;
    unsigned *bar;                   // [n_clips] rendezvous counters of this launch (zeroed at run start)
    v2e_frame_rec *rec_e1;           // record of e1: V2E_FLAG_SYNC_TIMEOUT lands here
    unsigned long long *dbg;
};

// emulator.py:830-842, 936-942 for one pixel; M <= max_iters is the caller's business
template <typename R>
__device__ __forceinline__ void finalize_px(const KArgs &a, uint32_t cw, bool use_refr, const TsGen &tg, bool valid, size_t sp, float thp,
                                            float thn, R lp_shot, float *tsold_slot, R &b, float &tsm, bool &b_dirty)
{
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    int fcount = mag;
    if (use_refr) {
        if (valid) tsold_slot[sp] = tsm; // the emission side re-derives which iterations passed from ts_mem as it was
        fcount = 0;
        for (int i = 0; i < mag; ++i) {
            const float t = tg(i);
            const float pt = 1.0f * t - tsm;
            if (pt > a.refr_f) { tsm = t; ++fcount; }
        }
        if (valid && fcount > 0) a.ts_mem[sp] = tsm;
    }
    if (valid) {
        const bool shot = a.do_shot && (cw & (CNT_SHOT_ON | CNT_SHOT_OFF));
        if (fcount > 0 || shot) {
            const float dp = (float)(neg ? 0 : fcount) * thp;
            const float dn = (float)(neg ? fcount : 0) * thn;
            b = b + (R)dp;
            b = b - (R)dn;
            if (shot) b = lp_shot;
            b_dirty = true;
        }
    }
}

// emulator_utils.py:137-173 for one pixel: signed count from lp and base (leak already applied)
template <typename R>
__device__ __forceinline__ uint32_t count_px(const KArgs &a, R lpn, R b, float thp, float thn, int &m)
{
    const R diff = (lpn + (R)0.0f) - b;
    const R pf = diff > (R)0 ? diff : (R)0;
    const R nf = (-diff) > (R)0 ? -diff : (R)0;
    const R tpd = a.scalar_thres ? (R)a.pos_div : (R)thp;
    const R tnd = a.scalar_thres ? (R)a.neg_div : (R)thn;
    const bool is_pos = diff > (R)0;
    const int q = (int)floor_div_pos<R>(is_pos ? pf : nf, is_pos ? tpd : tnd);
    m = q;
    if (q <= 0) return 0u;
    return is_pos ? ((uint32_t)q & CNT_MASK) : (((uint32_t)q & CNT_MASK) | CNT_NEG);
}

// lin-log / inten01 / low-pass / leak of one pixel for one frame (emulator_utils.py:18-134)
template <typename R, typename FT>
__device__ __forceinline__ R photoreceptor_px(const KArgs &a, FT px, const float *s_lutL, const double *s_lutI, const FrameCtl &c, R lp_old,
                                              float nr, float thp, float r, double &inten01, R &b, bool &b_dirty)
{
    constexpr bool U8 = sizeof(FT) == 1;
    double L; // lin-log of the frame, or the frame itself when it is log-encoded already (emulator.py:666)
    if (U8) {
        L = a.log_input ? (double)px : (double)s_lutL[(int)px];
        inten01 = s_lutI[(int)px];
    } else {
        const double x = (double)px;
        L = a.log_input ? x : (double)lin_log(x);
        inten01 = a.use_inten ? (x + 20.0) / 275.0 : 0.0;
    }
    R lpn;
    if (a.has_cutoff) {
        double eps = inten01 * c.dt_over_tau;
        if (eps > 1.0) eps = 1.0;
        lpn = (R)((1.0 - eps) * (double)lp_old + eps * (double)L);
    } else {
        lpn = (R)L;
    }
    if (a.do_leak) { // emulator_utils.py:126-129, float32 left to right
        const double delta_time = c.t_frame - c.t_prev;
        const float rate = (a.leak_hz_f * nr) * (1.0f - a.jit_f * r);
        const float delta_leak = ((float)delta_time * rate) * thp;
        b = b - (R)delta_leak;
        b_dirty = true;
    }
    return lpn;
}

__device__ __forceinline__ void block_max2_finish(int &m1, int &m2, int (*s_red2)[BLOCK / WAVE], int lane, int wave)
{
    m1 = wave_max_i32(m1);
    m2 = wave_max_i32(m2);
    if (lane == 0) { s_red2[0][wave] = m1; s_red2[1][wave] = m2; }
    __syncthreads();
    m1 = max(max(s_red2[0][0], s_red2[0][1]), max(s_red2[0][2], s_red2[0][3]));
    m2 = max(max(s_red2[1][0], s_red2[1][1]), max(s_red2[1][2], s_red2[1][3]));
    __syncthreads();
    m1 = __builtin_amdgcn_readfirstlane(m1);
    m2 = __builtin_amdgcn_readfirstlane(m2);
}

#define V2E_STAMP_2(i) do { if (sa.dbg && tid == 0) sa.dbg[(size_t)g * 16 + (i)] = wall_clock64(); } while (0)

template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_step2(KArgs a, Step2Args sa)
{
    __shared__ int s_red2[2][BLOCK / WAVE];
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ float s_lutL[256];
    __shared__ double s_lutI[256];
    constexpr bool U8 = sizeof(FT) == 1;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, g = blockIdx.x;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const bool has_c0 = sa.frame0 != nullptr, has_c1 = sa.frame1 != nullptr;

    __builtin_amdgcn_s_setprio(3);
    const uint32_t fbase_pending = sload_u32_issue(sa.fidx_base);
    V2E_STAMP_2(0);
    // ------------------------------------------------------------ every load of the launch, back to back
    R b = (R)0, lp = (R)0;
    float thp = 1.f, thn = 1.f, nr = 0.f, tsm = 0.f;
    uint32_t cw_e1 = 0;
    FT px0 = (FT)0, px1 = (FT)0;
    if (valid) {
        b = ((R *)a.base)[sp];
        thp = a.pos_thres[sp];
        thn = a.neg_thres[sp];
        if (a.has_cutoff || a.do_shot) lp = ((R *)a.lp)[sp];
        if (a.do_leak) nr = a.noise_rate[sp];
        if (a.has_refr && sa.has_e1) tsm = a.ts_mem[sp];
        if (sa.has_e1) cw_e1 = sa.cnt_e1[sp];
        if (has_c0) px0 = ((const FT *)sa.frame0)[(size_t)clip * a.npx + p];
        if (has_c1) px1 = ((const FT *)sa.frame1)[(size_t)clip * a.npx + p];
    }
    float lutL_r = 0.f;
    double lutI_r = 0.0;
    if (U8 && has_c0) {
        lutL_r = a.lut_L[tid];
        lutI_r = a.lut_I[tid];
    }
    int g1a = 0, g1b = 0, g1c = 0, g1d = 0, g2a = 0, g2b = 0, g2c = 0, g2d = 0; // maxima of e1 / e2: four per thread cover 1024 workgroups
    const size_t sg = (size_t)clip * sa.ngroups;
    if (sa.has_e1) {
        const int *gmv = sa.gmax_e1 + sg;
        if (tid < sa.ngroups) g1a = gmv[tid];
        if (tid + BLOCK < sa.ngroups) g1b = gmv[tid + BLOCK];
        if (tid + 2 * BLOCK < sa.ngroups) g1c = gmv[tid + 2 * BLOCK];
        if (tid + 3 * BLOCK < sa.ngroups) g1d = gmv[tid + 3 * BLOCK];
    }
    const bool check_e2 = sa.has_e2 && a.has_refr;
    if (check_e2) {
        const int *gmv = sa.gmax_e2 + sg;
        if (tid < sa.ngroups) g2a = gmv[tid];
        if (tid + BLOCK < sa.ngroups) g2b = gmv[tid + BLOCK];
        if (tid + 2 * BLOCK < sa.ngroups) g2c = gmv[tid + 2 * BLOCK];
        if (tid + 3 * BLOCK < sa.ngroups) g2d = gmv[tid + 3 * BLOCK];
    }
    const FrameCtl *ce1 = sa.ctl_e1 + clip, *ce2 = sa.ctl_e2 + clip;
    const FrameTab ft1(a.has_refr && sa.has_e1 ? ce1 : sa.ctl_c0 + clip, lane);
    const FrameTab ft2(check_e2 ? ce2 : sa.ctl_c0 + clip, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- arithmetic that depends on no memory but fbase, done while those loads are in flight
    const uint32_t fbase = sload_wait(fbase_pending);
    float r0 = 0.f, u0 = 0.f, r1 = 0.f, u1 = 0.f;
    const bool need_r = a.do_leak && a.jit_f != 0.f; // a zero jitter fraction multiplies the normal away (emulator_utils.py:126)
    if (valid && (need_r || a.do_shot)) {
        const uint32_t f0 = fbase + sa.fidx_c0;
        if (has_c0 && has_c1 && v2e_frame_half(f0) == 0u) { // c0, c1 are the two frames of one pair: one Philox call
            v2e_draw_pair(a.seed, (uint32_t)clip, v2e_frame_pair(f0), (uint32_t)p, need_r, &r0, &u0, &r1, &u1);
        } else {
            if (has_c0) v2e_draw_frame(a.seed, (uint32_t)clip, f0, (uint32_t)p, &r0, &u0);
            if (has_c1) v2e_draw_frame(a.seed, (uint32_t)clip, fbase + sa.fidx_c1, (uint32_t)p, &r1, &u1);
        }
    }
    V2E_STAMP_2(5);
    if (U8 && has_c0) { // published by the barriers of block_max2_finish
        s_lutL[tid] = lutL_r;
        s_lutI[tid] = lutI_r;
    }
    int M1 = max(max(g1a, g1b), max(g1c, g1d)), M2 = max(max(g2a, g2b), max(g2c, g2d));
    if (sa.ngroups > 4 * BLOCK) {
        if (sa.has_e1) for (int k = tid + 4 * BLOCK; k < sa.ngroups; k += BLOCK) M1 = max(M1, sa.gmax_e1[sg + k]);
        if (check_e2) for (int k = tid + 4 * BLOCK; k < sa.ngroups; k += BLOCK) M2 = max(M2, sa.gmax_e2[sg + k]);
    }
    block_max2_finish(M1, M2, s_red2, lane, wave);
    V2E_STAMP_2(1);
    bool b_dirty = false;

    // ------------------------------------------------------------ was the speculation on e2 right?
    if (check_e2 && M2 <= a.max_iters) {
        bool use_refr2;
        const TsGen tg2 = frame_tsgen(a, ce2, ft2, M2 > 0 ? M2 : 1, use_refr2);
        if (use_refr2) { // no: finalise e2 exactly from its checkpoint, re-count e1, learn the corrected M(e1)
            uint32_t cw2 = 0;
            R lp2 = (R)0;
            if (valid) {
                b = ((const R *)sa.bck_e2)[sp];
                cw2 = sa.cnt_e2[sp];
                if (cw2 & (CNT_SHOT_ON | CNT_SHOT_OFF)) lp2 = ((const R *)sa.lpn_e2)[sp]; // only shot pixels reset to lp
            }
            finalize_px<R>(a, cw2, true, tg2, valid, sp, thp, thn, lp2, sa.tsold_e2, b, tsm, b_dirty);
            int m = 0;
            if (valid) {
                if (a.do_leak) { // the leak step of e1 again (emulator_utils.py:126-129), same draw
                    float r = 0.f, u = 0.f;
                    v2e_draw_frame(a.seed, (uint32_t)clip, fbase + sa.fidx_e1, (uint32_t)p, &r, &u);
                    const FrameCtl c = *ce1;
                    const double delta_time = c.t_frame - c.t_prev;
                    const float rate = (a.leak_hz_f * nr) * (1.0f - a.jit_f * r);
                    const float delta_leak = ((float)delta_time * rate) * thp;
                    b = b - (R)delta_leak;
                }
                b_dirty = true;
                const uint32_t shot_old = cw_e1 & (CNT_SHOT_ON | CNT_SHOT_OFF); // shot bits do not depend on base
                cw_e1 = count_px<R>(a, lp, b, thp, thn, m) | shot_old;
                sa.cnt_e1[sp] = cw_e1;
            }
            m = wave_max_i32(m);
            if (lane == 0) s_red[wave] = m;
            __syncthreads();
            if (tid == 0) sa.gmax_e1[sg + g] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
            const bool ok = clip_barrier(sa.bar + clip, (unsigned)sa.ngroups);
            if (!ok && tid == 0) atomicOr(&sa.rec_e1[clip].flags, V2E_FLAG_SYNC_TIMEOUT);
            M1 = block_max_of_groups(sa.gmax_e1 + sg, sa.ngroups, s_red, tid, lane, wave);
        }
    }

    // ------------------------------------------------------------ exact finalise of e1
    if (sa.has_e1 && M1 <= a.max_iters) {
        bool use_refr1 = false;
        TsGen tg1(0.f, 0.f, 0.f, 1);
        if (a.has_refr) tg1 = frame_tsgen(a, ce1, ft1, M1 > 0 ? M1 : 1, use_refr1);
        finalize_px<R>(a, cw_e1, use_refr1, tg1, valid, sp, thp, thn, lp, sa.tsold_e1, b, tsm, b_dirty);
    }
    V2E_STAMP_2(2);

    // ------------------------------------------------------------ count c0, speculative finalise, count c1
    int m0 = 0, m1 = 0;
    if (has_c0) {
        if (valid) {
            double inten01;
            const FrameCtl c = sa.ctl_c0[clip];
            const R lpn = photoreceptor_px<R, FT>(a, px0, s_lutL, s_lutI, c, lp, nr, thp, r0, inten01, b, b_dirty);
            uint32_t cw = count_px<R>(a, lpn, b, thp, thn, m0);
            if (a.do_shot) cw |= shot_bits(a, inten01, c.shot_base, thp, thn, u0);
            sa.cnt_c0[sp] = cw;
            lp = lpn;
            if (has_c1) {
                if (a.has_refr) { // checkpoint for the next launch's validation (lp only where a shot event resets base to it)
                    ((R *)sa.bck_c0)[sp] = b;
                    if (cw & (CNT_SHOT_ON | CNT_SHOT_OFF)) ((R *)sa.lpn_c0)[sp] = lpn;
                }
                const TsGen none(0.f, 0.f, 0.f, 1);
                finalize_px<R>(a, cw, false, none, valid, sp, thp, thn, lpn, nullptr, b, tsm, b_dirty); // speculation: rule off
                const FrameCtl c1 = sa.ctl_c1[clip];
                const R lpn1 = photoreceptor_px<R, FT>(a, px1, s_lutL, s_lutI, c1, lp, nr, thp, r1, inten01, b, b_dirty);
                uint32_t cw1 = count_px<R>(a, lpn1, b, thp, thn, m1);
                if (a.do_shot) cw1 |= shot_bits(a, inten01, c1.shot_base, thp, thn, u1);
                sa.cnt_c1[sp] = cw1;
                lp = lpn1;
            }
            ((R *)a.lp)[sp] = lp;
        }
        V2E_STAMP_2(3);
        m0 = wave_max_i32(m0);
        m1 = wave_max_i32(m1);
        if (lane == 0) { s_red2[0][wave] = m0; s_red2[1][wave] = m1; }
        __syncthreads();
        if (tid == 0) {
            sa.gmax_c0[sg + g] = max(max(s_red2[0][0], s_red2[0][1]), max(s_red2[0][2], s_red2[0][3]));
            if (has_c1) sa.gmax_c1[sg + g] = max(max(s_red2[1][0], s_red2[1][1]), max(s_red2[1][2], s_red2[1][3]));
        }
    }
    if (valid && b_dirty) ((R *)a.base)[sp] = b;
    V2E_STAMP_2(4);
}

__global__ __launch_bounds__(BLOCK) void k_tot_multi(KArgs a, EmitArgs ea)
{
    __shared__ uint32_t s_wcnt[BLOCK / WAVE][WAVE];
    __shared__ int s_red[BLOCK / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, g = blockIdx.x, fe = ea.f0 + (int)blockIdx.z, slot = fe % ea.D;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = ((size_t)slot * ea.n_clips + clip) * a.npx_pad + p;
    const size_t sg = ((size_t)slot * ea.n_clips + clip) * ea.ngroups;
    const uint32_t cw = valid ? ea.cnt[sp] : 0u;
    const float tsm = (valid && ea.tsold) ? ea.tsold[sp] : 0.f; // meaningful only on refractory frames
    const int *gmv = ea.gmax + sg;
    int gm_part = 0;
    for (int k = tid; k < ea.ngroups; k += BLOCK) gm_part = max(gm_part, gmv[k]);
    const int gown = __builtin_amdgcn_readfirstlane(gmv[g]);
    const int gmax_old = __builtin_amdgcn_readfirstlane(ea.rowext[sg + g]);
    const FrameCtl *c = ea.ctl + (size_t)fe * ea.n_clips + clip;
    const FrameTab ft(c, lane);
    const int M = block_max_finish(gm_part, s_red, lane, wave);
    if (M > a.max_iters) { // frame is flagged by k_emit_multi; rows stay as they are
        if (tid == 0) ea.nw[sg + g] = 0u;
        return;
    }
    bool use_refr;
    const TsGen tg = frame_tsgen(a, c, ft, M > 0 ? M : 1, use_refr);
    const int gm = min(gown, a.max_iters);
    uint16_t *gcol = ea.gtT + ((size_t)slot * ea.n_clips + clip) * a.nkeys_cap * ea.ngp + g;
    uint32_t acc = 0;
    if (use_refr) group_key_totals<true>(a, cw, tsm, tg, gm, gcol, ea.ngp, s_wcnt, lane, wave, gmax_old, &acc);
    else group_key_totals<false>(a, cw, 0.f, tg, gm, gcol, ea.ngp, s_wcnt, lane, wave, gmax_old, &acc);
    if (wave == 0) {
        const uint32_t tot = wave_sum_u32(acc);
        if (lane == 0) {
            ea.nw[sg + g] = tot;
            ea.rowext[sg + g] = gm;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_scan2_multi(KArgs a, EmitArgs ea)
{
    const int fe = ea.f0 + (int)blockIdx.z, slot = fe % ea.D;
    scan2_body(a, ea.gtT + (size_t)slot * ea.n_clips * a.nkeys_cap * ea.ngp, ea.ngp, ea.gmax + (size_t)slot * ea.n_clips * ea.ngroups,
               ea.ngroups, ea.pre32 + (size_t)blockIdx.z * ea.n_clips * a.nkeys_cap * ea.ngp,
               ea.tot32 + (size_t)blockIdx.z * ea.n_clips * a.nkeys_cap);
}

// the iteration-by-iteration event writer for workgroup g of batch frame z (any M, any grid size)
__device__ __forceinline__ void emit_frame_iterwise(const KArgs &a, const EmitArgs &ea, const int g, const int z)
{
    __shared__ uint32_t s_T[WAVE], s_P[WAVE]; // per key of the current 64-key chunk: total / prefix over workgroups
    __shared__ uint32_t s_wcnt[BLOCK / WAVE][WAVE];
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ unsigned long long s_off[BLOCK / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, fe = ea.f0 + z, slot = fe % ea.D;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = ((size_t)slot * ea.n_clips + clip) * a.npx_pad + p;
    const size_t sg = ((size_t)slot * ea.n_clips + clip) * ea.ngroups;
    const uint32_t fbase = ea.fidx_base ? *ea.fidx_base : 0u;
    const uint32_t frame_idx = fbase + (uint32_t)fe;
    if (ea.ftab && !ea.ftab[(size_t)z * ea.n_clips + clip].big) return; // k_emit2_multi's frame

    // ------------------------------------------------------------ loads
    const uint32_t cw = valid ? ea.cnt[sp] : 0u;
    float tsm = (valid && ea.tsold) ? ea.tsold[sp] : 0.f;
    const uint16_t *gt = ea.gtT + ((size_t)slot * ea.n_clips + clip) * a.nkeys_cap * ea.ngp;
    const uint32_t *pre32 = ea.pre32 ? ea.pre32 + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap * ea.ngp : nullptr;
    const uint32_t *tot32 = ea.tot32 ? ea.tot32 + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap : nullptr;
    const bool krow_fast = !pre32 && ea.ngp == 512;
    uint4 kv[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) kv[j] = make_uint4(0u, 0u, 0u, 0u);
    if (krow_fast) {
#pragma unroll
        for (int j = 0; j < KPW; ++j) kv[j] = *(const uint4 *)(gt + (size_t)(wave + (BLOCK / WAVE) * j) * 512 + lane * 8);
    }
    int gm_part = 0;
    {
        const int *gmv = ea.gmax + sg;
        for (int k = tid; k < ea.ngroups; k += BLOCK) gm_part = max(gm_part, gmv[k]);
    }
    unsigned long long npart = 0; // events of the batch's earlier frames: this frame's offset within the batch
    for (int j = 0; j < z; ++j) {
        const uint32_t *nwj = ea.nw + ((size_t)((ea.f0 + j) % ea.D) * ea.n_clips + clip) * ea.ngroups;
        for (int k = tid; k < ea.ngroups; k += BLOCK) npart += nwj[k];
    }
    const unsigned long long off_in = ea.off_in[clip];
    const FrameCtl *c = ea.ctl + (size_t)fe * ea.n_clips + clip;
    const FrameTab ft(c, lane);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t pk[4] = {0, 0, 0, 0}; // shuffle round keys of iteration `lane` (first chunk)
    const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
    if (shuf) v2e_perm_keys(a.seed, (uint32_t)clip, frame_idx, (uint32_t)lane, pk);
    if (pre32) {
        if (wave == 0) {
            s_T[lane] = tot32[lane];
            s_P[lane] = pre32[(size_t)lane * ea.ngp + g];
        }
    } else if (krow_fast) {
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            uint32_t t, q;
            key_totals_loaded(kv[j], g, lane, t, q);
            if (lane == 0) { s_T[wave + (BLOCK / WAVE) * j] = t; s_P[wave + (BLOCK / WAVE) * j] = q; }
        }
    } else {
        for (int k = wave; k < KPRE && k < a.nkeys_cap; k += BLOCK / WAVE) {
            uint32_t t, q;
            key_totals(gt + (size_t)k * ea.ngp, ea.ngp, g, lane, t, q);
            if (lane == 0) { s_T[k] = t; s_P[k] = q; }
        }
    }
    {
        const uint32_t lo = wave_sum_u32((uint32_t)(npart & 0xFFFFFFull));
        const uint32_t hi = wave_sum_u32((uint32_t)(npart >> 24));
        if (lane == 0) s_off[wave] = (unsigned long long)lo + ((unsigned long long)hi << 24);
    }
    const int M = block_max_finish(gm_part, s_red, lane, wave);
    const unsigned long long ev0 = off_in + s_off[0] + s_off[1] + s_off[2] + s_off[3];

    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips;
    if (M > a.max_iters) {
        if (g == 0 && tid == 0) {
            rec[clip].max_events = M;
            rec[clip].flags |= V2E_FLAG_ITERS_CLAMPED;
            rec[clip].ev_offset = ev0;
            if (z == ea.nE - 1) ea.off_out[clip] = ev0;
        }
        return;
    }
    const int n = M > 0 ? M : 1;
    bool use_refr;
    const TsGen tg = frame_tsgen(a, c, ft, n, use_refr);
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    float4 *ev = ea.events + (size_t)clip * ea.cap;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float fx = (float)(p % a.W), fy = (float)(p / a.W);
    const int nk = 2 + 2 * M;
    uint32_t carry = 0, sum_on = 0, sum_off = 0;
    uint32_t son_tot = 0, soff_tot = 0, son_off = 0, soff_off = 0;
    bool dropped = false, alive = true;
    for (int kb = 0; kb < nk; kb += WAVE) {
        const int key = kb + lane;
        // totals over all workgroups / over earlier workgroups for the keys not fetched yet
        if (pre32) {
            if (kb > 0 && wave == 0 && key < nk) {
                s_T[lane] = tot32[key];
                s_P[lane] = pre32[(size_t)key * ea.ngp + g];
            }
        } else {
            for (int k = (kb == 0 ? KPRE : 0) + wave; k < WAVE && kb + k < nk; k += BLOCK / WAVE) {
                uint32_t t, q;
                key_totals(gt + (size_t)(kb + k) * ea.ngp, ea.ngp, g, lane, t, q);
                if (lane == 0) { s_T[k] = t; s_P[k] = q; }
            }
        }
        // pass 1: which of my iterations survive; per-wave key counts
        uint32_t mymask = 0, mine = 0;
        const int i_lo = kb == 0 ? 0 : (kb - 2) / 2;
        const int i_hi = (kb + WAVE - 2) / 2;
        for (int i = i_lo; i < i_hi && i < M && alive; ++i) {
            const bool cand = mag > i;
            if (__ballot(cand) == 0ull) { alive = false; break; }
            bool pass = cand;
            if (use_refr) {
                const float t = tg(i);
                const float pt = (cand ? 1.0f : 0.0f) * t - tsm;
                pass = pt > a.refr_f;
                if (pass) tsm = t;
            }
            if (pass) mymask |= 1u << (i - i_lo);
            const unsigned long long bo = __ballot(pass && !neg);
            const unsigned long long bf = __ballot(pass && neg);
            const int kl = 2 + 2 * i - kb;
            if (lane == kl) mine = (uint32_t)__popcll(bo);
            if (lane == kl + 1) mine = (uint32_t)__popcll(bf);
        }
        if (kb == 0) {
            const unsigned long long so = __ballot((cw & CNT_SHOT_ON) != 0);
            const unsigned long long sf = __ballot((cw & CNT_SHOT_OFF) != 0);
            if (lane == 0) mine = (uint32_t)__popcll(so);
            if (lane == 1) mine = (uint32_t)__popcll(sf);
        }
        s_wcnt[wave][lane] = mine;
        __syncthreads();
        const uint32_t T_k = key < nk ? s_T[lane] : 0u;
        const uint32_t P_k = key < nk ? s_P[lane] : 0u;
        uint32_t woff = 0;
#pragma unroll
        for (int q = 0; q < BLOCK / WAVE; ++q)
            if (q < wave) woff += s_wcnt[q][lane];
        const uint32_t off_k = P_k + woff;
        // shuffle domain of iteration `lane` (first chunk), all iterations at once
        uint32_t ps_sh = 0, ps_a = 1, ps_amask = 0, ps_n = 0;
        if (shuf && kb == 0 && lane < 31 && 3 + 2 * lane < nk) {
            ps_n = s_T[2 + 2 * lane] + s_T[3 + 2 * lane];
            v2e_perm_shape(ps_n, &ps_sh, &ps_a, &ps_amask);
        }
        const uint32_t sig_T = (key >= 2 && key < nk) ? T_k : 0u;
        const uint32_t kbase_k = carry + wave_excl_scan_u32(sig_T, lane);
        const uint32_t chunk_total = wave_sum_u32(sig_T);
        sum_on += wave_sum_u32((lane & 1) ? 0u : sig_T);
        sum_off += wave_sum_u32((lane & 1) ? sig_T : 0u);
        if (kb == 0) {
            son_tot = lane_value(T_k, 0); soff_tot = lane_value(T_k, 1);
            son_off = lane_value(off_k, 0); soff_off = lane_value(off_k, 1);
        }
        // pass 2: write this chunk's events
        if (__ballot(mymask != 0u)) {
            uint32_t wm = wave_or_u32(mymask); // iterations in which some lane of the wave fires
            while (wm) {
                const int ii = __ffs(wm) - 1;
                wm &= wm - 1;
                const int i = i_lo + ii;
                const bool pass = (mymask >> ii) & 1u;
                const unsigned long long bo = __ballot(pass && !neg);
                const unsigned long long bf = __ballot(pass && neg);
                const int kl = 2 + 2 * i - kb;
                const uint32_t it_base = lane_value(kbase_k, kl);
                const uint32_t tot_on = lane_value(T_k, kl);
                const uint32_t tot_off = lane_value(T_k, kl + 1);
                const uint32_t off_on = lane_value(off_k, kl);
                const uint32_t off_off = lane_value(off_k, kl + 1);
                v2e_perm_t pm;
                if (shuf) {
                    if (kb == 0) { // keys / domain computed lane-parallel above: fetch as scalars
                        pm.k[0] = lane_value(pk[0], ii); pm.k[1] = lane_value(pk[1], ii);
                        pm.k[2] = lane_value(pk[2], ii); pm.k[3] = lane_value(pk[3], ii);
                        pm.sh = lane_value(ps_sh, ii); pm.a = lane_value(ps_a, ii);
                        pm.amask = lane_value(ps_amask, ii); pm.n = lane_value(ps_n, ii);
                        pm.rmask = (1u << pm.sh) - 1u;
                    } else {
                        v2e_perm_init(&pm, a.seed, (uint32_t)clip, frame_idx, (uint32_t)i, tot_on + tot_off);
                    }
                }
                if (pass) {
                    uint32_t cidx = neg ? tot_on + off_off + (uint32_t)__popcll(bf & lt) : off_on + (uint32_t)__popcll(bo & lt);
                    if (shuf) cidx = v2e_perm_apply(&pm, cidx);
                    const unsigned long long row = ev0 + it_base + cidx;
                    if (row < ea.cap) ev[row] = make_float4(tg(i), fx, fy, neg ? -1.0f : 1.0f);
                    else dropped = true;
                }
            }
        }
        carry += chunk_total;
        __syncthreads();
    }
    // shot-noise events after all signal events (ON block, OFF block), ts[-1], unshuffled
    if (a.do_shot) {
        const bool s_on = (cw & CNT_SHOT_ON) != 0, s_off = (cw & CNT_SHOT_OFF) != 0;
        const unsigned long long so = __ballot(s_on), sf = __ballot(s_off);
        if (so | sf) {
            const float tl = tg(n - 1);
            if (s_on) {
                const unsigned long long row = ev0 + carry + son_off + (uint32_t)__popcll(so & lt);
                if (row < ea.cap) ev[row] = make_float4(tl, fx, fy, 1.0f);
                else dropped = true;
            }
            if (s_off) {
                const unsigned long long row = ev0 + carry + son_tot + soff_off + (uint32_t)__popcll(sf & lt);
                if (row < ea.cap) ev[row] = make_float4(tl, fx, fy, -1.0f);
                else dropped = true;
            }
        }
    }
    if (__ballot(dropped) != 0ull && lane == 0) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
    if (g == 0 && tid == 0) {
        const uint32_t n_events = carry + (a.do_shot ? son_tot + soff_tot : 0u);
        rec[clip].max_events = M;
        rec[clip].n_signal = carry;
        rec[clip].n_events = n_events;
        rec[clip].n_on = sum_on + (a.do_shot ? son_tot : 0u);
        rec[clip].n_off = sum_off + (a.do_shot ? soff_tot : 0u);
        rec[clip].ev_offset = ev0;
        if (z == ea.nE - 1) ea.off_out[clip] = ev0 + n_events;
    }
}

__global__ __launch_bounds__(BLOCK) void k_emit_multi(KArgs a, EmitArgs ea)
{
    emit_frame_iterwise(a, ea, (int)blockIdx.x, (int)blockIdx.z);
}

// With per-frame tables only frames with M > 31 are left for the iteration-wise writer: a small grid walks them
// (none, on almost every batch) instead of a full grid launching just to find that out.
__global__ __launch_bounds__(BLOCK) void k_emit_big(KArgs a, EmitArgs ea)
{
    for (int z = 0; z < ea.nE; ++z) {
        if (!ea.ftab[(size_t)z * ea.n_clips + blockIdx.y].big) continue;
        for (int g = (int)blockIdx.x; g < ea.ngroups; g += (int)gridDim.x) {
            emit_frame_iterwise(a, ea, g, z);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ per-frame tables (small grids)
// One workgroup per (frame, clip): everything about a frame's event list that is the same for all of its
// workgroups -- M, the per-key totals and their prefix over keys, the prefix over workgroups of every key row,
// the shuffle parameters of every iteration -- computed once instead of by each of the frame's 1408 waves.
__global__ __launch_bounds__(BLOCK) void k_frame_multi(KArgs a, EmitArgs ea)
{
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ uint32_t s_T[FT_KEYS];
    __shared__ uint32_t s_nw[BLOCK / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, z = blockIdx.z, fe = ea.f0 + z, slot = fe % ea.D;
    const size_t sg = ((size_t)slot * ea.n_clips + clip) * ea.ngroups;
    int gm_part = 0;
    uint32_t nsum = 0;
    for (int k = tid; k < ea.ngroups; k += BLOCK) {
        gm_part = max(gm_part, ea.gmax[sg + k]);
        nsum += ea.nw[sg + k];
    }
    nsum = wave_sum_u32(nsum);
    if (lane == 0) s_nw[wave] = nsum;
    const int M = block_max_finish(gm_part, s_red, lane, wave);
    const uint32_t N = s_nw[0] + s_nw[1] + s_nw[2] + s_nw[3];
    FrameTable *ft = ea.ftab + (size_t)z * ea.n_clips + clip;
    const bool big = M > 31 || M > a.max_iters;
    if (tid == 0) {
        ft->M = M;
        ft->n_events = M > a.max_iters ? 0u : N;
        ft->big = big ? 1u : 0u;
    }
    if (big) return;
    const int nk = 2 + 2 * M; // <= 64
    const uint16_t *gt = ea.gtT + ((size_t)slot * ea.n_clips + clip) * a.nkeys_cap * 512;
    uint32_t *pre = ea.pre512 + ((size_t)z * ea.n_clips + clip) * FT_KEYS * 512;
    for (int k = wave; k < nk; k += BLOCK / WAVE) { // one wave per key row: 512 workgroups, 8 per lane
        const uint4 v = *(const uint4 *)(gt + (size_t)k * 512 + lane * 8);
        const uint32_t w[8] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16, v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
        const uint32_t lane_tot = w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        uint32_t run = wave_excl_scan_u32(lane_tot, lane);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)(run + lane_tot), WAVE - 1);
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = run; run += w[j]; }
        uint4 *dst = (uint4 *)(pre + (size_t)k * 512 + lane * 8);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        if (lane == 0) s_T[k] = tot;
    }
    __syncthreads();
    if (wave == 0) {
        const int key = lane;
        const uint32_t T_k = key < nk ? s_T[key] : 0u;
        const uint32_t sig_T = key >= 2 ? T_k : 0u;
        const uint32_t kbase = wave_excl_scan_u32(sig_T, lane);
        const uint32_t n_signal = wave_sum_u32(sig_T);
        const uint32_t sum_on = wave_sum_u32((lane & 1) ? 0u : sig_T), sum_off = wave_sum_u32((lane & 1) ? sig_T : 0u);
        ft->T[lane] = T_k;
        ft->kbase[lane] = kbase;
        if (lane < 32) {
            uint32_t pk[4] = {0, 0, 0, 0}, ps_sh = 0, ps_a = 1, ps_amask = 0, ps_n = 0;
            if (a.shuffle && a.rng_mode == V2E_RNG_PHILOX && lane < 31 && 3 + 2 * lane < nk) {
                const uint32_t fbase = ea.fidx_base ? *ea.fidx_base : 0u;
                ps_n = s_T[2 + 2 * lane] + s_T[3 + 2 * lane];
                v2e_perm_shape(ps_n, &ps_sh, &ps_a, &ps_amask);
                v2e_perm_keys(a.seed, (uint32_t)clip, fbase + (uint32_t)fe, (uint32_t)lane, pk);
            }
            uint4 *pp = (uint4 *)ft->perm[lane];
            pp[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            pp[1] = make_uint4(ps_sh, ps_a, ps_amask, ps_n);
        }
        if (lane == 0) {
            v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips + clip;
            const uint32_t son = a.do_shot ? s_T[0] : 0u, soff = a.do_shot ? s_T[1] : 0u;
            rec->max_events = M;
            rec->n_signal = n_signal;
            rec->n_events = n_signal + son + soff;
            rec->n_on = sum_on + son;
            rec->n_off = sum_off + soff;
        }
    }
}

// Emission for frames with a table (M <= 31, one 64-key chunk).  Pass 1 is pixel-parallel as before (which of
// my iterations pass; ballots give the rank among the wave's same-polarity events of the iteration) but instead
// of writing events iteration by iteration -- a wave-uniform loop whose shuffle arithmetic runs for the few
// lanes that fire -- every passing (lane, iteration) leaves a 4-byte record in LDS, and the wave then walks its
// own records with all lanes busy: one event per lane, parameters gathered by lane index (ds_bpermute).
// Event rows written through to memory (system-scope buffer store): they are final output that nothing on the
// device reads back, and as dirty L2 lines they would be written back by the release at the end of every chain
// launch that happens to run meanwhile.
typedef float v2e_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_event_wt(float4 *ev_clip, unsigned long long row, float t, float x, float y, float pol)
{
    if (WT_EVENTS && row < 0x7000000ull) { // byte offset below 2^31
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)ev_clip, 0, 0x7fffffff, 0x00020000);
        const v2e_f4 v = {t, x, y, pol};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(row * 16ull), 0, 17); // sc0 sc1
    } else {
        ev_clip[row] = make_float4(t, x, y, pol);
    }
}

__global__ __launch_bounds__(BLOCK) void k_emit2_multi(KArgs a, EmitArgs ea)
{
    extern __shared__ uint32_t s_rec[]; // [BLOCK / WAVE][capw]
    __shared__ uint32_t s_wcnt[BLOCK / WAVE][WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, g = blockIdx.x, z = blockIdx.z, fe = ea.f0 + z, slot = fe % ea.D;
    const FrameTable *ft = ea.ftab + (size_t)z * ea.n_clips + clip;
    if (ft->big) return; // k_emit_multi's frame
    const int M = __builtin_amdgcn_readfirstlane(ft->M);
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = ((size_t)slot * ea.n_clips + clip) * a.npx_pad + p;

    // ------------------------------------------------------------ loads
    const uint32_t cw = valid ? ea.cnt[sp] : 0u;
    float tsm = (valid && ea.tsold) ? ea.tsold[sp] : 0.f;
    const int nk = 2 + 2 * M;
    const int key = lane;
    const uint32_t T_k = ft->T[key], kbase_k = ft->kbase[key]; // zero / total beyond the frame's keys
    uint32_t P_k = 0;
    if (key < nk) P_k = ea.pre512[(((size_t)z * ea.n_clips + clip) * FT_KEYS + key) * 512 + g];
    const uint4 pa = lane < 32 ? ((const uint4 *)ft->perm[lane])[0] : make_uint4(0u, 0u, 0u, 0u);
    const uint4 pb = lane < 32 ? ((const uint4 *)ft->perm[lane])[1] : make_uint4(0u, 1u, 0u, 0u);
    unsigned long long ev0 = ea.off_in[clip];
    for (int j = 0; j < z; ++j) ev0 += ea.ftab[(size_t)j * ea.n_clips + clip].n_events;
    const FrameCtl *c = ea.ctl + (size_t)fe * ea.n_clips + clip;
    const FrameTab ftb(c, lane);
    const int n = M > 0 ? M : 1;
    bool use_refr;
    const TsGen tg = frame_tsgen(a, c, ftb, n, use_refr);
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    float4 *ev = ea.events + (size_t)clip * ea.cap;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float fx = (float)(p % a.W), fy = (float)(p / a.W);
    const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
    uint32_t *rec_w = s_rec + (size_t)wave * ea.capw;

    // ------------------------------------------------------------ pass 1: records + per-wave key counts
    uint32_t mine = 0, nrec = 0;
    for (int i = 0; i < M; ++i) {
        const bool cand = mag > i;
        if (__ballot(cand) == 0ull) break;
        bool pass = cand;
        if (use_refr) {
            const float t = tg(i);
            const float pt = (cand ? 1.0f : 0.0f) * t - tsm;
            pass = pt > a.refr_f;
            if (pass) tsm = t;
        }
        const unsigned long long bo = __ballot(pass && !neg);
        const unsigned long long bf = __ballot(pass && neg);
        const int kl = 2 + 2 * i;
        if (lane == kl) mine = (uint32_t)__popcll(bo);
        if (lane == kl + 1) mine = (uint32_t)__popcll(bf);
        if (pass) {
            const uint32_t rank = (uint32_t)__popcll((neg ? bf : bo) & lt);
            const uint32_t pos = nrec + (uint32_t)__popcll((bo | bf) & lt);
            if (pos < (uint32_t)ea.capw) rec_w[pos] = (uint32_t)lane | ((uint32_t)i << 6) | ((neg ? 1u : 0u) << 11) | (rank << 12);
        }
        nrec += (uint32_t)__popcll(bo | bf);
    }
    {
        const unsigned long long so = __ballot((cw & CNT_SHOT_ON) != 0);
        const unsigned long long sf = __ballot((cw & CNT_SHOT_OFF) != 0);
        if (lane == 0) mine = (uint32_t)__popcll(so);
        if (lane == 1) mine = (uint32_t)__popcll(sf);
    }
    s_wcnt[wave][lane] = mine;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int q = 0; q < BLOCK / WAVE; ++q)
        if (q < wave) woff += s_wcnt[q][lane];
    const uint32_t off_k = P_k + woff;
    const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)(kbase_k + T_k), WAVE - 1); // all signal events of the frame
    bool dropped = nrec > (uint32_t)ea.capw;

    // ------------------------------------------------------------ pass 2: one event per lane
    const uint32_t nrec_c = min(nrec, (uint32_t)ea.capw);
    for (uint32_t e0 = 0; e0 < nrec_c; e0 += WAVE) {
        const bool has = e0 + lane < nrec_c;
        const uint32_t r = has ? rec_w[e0 + lane] : 0u;
        const int src = (int)(r & 63u), i = (int)((r >> 6) & 31u);
        const bool eneg = (r >> 11) & 1u;
        const uint32_t rank = r >> 12;
        const int kl = 2 + 2 * i;
        const uint32_t it_base = (uint32_t)__shfl((int)kbase_k, kl);
        const uint32_t tot_on = (uint32_t)__shfl((int)T_k, kl);
        const uint32_t off = (uint32_t)__shfl((int)off_k, kl + (eneg ? 1 : 0));
        const float ex = __shfl(fx, src), ey = __shfl(fy, src);
        uint32_t cidx = (eneg ? tot_on : 0u) + off + rank;
        if (shuf) {
            v2e_perm_t pm;
            pm.k[0] = (uint32_t)__shfl((int)pa.x, i); pm.k[1] = (uint32_t)__shfl((int)pa.y, i);
            pm.k[2] = (uint32_t)__shfl((int)pa.z, i); pm.k[3] = (uint32_t)__shfl((int)pa.w, i);
            pm.sh = (uint32_t)__shfl((int)pb.x, i); pm.a = (uint32_t)__shfl((int)pb.y, i);
            pm.amask = (uint32_t)__shfl((int)pb.z, i); pm.n = (uint32_t)__shfl((int)pb.w, i);
            pm.rmask = (1u << pm.sh) - 1u;
            if (has) cidx = v2e_perm_apply(&pm, cidx);
        }
        if (has) {
            const unsigned long long row = ev0 + it_base + cidx;
            if (row < ea.cap) store_event_wt(ev, row, tg(i), ex, ey, eneg ? -1.0f : 1.0f);
            else dropped = true;
        }
    }
    // shot-noise events after all signal events (ON block, OFF block), ts[-1], unshuffled
    const uint32_t son_tot = lane_value(T_k, 0), soff_tot = lane_value(T_k, 1);
    if (a.do_shot) {
        const uint32_t son_off = lane_value(off_k, 0), soff_off = lane_value(off_k, 1);
        const bool s_on = (cw & CNT_SHOT_ON) != 0, s_off = (cw & CNT_SHOT_OFF) != 0;
        const unsigned long long so = __ballot(s_on), sf = __ballot(s_off);
        if (so | sf) {
            const float tl = tg(n - 1);
            if (s_on) {
                const unsigned long long row = ev0 + carry + son_off + (uint32_t)__popcll(so & lt);
                if (row < ea.cap) store_event_wt(ev, row, tl, fx, fy, 1.0f);
                else dropped = true;
            }
            if (s_off) {
                const unsigned long long row = ev0 + carry + son_tot + soff_off + (uint32_t)__popcll(sf & lt);
                if (row < ea.cap) store_event_wt(ev, row, tl, fx, fy, -1.0f);
                else dropped = true;
            }
        }
    }
    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips;
    if (__ballot(dropped) != 0ull && lane == 0) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
    if (g == 0 && tid == 0) {
        rec[clip].ev_offset = ev0;
        if (z == ea.nE - 1) ea.off_out[clip] = ev0 + carry + (a.do_shot ? son_tot + soff_tot : 0u);
    }
}
