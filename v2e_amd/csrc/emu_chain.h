// emu_chain.h -- the device-resident run: K frames per chain launch, per-pixel state in registers (included by emu.hip).
//
// The frame-to-frame dependency of the DVS pixel model is per pixel, except for ONE global number per frame: the
// frame's max event count M, and only through the refractory rule (emulator.py:830: the rule is applied iff
// refractory_period_s > delta_time / M).  When the rule is off a pixel's pass count is its own count, so a pixel
// can be advanced through any number of frames without looking at another pixel.  k_chain does exactly that:
//
//   one launch = K consecutive frames.  A thread owns a pixel: it loads base_log_frame / lp_log_frame /
//   timestamp_mem / thresholds / noise rate ONCE, then per frame: lin-log (LDS table) -> IIR low-pass -> leak ->
//   exact floor-division -> shot-noise decision -> count word to the frame's ring slot -> finalise (emulator.py:
//   936-942) ASSUMING the rule is off -> next frame.  State goes back to HBM once per launch: the per-pixel state
//   crosses HBM once per K frames, and a launch boundary (~4 us between dependent launches, longer than two
//   frames of arithmetic) is paid once per K frames.
//
//   The chain is a dependency chain of ~1.4 waves per SIMD at 346x260: every instruction in its frame loop is exposed
//   latency.  So the loop holds ONLY what the pixel state needs; everything the event list needs beyond the count word --
//   the per-wave max and (iteration, polarity) totals -- is recomputed from the count words by k_ctot on the emission
//   stream, at full occupancy (round 2 had it in the chain: half of its instruction stream).
//
//   Speculation check: a wave with a lane whose count reaches the frame's rule threshold (FrameCtl::refr_on_n, host-
//   computed with the reference's own float64 predicate) publishes its max with an atomicMax into the launch's gM row.
//   The NEXT launch reads that row first.  All zero (almost always): the previous launch was right.  Otherwise the
//   first flagged frame j was a rule-on frame and M(j) = gM[j] is exact (everything before j was right), and the
//   previous launch is REDONE from its own input state (state planes ping-pong between launches, so it is still
//   there) with j finalised by the rule -- ts_mem as it was goes to the frame's tsold slot for the emission side, M to
//   its ruleM slot -- frames after j run under the previous pass's flagged maxima as predictions, published into the
//   next gM row, one grid rendezvous (clip_barrier, co-resident grids only), and the check repeats on the frames after j.
//   When j is the launch's last frame nothing is left to verify and there is no rendezvous: grids too large to be
//   co-resident run with K = 1 and need none at all.
//   lp_log_frame never depends on the speculation; everything is deterministic.
//
//   Clips (independent pixel arrays) beyond what is co-resident are walked by a loop inside the workgroup.
#pragma once

// lock-step frames in redo passes (see k_chain): measured round 4, break-even at best; compiled out by default (the per-frame
// test costs the common path), -DV2E_CHAIN_LOCKSTEP=1 + V2E_AMD_LOCKSTEP=1 bring it back
#ifndef V2E_CHAIN_LOCKSTEP
#define V2E_CHAIN_LOCKSTEP 0
#endif
// dev tool: wall-clock stamps of one launch (ChainArgs::dbg); compiled out unless -DV2E_CHAIN_STAMPS
#ifndef V2E_CHAIN_STAMPS
#define V2E_CHAIN_STAMPS 0
#endif

constexpr int CHAIN_K_MAX = 64; // (a lane per frame of a launch: rule thresholds, maxima, predictions)
constexpr int CFRAME_THREADS = 1024;
constexpr int CHAIN_SUB = 8; // frames whose records are in LDS at a time (4 KB per frame and workgroup)
constexpr int CHAIN_MAX_ITERS = 1024; // k_cframe keeps one total per key in LDS

// exact floor(a/b) for a >= 0, b > 0: equals c10::div_floor_floating (whose fmod / re-divide /
// "+1 if frac > 0.5" steps exist to return exactly this) without the fmod loop.
template <typename R> __device__ __forceinline__ R floor_div_pos(R a, R b)
{
    if (!(b > (R)0) || !(a >= (R)0)) return div_floor<R>(a, b); // generic path keeps every corner case
    if (a < b) return (R)0;
    R q = floor(a / b);
    R r = fma(-q, b, a); // exactly rounded a - q*b: its sign is the true sign
    if (r < (R)0) q -= (R)1;
    else if (r >= b) q += (R)1;
    return q;
}

// Grid-wide rendezvous of the `target` workgroups of one clip (MI355X guide, Guideline 16 hand-off
// in its counter form): every wave drains its stores, one lane does the agent-scope release, the
// relaxed arrive, a relaxed bounded poll, and the agent-scope acquire; __syncthreads() extends it
// to the workgroup.  Requires every workgroup of the grid to be resident (checked by the host).
// light: no release / acquire FENCES around it (an agent-scope release writes this XCD's dirty L2 lines back, an acquire
// invalidates its L2).  Correct where everything the workgroups exchange across the rendezvous is itself written and read
// with agent-scope atomics -- the chain's rule-on maxima are (atomicMax / __hip_atomic_load): each wave has waited for its own
// atomics (s_waitcnt vmcnt(0): they are acknowledged where they are performed) before its workgroup arrives.
__device__ __forceinline__ bool clip_barrier(unsigned *ctr, unsigned target, bool light = false)
{
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!light) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (light) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
            if (++spins > 2000000u) { ok = 0; break; } // bounded: never hang the GPU
        }
        if (!light) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

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

// Event rows written through to memory (system-scope buffer store): they are final output that nothing on the
// device reads back, and as dirty L2 lines they would be written back by the release at the end of every chain
// launch that happens to run meanwhile.
typedef float v2e_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int v2e_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_event_wt(float4 *ev_clip, unsigned long long row, float t, float x, float y, float pol)
{
    if (row < 0x7000000ull) { // byte offset below 2^31
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)ev_clip, 0, 0x7fffffff, 0x00020000);
        const v2e_f4 v = {t, x, y, pol};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(row * 16ull), 0, 17); // sc0 sc1
    } else {
        ev_clip[row] = make_float4(t, x, y, pol);
    }
}

struct ChainArgs {
    const void *const *frames_pp;     // FUSED: *frames_pp = frames of the run (a device variable the run's upload fills: the pointer is not
                                      // baked into a captured graph), frame f at *frames_pp + f * frame_stride (bytes)
    unsigned long long frame_stride;
    const uint32_t *fidx_base;        // FUSED: the run's first frame index
    const FrameCtl *ctl;              // [n_frames][n_clips]
    int f0, nf;                       // this launch advances frames [f0, f0 + nf) of the run (nf = 0: tail launch)
    int pf0, pnf;                     // the previous launch's frames, to be validated (pnf = 0: nothing to validate)
    int D, n_clips, K, ngroups;
    int slot_f0, slot_pf0;            // f0 % D, pf0 % D from the host (the ring slot of frame fs + k is slot(fs) + k, wrapped: K <= D)
    uint32_t *cnt;                    // [D][n_clips][npx_pad] count words, slot = frame % D
    uint32_t *ruleM;                  // [D][n_clips] M of a frame finalised by the refractory rule, 0 otherwise
    float *tsold;                     // [D][n_clips][npx_pad] ts_mem before a rule-on frame's update, or nullptr
    const uint4 *rec;                 // [D][n_clips][npx_pad] k_ahead's per-(frame, pixel) records
    uint32_t *gM_prev, *gM_cur;       // [K + 1][n_clips][K] rule-on maxima: row r = after r redo passes
    unsigned *bar_prev;               // [K][n_clips] rendezvous counters of the redo passes on the previous launch
    const void *base_in, *lp_in;      // state as the previous launch left it
    const float *ts_in;
    void *base_out, *lp_out;          // where this launch leaves it
    float *ts_out;
    const void *base_pin, *lp_pin;    // what the previous launch started from (redo)
    const float *ts_pin;
    void *base_fix, *lp_fix;          // == *_in, written after a redo so that the next launch can redo this one
    float *ts_fix;
    // refractory runs, K > CHAIN_SUB: the state before frames 8, 16, 24 of a pass ([3][n_clips][npx_pad] each), so that a
    // redo restarts at the checkpoint below the first frame to fix instead of at the launch's first frame
    void *ckc_base, *ckc_lp;          // written by this launch's own pass
    float *ckc_ts;
    void *ckp_base, *ckp_lp;          // the previous launch's: read to restart its redo, rewritten by the redo passes
    float *ckp_ts;
    v2e_frame_rec *recs;              // [n_frames][n_clips]
    int store_out;                    // tail launch: state must be copied to *_out even without a redo
    int prio;                         // wave priority of the chain (3: it outranks the emission waves sharing its SIMDs)
    int bar_light;                    // the redo rendezvous without release / acquire fences (the maxima rows are atomics)
    int lockstep;                     // redo passes take the frame(s) right behind a rule-on frame in lock-step (see k_chain)
    unsigned long long *dbg;          // dev tool: [ngroups][16] wall-clock stamps of one launch, or nullptr
    unsigned long long *const *stamp_pp; // *stamp_pp: the run's [launch][2] time stamps (v2e_emu_launch_stamps) or nullptr
    int lidx;                         // this launch's index in the run
};

// Per-frame outputs of the chain are written through to memory: as dirty L2 lines they would all be written back by the
// release at the end of the launch, on the critical path between two dependent launches.
#define WT_STORE(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

#define V2E_STAMP_C(i) do { if (V2E_CHAIN_STAMPS && ca.dbg && tid == 0) ca.dbg[(size_t)g * 16 + (i)] = wall_clock64(); } while (0)

// One frame's state-independent quantities of a pixel as a 16-byte record (layout: see k_ahead):
// lin-log / eps (emulator_utils.py:18-45, 80-96), leak step (:126-129, float32 left to right), shot decisions (:326-349).
template <typename FT>
__device__ __forceinline__ uint4 make_frame_record(const KArgs &a, FT px, const float *s_lutL, const double *s_lutI, double dt_over_tau,
                                                   double shot_base, float dtime_f, float lk, float thp, float ppre, float npre, float rr,
                                                   float uu)
{
    constexpr bool U8 = sizeof(FT) == 1;
    double L, inten01;
    if (U8) {
        L = a.log_input ? (double)px : (double)s_lutL[(int)px];
        inten01 = s_lutI[(int)px];
    } else {
        const double x = (double)px;
        L = a.log_input ? x : (double)lin_log(x);
        inten01 = a.use_inten ? (x + 20.0) / 275.0 : 0.0;
    }
    double eps = 0.0;
    if (a.has_cutoff) {
        eps = inten01 * dt_over_tau;
        if (eps > 1.0) eps = 1.0;
    }
    float dl = 0.f;
    if (a.do_leak) {
        const float rate = lk * (1.0f - a.jit_f * rr);
        dl = (dtime_f * rate) * thp;
    }
    uint32_t sb = 0;
    if (a.do_shot) {
        const double F = shot_base * (a.inten_slope * inten01 + 1);
        if ((double)uu > 1 - F * (double)ppre) sb |= 1u; // ON -> bit 30 of the high word
        if ((double)uu < F * (double)npre) sb |= 2u;     // OFF -> bit 31
    }
    const unsigned long long eb = (unsigned long long)__double_as_longlong(eps);
    uint4 r;
    r.x = (uint32_t)eb;
    r.y = (uint32_t)(eb >> 32) | (sb << 30);
    r.z = __float_as_uint((float)L);
    r.w = __float_as_uint(dl);
    return r;
}

// Everything about a frame that depends on no state, for all frames of a chain launch at once: lin-log of the pixel,
// the low-pass coefficient eps (emulator_utils.py:80-96), the Philox draws, the leak step delta_leak
// (emulator_utils.py:126-129, float32 left to right) and the shot-noise decisions (emulator_utils.py:326-349).  One thread per
// (pixel, frame pair), so the draws of a pair cost one Philox call and the work runs at full occupancy beside the chain
// instead of inside its one-wave-per-SIMD dependency chain.  Result: one 16-byte record per (frame, pixel)
//   .x .y  eps (float64, in [0,1]: top exponent bit and sign bit are free and carry the shot ON / OFF decisions)
//   .z     lin-log value L (float32; the frame itself when it is log-encoded already)
//   .w     delta_leak (float32)
struct AheadArgs {
    const void *const *frames_pp; // *frames_pp = frames of the run (see ChainArgs)
    unsigned long long frame_stride;
    const FrameCtl *ctl;
    const uint32_t *fidx_base;
    int f0, nf, D, n_clips;
    int slot0;  // f0 % D, from the host (a 64-bit remainder per frame and thread is ~60 instructions of a kernel that has 240)
    int ppt;    // frame pairs per thread (grid z = ceil(pairs / ppt))
    uint4 *rec; // [D][n_clips][npx_pad]
};

template <typename FT>
__global__ __launch_bounds__(BLOCK) void k_ahead(KArgs a, AheadArgs aa)
{
    __shared__ float s_lutL[256];
    __shared__ double s_lutI[256];
    constexpr bool U8 = sizeof(FT) == 1;
    const int tid = threadIdx.x;
    if (U8) {
        s_lutL[tid] = a.lut_L[tid];
        s_lutI[tid] = a.lut_I[tid];
    }
    __syncthreads();
    const int clip = blockIdx.y, p = blockIdx.x * BLOCK + tid;
    if (p >= a.npx) return;
    const uint32_t fbase = *aa.fidx_base;
    const char *const frames = (const char *)*aa.frames_pp;
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const bool need_r = a.do_leak && a.jit_f != 0.f;
    const float thp = a.pos_thres[sp], thn = a.neg_thres[sp];
    const float lk = a.do_leak ? a.leak_hz_f * a.noise_rate[sp] : 0.f;
    const float ppre = a.scalar_thres ? a.pos_pre_scalar : a.pos_nom_f / thp; // emulator.py:475-478
    const float npre = a.scalar_thres ? a.neg_pre_scalar : a.neg_nom_f / thn;
    v2e_philox_keys ks; // the seed's ten round keys: scalar registers, computed once per thread (not per call: 35 scalar instructions a pair)
    v2e_philox_key_schedule((uint32_t)a.seed, (uint32_t)(a.seed >> 32), &ks);
    for (int zz = 0; zz < aa.ppt; ++zz) {
        // pair z of the launch: global frames 2q-1 (odd) and 2q (even), q = pair of the launch's first frame + z
        const uint32_t q = v2e_frame_pair(fbase + (uint32_t)aa.f0) + blockIdx.z * (uint32_t)aa.ppt + (uint32_t)zz;
        const int f_odd = (int)(2u * q - 1u - fbase); // run-relative (frame f0 - 1 at the earliest: the pair that holds frame f0)
        const bool in0 = f_odd >= aa.f0 && f_odd < aa.f0 + aa.nf, in1 = f_odd + 1 >= aa.f0 && f_odd + 1 < aa.f0 + aa.nf;
        if (!in0 && !in1) continue;
        float r_odd = 0.f, u_odd = 0.f, r_even = 0.f, u_even = 0.f;
        if (need_r || a.do_shot) v2e_draw_pair_ks(&ks, (uint32_t)clip, q, (uint32_t)p, need_r, &r_odd, &u_odd, &r_even, &u_even);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (!(half ? in1 : in0)) continue;
            const int f = f_odd + half;
            int sl = aa.slot0 + (f - aa.f0); // ring slot f % D (nf <= D)
            if (sl >= aa.D) sl -= aa.D;
            // (32-bit indices: the record ring is below 4 GB -- chain_fused_records -- and a run has fewer than 2^31 / n_clips frames;
            //  64-bit products of uniform values are four scalar instructions each)
            const FrameCtl *c = aa.ctl + (f * aa.n_clips + clip);
            const FT px = ((const FT *)(frames + (size_t)f * aa.frame_stride))[(size_t)clip * a.npx + p];
            const uint4 r = make_frame_record<FT>(a, px, s_lutL, s_lutI, c->dt_over_tau, c->shot_base, (float)(c->t_frame - c->t_prev), lk, thp,
                                                  ppre, npre, half ? r_even : r_odd, half ? u_even : u_odd);
            aa.rec[((uint32_t)sl * (uint32_t)aa.n_clips + (uint32_t)clip) * (uint32_t)a.npx_pad + (uint32_t)p] = r;
        }
    }
}

// exact floor(a/b), a >= 0, b > 0, from a reciprocal computed once per launch: a*rb is within a few ulp of a/b, so
// its floor is the true floor or one off, and the exactly rounded remainder a - q*b decides (same value as
// floor_div_pos, which equals c10::div_floor_floating for these operands).  Anything unusual takes floor_div_pos.
template <typename R> __device__ __forceinline__ R floor_div_rcp(R a, R b, R rb)
{
    // branch-free on the path every lane takes (the frame loop of k_chain is one dependency chain per wave: a divergent
    // branch around two instructions costs more than the instructions): quotient estimate, exact remainder, one-step fix-up
    // by selects; a < b gives q = 0 through the same arithmetic (floor(a rb) is 0 or 1, fixed by the remainder's sign)
    R q = floor(a * rb);
    R r = fma(-q, b, a);
    const bool lo = r < (R)0, hi = r >= b;
    q = lo ? q - (R)1 : (hi ? q + (R)1 : q);
    r = lo ? r + b : (hi ? r - b : r);
    if (__builtin_expect(!(r >= (R)0 && r < b), 0)) return floor_div_pos<R>(a, b); // anything unusual (NaN, b <= 0, a huge)
    return q;
}

// FUSED = false: the frames' records come from k_ahead through LDS (small grids: the chain is one wave per SIMD and every
// instruction in it is latency).  FUSED = true: the chain builds each record itself (large grids: the occupancy hides
// latencies, and the records' 32 B per pixel and frame of extra HBM traffic would be what bounds the run).
// Dynamic LDS (FUSED = false): [CHAIN_SUB][BLOCK] uint4 records.
// ALLON: instantiation for runs with a photoreceptor cutoff, leak, shot noise AND a refractory period (the v2e CLI defaults,
// i.e. the benchmark): the per-frame tests of those run-time switches -- a scalar compare and a branch each, in a loop whose
// every instruction is exposed latency -- are compiled out.  ALLON = false reads the switches from KArgs.
// Five waves per SIMD for the fused instantiations on float64 state and uint8 frames (1280x720, multi-clip runs: throughput-bound,
// 96 VGPRs without a spill; measured 1280x720 noisy 10.25 -> 10.5 Gev/s); the other fused ones would spill a few registers.
template <typename R, typename FT, bool FUSED, bool ALLON = false>
__global__ __launch_bounds__(BLOCK)
__attribute__((amdgpu_waves_per_eu((FUSED && sizeof(R) == 8 && sizeof(FT) == 1) ? 5 : 1, 8)))
void k_chain(KArgs a_in, ChainArgs ca)
{
    KArgs a = a_in;
    if (ALLON) { a.has_cutoff = 1; a.do_leak = 1; a.do_shot = 1; a.has_refr = 1; a.use_inten = 1; }
    extern __shared__ uint4 s_arec[];         // [CHAIN_SUB][BLOCK] k_ahead's records of the frames in flight
    __shared__ float s_lutL[FUSED ? 256 : 1]; // FUSED: lin-log tables and the pass's frame scalars
    __shared__ double s_lutI[FUSED ? 256 : 1];
    __shared__ double s_dtau[FUSED ? CHAIN_K_MAX : 1], s_shot[FUSED ? CHAIN_K_MAX : 1];
    __shared__ float s_dtime[FUSED ? CHAIN_K_MAX : 1];
    constexpr bool U8 = sizeof(FT) == 1;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int g = blockIdx.x;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    if (ca.prio) __builtin_amdgcn_s_setprio(3); // the dependency chain outranks the emission waves sharing the SIMD
    V2E_STAMP_C(0);
    // measurement (v2e_emu_launch_stamps; off: one scalar load and branch per launch): the launch's first workgroup start and last
    // workgroup end on the device's 100 MHz wall clock -- its duration as it runs in the TIMED configuration (graph replay, side streams)
    unsigned long long *const stamp = *ca.stamp_pp;
    if (stamp && tid == 0) atomicMax(stamp + 2 * ca.lidx, ~wall_clock64());
    uint32_t fbase = 0u;
    const char *frames = nullptr;
    if (FUSED) {
        fbase = *ca.fidx_base;
        frames = (const char *)*ca.frames_pp;
        if (U8) {
            s_lutL[tid] = a.lut_L[tid];
            s_lutI[tid] = a.lut_I[tid];
        }
    }
    const bool need_r = a.do_leak && a.jit_f != 0.f;
    // (the seed's Philox round keys precomputed for the FUSED calls, as k_ahead does: 20 more live scalar registers in a kernel that
    //  already spills them -- 64 clips 14.4-14.6 -> 13.7-13.8 Gev/s, 1280x720 unchanged; round 6, A/B of two builds in one session)

    for (int clip = blockIdx.y; clip < ca.n_clips; clip += (int)gridDim.y) {
        if (clip != (int)blockIdx.y) __syncthreads(); // the LDS tables of the previous clip are no longer read
        const size_t sp = (size_t)clip * a.npx_pad + p;
        // A pass's records go through LDS, CHAIN_SUB frames at a time, their loads in flight at once.  The next CHAIN_SUB
        // frames' records are fetched into registers BEFORE the current ones are processed and moved to LDS after them.
        // Every thread reads back only what it wrote: no barrier.
        // ring slot of a pass's frame k: the slot of its first frame (from the host: no division in the kernel) + k, wrapped (K <= D)
        auto wrap_slot = [&](const int sl) __attribute__((always_inline)) -> int { return sl >= ca.D ? sl - ca.D : sl; };
        // !FUSED: the ring planes through buffer resources -- the ring slot goes into the instruction's SCALAR offset, the lane's pixel
        // into its vector offset, computed once: an access at a sub-pass boundary is one scalar multiply and the memory instruction
        // where a 64-bit `(slot * clips + clip) * pixels + p` per access was a quarter of the boundary's ~300 instructions (round-4
        // review, item 2a).  The host guarantees the planes stay below 4 GB on this path (chain_fused_records).
        const uint32_t lane_px = (uint32_t)((size_t)clip * a.npx_pad + p);
        const uint32_t slot_px = (uint32_t)((size_t)ca.n_clips * a.npx_pad); // pixels of one ring slot
        const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ca.rec, 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t cnt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ca.cnt, 0, -1, 0x00020000);
        auto load_rec = [&](const int sl) __attribute__((always_inline)) -> uint4 {
            const v2e_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (int)(lane_px * 16u), (int)((uint32_t)sl * slot_px * 16u), 0);
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        auto stage = [&](const int sl0, const int fn) __attribute__((always_inline)) { // the first CHAIN_SUB frames of a pass (sl0: the first one's slot)
            if (FUSED) return;
            // (a lane beyond the frame stages ZERO records: with its zero state they give no events -- the frame loop carries no
            // per-lane validity test)
            uint4 t[CHAIN_SUB];
            int sl = sl0;
#pragma unroll
            for (int j = 0; j < CHAIN_SUB; ++j) {
                t[j] = make_uint4(0u, 0u, 0u, 0u);
                if (j < fn && valid) t[j] = load_rec(sl);
                if (++sl == ca.D) sl = 0;
            }
#pragma unroll
            for (int j = 0; j < CHAIN_SUB; ++j)
                if (j < fn) s_arec[(size_t)j * BLOCK + tid] = t[j];
        };
        auto fill_scalars = [&](const int fs, const int fn) __attribute__((always_inline)) { // FUSED: the pass's frame scalars
            if (!FUSED) return;
            __syncthreads();
            if (tid < fn) {
                const FrameCtl *c = ca.ctl + (size_t)(fs + tid) * ca.n_clips + clip;
                s_dtau[tid] = c->dt_over_tau;
                s_shot[tid] = c->shot_base;
                s_dtime[tid] = (float)(c->t_frame - c->t_prev);
            }
            __syncthreads();
        };
        // ---- every load of the prologue is issued before the first one is consumed (one memory round trip, not one per
        // item): read-only planes, state, the previous launch's rule-on row, this launch's frame scalars, and last the
        // staged records, whose arrival (in-order retirement) means everything before them has arrived too
        float thp = 1.f, thn = 1.f;
        R b = (R)0, lp = (R)0;
        float tsm = 0.f;
        const bool lp_state = a.has_cutoff || a.do_shot || ca.store_out; // lp_log_frame carried through the launch
        if (valid) {
            thp = a.pos_thres[sp];
            thn = a.neg_thres[sp];
            b = ((const R *)ca.base_in)[sp];
            if (lp_state) lp = ((const R *)ca.lp_in)[sp];
            if (a.has_refr) tsm = ca.ts_in[sp];
        }
        bool own = !(a.has_refr && ca.pnf > 0);
        uint32_t gM_v = 0u; // lane k: rule-on max of the previous launch's frame k as its own pass left it
        if (!own && lane < ca.pnf) gM_v = ca.gM_prev[(size_t)clip * ca.K + lane];
        // lane k of every wave: the rule threshold (FrameCtl::refr_on_n) of frame k of this launch / of the previous one
        uint32_t mon_own = 0xFFFFFFFFu, mon_prev = 0xFFFFFFFFu;
        if (a.has_refr) {
            if (lane < ca.nf) mon_own = ca.ctl[(size_t)(ca.f0 + lane) * ca.n_clips + clip].refr_on_n;
            if (!own && lane < ca.pnf) mon_prev = ca.ctl[(size_t)(ca.pf0 + lane) * ca.n_clips + clip].refr_on_n;
        }
        double dtau_r = 0.0, shot_r = 0.0;
        float dtime_r = 0.f, nr = 0.f;
        if (FUSED && tid < ca.nf) {
            const FrameCtl *c = ca.ctl + (size_t)(ca.f0 + tid) * ca.n_clips + clip;
            dtau_r = c->dt_over_tau;
            shot_r = c->shot_base;
            dtime_r = (float)(c->t_frame - c->t_prev);
        }
        if (FUSED && valid && a.do_leak) nr = a.noise_rate[sp];
        stage(ca.slot_f0, ca.nf); // this launch's own frames (almost always the only pass)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(b), "+v"(lp), "+v"(tsm), "+v"(thp), "+v"(thn), "+v"(gM_v), "+v"(mon_own), "+v"(mon_prev) : : "memory");
        if (FUSED) {
            if (tid < CHAIN_K_MAX) {
                s_dtau[tid] = dtau_r;
                s_shot[tid] = shot_r;
                s_dtime[tid] = dtime_r;
            }
            __syncthreads();
        }
        const float lk = a.leak_hz_f * nr;                                          // FUSED: emulator_utils.py:126
        const float ppre = a.scalar_thres ? a.pos_pre_scalar : a.pos_nom_f / thp;   // FUSED: emulator.py:475-478
        const float npre = a.scalar_thres ? a.neg_pre_scalar : a.neg_nom_f / thn;
        const R tpd = a.scalar_thres ? (R)a.pos_div : (R)thp;       // emulator_utils.py:154-157 divisors
        const R tnd = a.scalar_thres ? (R)a.neg_div : (R)thn;
        R rtp = (R)1 / tpd, rtn = (R)1 / tnd;
        // (opaque to the optimiser: it would otherwise sink the two divisions into the frame loop, behind the select)
        asm volatile("" : "+v"(rtp), "+v"(rtn));
        V2E_STAMP_C(1);
        // ---- passes: [redo of the previous launch]* then this launch's own frames
        bool redone = false;
        int round = 0, last_exact = -1;
        uint32_t pred_v = 0u;  // lane k: the M frame k was run under in the pass whose row is being checked
        uint32_t exact_v = 0u; // lane k: M of the previous launch's frame k in the coming redo pass (0: speculate)
        int c0 = 0;            // first frame of the pass (a redo pass may restart at a checkpoint)
        int bar_idx = 0;       // rendezvous of this launch so far (each has its own counter; the same sequence in every workgroup)
        // Lock-step frame of a redo pass.  Fixing a rule-on frame j leaves the pixels whose events the rule filtered with their
        // brightness difference, so frame j + 1 fires more and turns out rule-on itself in about half the cases -- a surprise no
        // prediction covers, which used to cost one more whole pass.  A redo pass therefore takes the frame right behind a frame
        // it finalised by the rule in LOCK-STEP: counts, publish, one (fence-free, ~5 us) rendezvous, the frame's exact M read
        // back, finalise -- and goes on like that while the frames keep being rule-on.
        int lock_k = -1;
        for (;;) {
            int fs, fn;
            uint32_t *gM_dst;
            if (!own) {
                // first frame after last_exact on which some wave reached the rule threshold: lane k = frame k
                if (round > 0) {
                    gM_v = 0u;
                    // (an agent-scope load: other workgroups' atomicMax of this launch, with no acquire fence behind the rendezvous)
                    if (lane < ca.pnf) gM_v = __hip_atomic_load(ca.gM_prev + ((size_t)round * ca.n_clips + clip) * ca.K + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // The pass that wrote this row ran frame k > last_exact under the PREDICTION pred_v[k] (0 = rule off; the own
                // pass predicts 0 everywhere).  Where row and prediction agree the frame was computed exactly; the first
                // disagreement j has everything before it exact, hence M(j) = row[j] exact (0: the rule is off there after
                // all).  The next pass takes j exactly and the row's values after j as its predictions -- a launch with
                // several rule-on frames usually settles in ONE redo pass instead of one pass per such frame.
                const unsigned long long hit = __ballot(gM_v != pred_v && lane > last_exact);
                const int j = hit ? (int)__builtin_ctzll(hit) : -1;
                if (j < 0) {
                    own = true;
                } else {
                    if (lane > last_exact) exact_v = gM_v; // < j: verified, j: exact, > j: predictions
                    pred_v = lane > j ? gM_v : 0u;
                    last_exact = j;
                    ++round;
                    redone = true;
                    lock_k = (V2E_CHAIN_LOCKSTEP && ca.lockstep && __builtin_amdgcn_readlane((int)gM_v, j) != 0 && j + 1 < ca.pnf) ? j + 1 : -1;
                    // restart point: every frame before j is exact, so is the checkpoint at or below j (written by the
                    // own pass, or by the redo pass before this one, in which the frames up to j were exact already)
                    c0 = ca.ckp_base ? (j / CHAIN_SUB) * CHAIN_SUB : 0;
                    if (valid) {
                        if (c0 == 0) { // the previous launch's input state
                            b = ((const R *)ca.base_pin)[sp];
                            if (lp_state) lp = ((const R *)ca.lp_pin)[sp];
                            tsm = ca.ts_pin[sp];
                        } else {
                            const size_t cs = ((size_t)(c0 / CHAIN_SUB - 1) * ca.n_clips + clip) * a.npx_pad + p;
                            b = ((const R *)ca.ckp_base)[cs];
                            if (lp_state) lp = ((const R *)ca.ckp_lp)[cs];
                            tsm = ca.ckp_ts[cs];
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b), "+v"(lp), "+v"(tsm) : : "memory");
                }
            }
            if (own) { fs = ca.f0; fn = ca.nf; gM_dst = ca.gM_cur + (size_t)clip * ca.K; c0 = 0; }
            else { fs = ca.pf0; fn = ca.pnf; gM_dst = ca.gM_prev + ((size_t)round * ca.n_clips + clip) * ca.K; }
            const int fs_slot = own ? ca.slot_f0 : ca.slot_pf0;
            if (redone) { // a redo pass, or the own pass after one: its inputs replace what the prologue staged
                fill_scalars(fs, fn);
                stage(wrap_slot(fs_slot + c0), fn - c0);
            }
            const uint32_t mon_v = own ? mon_own : mon_prev;
            // checkpoints of this pass go to the set of the launch whose frames it runs
            void *const ck_base = own ? ca.ckc_base : ca.ckp_base;
            void *const ck_lp = own ? ca.ckc_lp : ca.ckp_lp;
            float *const ck_ts = own ? ca.ckc_ts : ca.ckp_ts;
            if (own) V2E_STAMP_C(2);
            float r_odd = 0.f, u_odd = 0.f, r_even = 0.f, u_even = 0.f; // FUSED: the draws of the current frame pair
            bool have_pair = false;
            // lane k: M of frame k where this pass finalised it by the rule, else 0 -- written to the frames' ruleM slots behind the
            // pass (a store per frame and workgroup cost the common path three scalar instructions and a branch)
            uint32_t rule_v = 0u;
            // guard of the reciprocal floor division (below): a threshold that is not positive takes the generic path
            const bool thr_bad = !(tpd > (R)0) || !(tnd > (R)0);
            constexpr double QMAX = sizeof(R) == 8 ? 1.0e9 : 1.0e6; // quotients beyond it (and NaN) take the generic path
            // One frame of one pixel.  k: frame of the pass; js: its slot among the CHAIN_SUB frames staged in LDS (k % CHAIN_SUB; a
            // compile-time constant in the unrolled sub-pass); returns the frame's count word.
            auto frame_body = [&](const int k, const int js) __attribute__((always_inline)) -> uint32_t {
                const int f = fs + k;
                const uint32_t Mon = (uint32_t)__builtin_amdgcn_readlane((int)mon_v, k);
                uint32_t exM = own ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)exact_v, k);
                // the frame's record: eps (+ shot decisions in its two free bits), lin-log value, leak step
                uint4 rc = make_uint4(0u, 0u, 0u, 0u);
                if (FUSED) {
                    FT px = (FT)0;
                    if (valid) px = ((const FT *)(frames + (size_t)f * ca.frame_stride))[(size_t)clip * a.npx + p];
                    const uint32_t gf = fbase + (uint32_t)f;
                    if (need_r || a.do_shot) { // one Philox call per pair of frames (v2e_detmath.h)
                        if (!have_pair || v2e_frame_half(gf) == 0u)
                            v2e_draw_pair(a.seed, (uint32_t)clip, v2e_frame_pair(gf), (uint32_t)p, need_r, &r_odd, &u_odd, &r_even, &u_even);
                        have_pair = true;
                    }
                    const bool even = v2e_frame_half(gf) != 0u;
                    rc = make_frame_record<FT>(a, px, s_lutL, s_lutI, s_dtau[k], s_shot[k], s_dtime[k], lk, thp, ppre, npre,
                                               even ? r_even : r_odd, even ? u_even : u_odd);
                } else {
                    rc = s_arec[(size_t)js * BLOCK + tid];
                }
                const double eps = __longlong_as_double((long long)(((unsigned long long)(rc.y & 0x3FFFFFFFu) << 32) | rc.x));
                const double L = (double)__uint_as_float(rc.z);
                R lpn; // emulator_utils.py:96
                if (a.has_cutoff) lpn = (R)((1.0 - eps) * (double)lp + eps * L);
                else lpn = (R)L;
                if (a.do_leak) b = b - (R)__uint_as_float(rc.w); // emulator_utils.py:131
                // counts (emulator_utils.py:137-173): diff has one sign, one exact floor division.  floor(|diff| / theta) from the
                // reciprocal computed once per launch: |diff| * (1 / theta) is within a few ulp of the quotient, so its floor q0 is the
                // true floor or one off, and the remainder |diff| - q0 theta -- EXACT in one fma for such a q0 -- says which (the value
                // floor_div_pos returns, which equals c10::div_floor_floating for these operands).  The fix-up is applied to the
                // integer; anything unusual (NaN, a quotient beyond QMAX, a threshold <= 0) takes floor_div_pos on the operands the
                // reference's relu leaves.
                const R diff = (lpn + (R)0.0f) - b;
                const bool is_pos = diff > (R)0;
                const bool neg = !is_pos;
                const R bt = is_pos ? tpd : tnd, rbt = is_pos ? rtp : rtn;
                const R am = sizeof(R) == 8 ? (R)__builtin_fabs((double)diff) : (R)__builtin_fabsf((float)diff);
                const R q0 = floor(am * rbt);
                const R r0 = fma(-q0, bt, am);
                int mag = (int)q0;
                mag -= (r0 < (R)0) ? 1 : 0;
                mag += (r0 >= bt) ? 1 : 0;
                asm volatile("" : "+v"(mag)); // (every lane takes the arithmetic above; the compiler otherwise sinks it under the guard's mask)
                if (__builtin_expect(!(q0 <= (R)QMAX) || thr_bad, 0)) {
                    const R mg = is_pos ? diff : ((-diff) > (R)0 ? -diff : (R)0);
                    const int q = (int)floor_div_pos<R>(mg, bt);
                    mag = q > 0 ? q : 0;
                }
                if (FUSED && !valid) mag = 0; // (k_ahead's records of a pixel beyond the frame are zero: no events by themselves)
                // count word: count | sign (events only) | the record's shot decisions
                uint32_t cw = ((uint32_t)mag & CNT_MASK) | ((mag > 0 && neg) ? CNT_NEG : 0u);
                if (a.do_shot) cw |= (rc.y >> 5) & (CNT_SHOT_ON | CNT_SHOT_OFF);
                if (FUSED && !valid) cw = 0u;
                if (FUSED) {
                    if (valid) WT_STORE(&ca.cnt[((size_t)wrap_slot(fs_slot + k) * ca.n_clips + clip) * a.npx_pad + p], cw);
                }
                const int magv = mag;
                // speculation check: only a wave with a lane that reaches the rule threshold says so
                if (a.has_refr && k > (own ? -1 : last_exact) && __ballot((uint32_t)magv >= Mon) != 0ull) {
                    const int wm = wave_max_i32(magv);
                    if (lane == 0) atomicMax(gM_dst + k, (uint32_t)wm);
                }
                if (V2E_CHAIN_LOCKSTEP && !own && k == lock_k) { // lock-step: every frame before this one is exact, so the published maximum is M(k)
                    __builtin_amdgcn_s_setprio(0);
                    const bool okb = clip_barrier(ca.bar_prev + (size_t)bar_idx * ca.n_clips + clip, (unsigned)ca.ngroups, ca.bar_light != 0);
                    ++bar_idx;
                    if (ca.prio) __builtin_amdgcn_s_setprio(3);
                    if (!okb && tid == 0) atomicOr(&ca.recs[(size_t)ca.pf0 * ca.n_clips + clip].flags, V2E_FLAG_SYNC_TIMEOUT);
                    uint32_t mk = 0u;
                    if (lane == 0) mk = __hip_atomic_load(gM_dst + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mk = (uint32_t)__builtin_amdgcn_readfirstlane((int)mk);
                    exM = mk; // >= the rule threshold, or 0: nobody reached it
                    if (lane == k) { exact_v = mk; pred_v = mk; }
                    last_exact = k;
                    lock_k = (mk != 0u && k + 1 < fn) ? k + 1 : -1;
                }
                // ---- finalise (emulator.py:830-842, 936-942) -- by the refractory rule where M is known to switch it on.
                // The rule-on walk is its own branch: nothing it loads (timestamp tables) may be live in the common path,
                // or the common path inherits a wait for every outstanding store.
                int fcount = magv;
                if (exM != 0u) {
                    const FrameCtl *c = ca.ctl + (size_t)f * ca.n_clips + clip;
                    const FrameTab ftb(c, lane);
                    bool ruled = false;
                    const TsGen tg = frame_tsgen(a, c, ftb, (int)exM, ruled);
                    if (ruled) {
                        if (lane == k) rule_v = exM;
                        if (valid) ca.tsold[((size_t)wrap_slot(fs_slot + k) * ca.n_clips + clip) * a.npx_pad + p] = tsm; // ts_mem as it was
                        fcount = 0;
                        // (bounded: beyond max_iters <= 1024 events the frame is flagged and the run fails anyway; a wild count must not
                        // turn into minutes of spinning)
                        const int nit = min(magv, 1 << 16);
                        for (int i = 0; i < nit; ++i) { // emulator.py:836-842, this pixel's iterations
                            const float t = tg(i);
                            const float pt = 1.0f * t - tsm;
                            if (pt > a.refr_f) { tsm = t; ++fcount; }
                        }
                    }
                }
                // base += pos_events * pos_thres; base -= neg_events * neg_thres (one of the two products is +0 and leaves the sum as it
                // is; x - y == x + (-y) bit for bit), then the shot-noise reset -- unconditionally: a masked branch around it cost more
                // scalar instructions than the arithmetic has vector ones
                {
                    float d = (float)fcount * (neg ? thn : thp);
                    d = neg ? -d : d;
                    const R bn = b + (R)d;
                    const bool shot = a.do_shot && (rc.y >> 30) != 0u;
                    b = shot ? lpn : bn;
                }
                lp = lpn;
                if (V2E_CHAIN_STAMPS && own && k < 12) V2E_STAMP_C(3 + k);
                return cw;
            };
            // records of the pass's frames [kn, kn + CHAIN_SUB) into registers (in flight while the frames before them compute)
            uint4 nx[CHAIN_SUB];
            auto fetch_next = [&](const int kn) __attribute__((always_inline)) {
                const int cn = min(CHAIN_SUB, fn - kn); // <= 0: none
                int sl = cn > 0 ? wrap_slot(fs_slot + kn) : 0;
#pragma unroll
                for (int j = 0; j < CHAIN_SUB; ++j) {
                    nx[j] = make_uint4(0u, 0u, 0u, 0u);
                    if (!FUSED && valid && j < cn) nx[j] = load_rec(sl);
                    if (++sl == ca.D) sl = 0;
                }
            };
            fetch_next(c0 + CHAIN_SUB);
            for (int k0 = c0; k0 < fn; k0 += CHAIN_SUB) {
                if (ck_base && k0 > c0 && valid) { // state before frame k0
                    const size_t cs = ((size_t)(k0 / CHAIN_SUB - 1) * ca.n_clips + clip) * a.npx_pad + p;
                    ((R *)ck_base)[cs] = b;
                    if (lp_state) ((R *)ck_lp)[cs] = lp;
                    ck_ts[cs] = tsm;
                }
                __builtin_amdgcn_sched_barrier(0);
                const int kend = min(k0 + CHAIN_SUB, fn);
                const int slot0 = wrap_slot(fs_slot + k0);
                // the sub-pass's count words stay in registers and go out together behind it (a store inside the frame loop would
                // stand between the loop and the arrival of the next sub-pass's records: vector memory operations retire in order)
                uint32_t cwq[CHAIN_SUB];
#pragma unroll
                for (int j = 0; j < CHAIN_SUB; ++j) cwq[j] = 0u;
                if (kend - k0 == CHAIN_SUB) { // a full sub-pass, unrolled: LDS offsets, ring slots and lane indices fold
#pragma unroll
                    for (int j = 0; j < CHAIN_SUB; ++j) cwq[j] = frame_body(k0 + j, j);
                } else {
#pragma unroll
                    for (int j = 0; j < CHAIN_SUB; ++j)
                        if (k0 + j < kend) cwq[j] = frame_body(k0 + j, j);
                }
                if (!FUSED) {
                    const int rem = fn - k0 - CHAIN_SUB; // frames of the pass behind this sub-pass
                    // (A straight-line form of this boundary for full sub-passes whose slots do not wrap -- 8 LDS writes, 8 loads, 8 stores, no
                    // per-record test -- was built in round 6 and measured 1-2 us per launch SLOWER, A/B x 3 in one session: the tests are
                    // scalar instructions beside vector work; profiles/r06_emulator_experiments.txt item 11.)
                    const int nnext = min(CHAIN_SUB, rem); // frames of the next sub-pass (<= 0: none)
#pragma unroll
                    for (int j = 0; j < CHAIN_SUB; ++j) // (every lane: a lane beyond the frame keeps zero records in LDS)
                        if (j < nnext) s_arec[(size_t)j * BLOCK + tid] = nx[j];
                    // the sub-pass after the next: its loads go out BEFORE this sub-pass's count words (vector memory operations
                    // retire in order; the wait above then only ever covers operations a whole sub-pass old)
                    fetch_next(k0 + 2 * CHAIN_SUB);
                    if (valid) {
                        int sl = slot0;
#pragma unroll
                        for (int j = 0; j < CHAIN_SUB; ++j) { // the sub-pass's count words
                            if (k0 + j < kend)
                                __builtin_amdgcn_raw_buffer_store_b32(cwq[j], cnt_rsrc, (int)(lane_px * 4u), (int)((uint32_t)sl * slot_px * 4u), 17); // sc0 sc1: written through, as WT_STORE
                            if (++sl == ca.D) sl = 0;
                        }
                    }
                }
            }
            // the pass's ruleM slots: M where the rule finalised the frame, 0 elsewhere (the emission reads them; one wave writes them)
            if (a.has_refr && g == 0 && tid < WAVE) {
                int sl = fs_slot + lane;
                if (sl >= ca.D) sl -= ca.D;
                if (lane >= c0 && lane < fn) WT_STORE(&ca.ruleM[(size_t)sl * ca.n_clips + clip], rule_v);
            }
            if (own) break;
            // a redo pass: leave the corrected state where the next launch's own redo would look for it, and let every
            // workgroup of the clip publish before the check is repeated
            if (valid) {
                ((R *)ca.base_fix)[sp] = b;
                if (lp_state) ((R *)ca.lp_fix)[sp] = lp;
                ca.ts_fix[sp] = tsm;
            }
            if (last_exact >= ca.pnf - 1) continue; // the fixed frame was the launch's last: nothing left to verify, no rendezvous
            // (not at raised priority: a spinning wave that outranks the other kernels' waves on its SIMD keeps them from
            // finishing, and the workgroups this one waits for may need their slots)
            __builtin_amdgcn_s_setprio(0);
            const bool ok = clip_barrier(ca.bar_prev + (size_t)bar_idx * ca.n_clips + clip, (unsigned)ca.ngroups, ca.bar_light != 0);
            ++bar_idx;
            if (ca.prio) __builtin_amdgcn_s_setprio(3);
            if (!ok && tid == 0) atomicOr(&ca.recs[(size_t)ca.pf0 * ca.n_clips + clip].flags, V2E_FLAG_SYNC_TIMEOUT);
        }
        if (valid && (ca.nf > 0 || ca.store_out || redone)) {
            ((R *)ca.base_out)[sp] = b;
            if (ca.nf > 0 || lp_state) ((R *)ca.lp_out)[sp] = lp;
            if (a.has_refr) ca.ts_out[sp] = tsm;
        }
        V2E_STAMP_C(15);
    }
    if (stamp && tid == 0) atomicMax(stamp + 2 * ca.lidx + 1, wall_clock64());
}

// ------------------------------------------------------------------ emission side of the chain
struct CFrame { // per (batch frame, clip)
    int M;
    uint32_t n_events, n_signal, discarded;
};

struct CEmitArgs {
    const FrameCtl *ctl;
    v2e_frame_rec *recs;
    const uint32_t *fidx_base;
    int f0, nE, D, n_clips, nwp, nwaves, E;
    int slot0;           // f0 % D, from the host
    const uint32_t *cnt;
    const uint32_t *ruleM; // [D][n_clips] (refractory runs) or nullptr
    uint16_t *wmax;      // [E][n_clips][nwp] per-group max count (k_ctot); nwaves = groups of 256 pixels, nwp = that padded to 16
    uint16_t *wtot;      // [E][n_clips][nkeys_cap][nwp] per-group key totals, <= 256 each (k_ctot)
    const float *tsold;
    CFrame *cf;          // [E][n_clips]
    uint32_t *cT;        // [E][n_clips][nkeys_cap] events per key over all waves
    uint32_t *ckbase;    // [E][n_clips][nkeys_cap] first row of the key's iteration within the frame (signal keys)
    uint32_t *cperm;     // [E][n_clips][max_iters][8] shuffle round keys k0..k3, sh, a, amask, n per iteration
    uint32_t *cpre;      // [E][n_clips][nkeys_cap][nwp] exclusive prefix over waves per key
    float4 *events;
    unsigned long long cap;
    const unsigned long long *off_in;
    unsigned long long *off_out;
    unsigned *cdone;     // [E][n_clips] k_cframe: row workgroups done per frame (self-resetting)
    int capw, ich; // event records per wave in LDS; iterations per pass of k_cemit (64 * ich <= capw, 2 * ich <= 62)
    int zpw_tot, zpw_emit; // frames a workgroup of k_ctot / k_cemit walks (grid z = ceil(nE / that)); see enqueue_run_chain
    int coff_in_cemit;     // the next batch's event offset is written by k_cemit (one stream for tables and rows) instead of k_coff
    uint32_t *cmask;       // [E][n_clips][nkeys_cap][nwp][2 GPX] (key-major: a frame uses its low keys only, so what is touched is dense) per (key, group): which of the group's pixels have an event of that
                           // (iteration, polarity), one 64-bit ballot per sub-group (k_ctot -> k_cpull); nullptr: k_cemit writes the rows
    int wpf, p2;           // k_cpull: workgroups per frame; length of a prefix row in LDS (power of two >= nwaves; two-level: >= nwp / 16)
    uint32_t *cpre16;      // [E][n_clips][nkeys_cap][nwp / 16] every 16th entry of cpre (k_cframe*): the coarse level of k_cpull<true>
};

// ---- Emission groups.  The event list is assembled per GROUP of 256 consecutive pixels, one wave per group (GPX = 4 pixels
// per lane: sub-group j = pixels [64 j, 64 j + 64) of the group).  A frame of the benchmark clip has ~25 events per 64
// pixels (9 at 1280x720 noisy): with one wave per 64 pixels most of what the emission kernels executed was per-wave set-up
// and a 64-lane event pass that was a third full; with 256 pixels per wave the set-up, the per-key tables (352 entries per
// frame and key at 346x260 instead of 1 408) and the prefix over them shrink fourfold and the event pass runs full.
#ifndef V2E_GPX
#define V2E_GPX 4
#endif
constexpr int GPX = V2E_GPX;
constexpr int GROUP_PX = GPX * WAVE;
constexpr int GROUP_SRC_BITS = GPX <= 4 ? 8 : (GPX <= 8 ? 9 : 10); // bits of a pixel's index within its group

// What the event list needs from a frame's count words beyond the words themselves, per group (every wave on its own, no
// workgroup barrier): the group's max count and its (iteration, polarity) event totals (ballot / popcount) as key-major u16
// rows [key][group]: key 0/1 shot ON/OFF, key 2+2i / 3+2i iteration i ON/OFF -- after the refractory filter on the frames
// the chain finalised by the rule (ruleM != 0: the recurrence against ts_mem as it was, emulator.py:836-842).  A prefix
// over groups (k_cframe) turns them into row offsets; k_cemit writes the rows.  A wave takes CTOT_ZF frames of its group
// (their count words in flight together).
constexpr int CTOT_ZF = 2;

__global__ __launch_bounds__(BLOCK) void k_ctot(KArgs a, CEmitArgs ea)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y;
    const int grp = blockIdx.x * (BLOCK / WAVE) + wave;
    if (grp >= ea.nwaves) return;
    const int p0 = grp * GROUP_PX + lane;
    const int zb = (int)blockIdx.z * CTOT_ZF;
    uint32_t cwq[CTOT_ZF][GPX], rMq[CTOT_ZF];
    int slotq[CTOT_ZF];
    {
        int sl = ea.slot0 + zb; // nE <= D
        if (sl >= ea.D) sl -= ea.D;
#pragma unroll
        for (int q = 0; q < CTOT_ZF; ++q) {
            slotq[q] = sl;
            rMq[q] = 0u;
#pragma unroll
            for (int j = 0; j < GPX; ++j) {
                cwq[q][j] = 0u;
                if (zb + q < ea.nE && p0 + j * WAVE < a.npx) cwq[q][j] = ea.cnt[((size_t)sl * ea.n_clips + clip) * a.npx_pad + p0 + j * WAVE];
            }
            if (zb + q < ea.nE && ea.ruleM) rMq[q] = ea.ruleM[(size_t)sl * ea.n_clips + clip];
            if (++sl == ea.D) sl = 0;
        }
    }
#pragma unroll
    for (int q = 0; q < CTOT_ZF; ++q) {
        const int z = zb + q;
        if (z >= ea.nE) break;
        const size_t zc = (size_t)z * ea.n_clips + clip;
        const uint32_t rM = (uint32_t)__builtin_amdgcn_readfirstlane((int)rMq[q]);
        int magv[GPX];
        bool neg[GPX];
        unsigned long long negb[GPX]; // the sub-group's OFF lanes (a lane without events is neither)
        int mmax = 0;
        uint32_t son = 0, soff = 0;
        unsigned long long sob[GPX], sfb[GPX];
#pragma unroll
        for (int j = 0; j < GPX; ++j) {
            const uint32_t cw = cwq[q][j];
            magv[j] = (int)(cw & CNT_MASK);
            neg[j] = (cw & CNT_NEG) != 0;
            negb[j] = __ballot(neg[j]);
            mmax = max(mmax, magv[j]);
            sob[j] = __ballot((cw & CNT_SHOT_ON) != 0);
            sfb[j] = __ballot((cw & CNT_SHOT_OFF) != 0);
            son += (uint32_t)__popcll(sob[j]);
            soff += (uint32_t)__popcll(sfb[j]);
        }
        const int wm = wave_max_i32(mmax);
        if (lane == 0) ea.wmax[zc * ea.nwp + grp] = (uint16_t)min(wm, 65535);
        uint16_t *trow = ea.wtot + (zc * a.nkeys_cap) * ea.nwp + grp;
        // the pull's masks: lanes 0 .. 4 GPX - 1 hold the words of an iteration's ON key (2 GPX of them) and OFF key
        // (lanes 0 .. 2 GPX - 1: the ON key's row, the next 2 GPX: the OFF key's, one row further)
        uint32_t *mrow = ea.cmask ? ea.cmask + ((zc * a.nkeys_cap + (lane >= 2 * GPX ? 1 : 0)) * ea.nwp + grp) * (2 * GPX) + (lane & (2 * GPX - 1)) : nullptr;
        auto put_masks = [&](const int key_on, const unsigned long long (&on_m)[GPX], const unsigned long long (&off_m)[GPX]) __attribute__((always_inline)) {
            uint32_t mw = 0u;
#pragma unroll
            for (int j = 0; j < GPX; ++j) { // (no writelane builtin in this compiler)
                asm("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)on_m[j]), "n"(2 * j));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)(on_m[j] >> 32)), "n"(2 * j + 1));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)off_m[j]), "n"(2 * GPX + 2 * j));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(mw) : "s"((uint32_t)(off_m[j] >> 32)), "n"(2 * GPX + 2 * j + 1));
            }
            if (lane < 4 * GPX) mrow[(size_t)key_on * ea.nwp * (2 * GPX)] = mw;
        };
        const int wmc = min(wm, a.max_iters); // beyond max_iters the frame is flagged and not emitted
        const int nkw = 2 + 2 * wmc;
        bool ruled = false;
        TsGen tg(0.f, 0.f, 0.f, 1);
        float tsm[GPX];
#pragma unroll
        for (int j = 0; j < GPX; ++j) tsm[j] = 0.f;
        if (rM != 0u) { // rule-on frame (rare): the filter needs the frame's time stamps and ts_mem as it was
            const FrameCtl *c = ea.ctl + (size_t)(ea.f0 + z) * ea.n_clips + clip;
            const FrameTab ftb(c, lane);
            tg = frame_tsgen(a, c, ftb, (int)rM, ruled);
            if (ea.tsold) {
#pragma unroll
                for (int j = 0; j < GPX; ++j)
                    if (p0 + j * WAVE < a.npx) tsm[j] = ea.tsold[((size_t)slotq[q] * ea.n_clips + clip) * a.npx_pad + p0 + j * WAVE];
            }
        }
        for (int kb = 0; kb < nkw; kb += WAVE) {
            uint32_t mine = 0;
            const int i_lo = kb == 0 ? 0 : (kb - 2) / 2;
            const int i_hi = min((kb + WAVE - 2) / 2, wmc);
            for (int i = i_lo; i < i_hi; ++i) {
                uint32_t on = 0, off = 0;
                unsigned long long on_m[GPX], off_m[GPX];
                if (!ruled) { // the common case: one compare per sub-group, the polarity split on the scalar unit
#pragma unroll
                    for (int j = 0; j < GPX; ++j) {
                        const unsigned long long cb = __ballot(magv[j] > i);
                        on_m[j] = cb & ~negb[j];
                        off_m[j] = cb & negb[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < GPX; ++j) {
                        bool pass = magv[j] > i;
                        if (pass) {
                            const float t = tg(i);
                            const float pt = 1.0f * t - tsm[j];
                            pass = pt > a.refr_f;
                            if (pass) tsm[j] = t;
                        }
                        on_m[j] = __ballot(pass && !neg[j]);
                        off_m[j] = __ballot(pass && neg[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < GPX; ++j) {
                    on += (uint32_t)__popcll(on_m[j]);
                    off += (uint32_t)__popcll(off_m[j]);
                }
                if (mrow) put_masks(2 + 2 * i, on_m, off_m);
                const int kl = 2 + 2 * i - kb;
                if (lane == kl) mine = on;
                if (lane == kl + 1) mine = off;
            }
            if (kb == 0) {
                if (lane == 0) mine = son;
                if (lane == 1) mine = soff;
                if (mrow && (son | soff) != 0u) put_masks(0, sob, sfb);
            }
            if (kb + lane < nkw) trow[(size_t)(kb + lane) * ea.nwp] = (uint16_t)mine;
        }
    }
}

// 16 consecutive entries of a key row (u16 totals of 16 groups) with the rows a group did not write masked: a group writes
// only the rows of its own iterations; above them the table holds an older frame's values
__device__ __forceinline__ void key_row16(const uint16_t *__restrict__ row_at, const uint16_t *__restrict__ wm_at, uint32_t k, uint32_t v[16])
{
    const uint4 t0 = *(const uint4 *)row_at, t1 = *(const uint4 *)(row_at + 8);
    const uint4 m0 = *(const uint4 *)wm_at, m1 = *(const uint4 *)(wm_at + 8);
    const uint32_t tw[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    const uint32_t mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t wmj = (mw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        v[j] = k < 2u + 2u * wmj ? ((tw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) : 0u;
    }
}

// Small grids (a key row is a couple of steps of one wave): one workgroup per (frame, clip), one wave per key row.
__global__ __launch_bounds__(CFRAME_THREADS) void k_cframe1(KArgs a, CEmitArgs ea)
{
    __shared__ int s_red[CFRAME_THREADS / WAVE];
    __shared__ uint32_t s_T[2 * CHAIN_MAX_ITERS + 2];
    constexpr int NW = CFRAME_THREADS / WAVE; // one wave per key row: 16 rows at a time (a frame has 2 + 2 M of them)
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, z = blockIdx.z, fe = ea.f0 + z;
    const uint16_t *wm = ea.wmax + ((size_t)z * ea.n_clips + clip) * ea.nwp;
    int m = 0;
    for (int k = tid; k < ea.nwaves; k += CFRAME_THREADS) m = max(m, (int)wm[k]);
    m = wave_max_i32(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    int M = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) M = max(M, s_red[q]);
    M = __builtin_amdgcn_readfirstlane(M);
    CFrame *cf = ea.cf + (size_t)z * ea.n_clips + clip;
    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips + clip;
    if (M > a.max_iters) { // the frame is not emitted; the caller sees the flag
        if (tid == 0) {
            cf->M = M; cf->n_events = 0u; cf->n_signal = 0u; cf->discarded = 1u;
            rec->max_events = M;
            atomicOr(&rec->flags, V2E_FLAG_ITERS_CLAMPED);
        }
        return;
    }
    const int nk = 2 + 2 * M;
    const uint16_t *tot = ea.wtot + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap * ea.nwp;
    uint32_t *pre = ea.cpre + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap * ea.nwp;
    for (int k = wave; k < nk; k += NW) { // one wave per key row, 16 waves' totals per lane and step
        uint32_t carry = 0;
        for (int w0 = 0; w0 < ea.nwp; w0 += 16 * WAVE) {
            const int wi = w0 + lane * 16;
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0u;
            if (wi < ea.nwp) {
                key_row16(tot + (size_t)k * ea.nwp + wi, wm + wi, (uint32_t)k, v);
            }
            uint32_t lane_tot = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) lane_tot += v[j];
            uint32_t run = carry + wave_excl_scan_u32(lane_tot, lane);
            carry += wave_sum_u32(lane_tot);
            if (wi < ea.nwp) {
                uint32_t o[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { o[j] = run; run += v[j]; }
                uint4 *dst = (uint4 *)(pre + (size_t)k * ea.nwp + wi);
                if (ea.cpre16) ea.cpre16[(((size_t)z * ea.n_clips + clip) * a.nkeys_cap + k) * (ea.nwp / 16) + wi / 16] = o[0];
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
                dst[2] = make_uint4(o[8], o[9], o[10], o[11]);
                dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
            }
        }
        if (lane == 0) s_T[k] = carry;
    }
    __syncthreads();
    if (wave != 0) return;
    uint32_t *cT = ea.cT + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap;
    uint32_t *ckb = ea.ckbase + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap;
    uint32_t carry = 0, sum_on = 0, sum_off = 0;
    for (int kb = 0; kb < nk; kb += WAVE) {
        const int key = kb + lane;
        const uint32_t T_k = key < nk ? s_T[key] : 0u;
        const uint32_t sig = key >= 2 ? T_k : 0u;
        const uint32_t kbase = carry + wave_excl_scan_u32(sig, lane);
        if (key < a.nkeys_cap) { cT[key] = T_k; ckb[key] = kbase; }
        sum_on += wave_sum_u32((lane & 1) ? 0u : sig);
        sum_off += wave_sum_u32((lane & 1) ? sig : 0u);
        carry += wave_sum_u32(sig);
    }
    const uint32_t n_signal = carry;
    uint32_t *perm = ea.cperm + ((size_t)z * ea.n_clips + clip) * a.max_iters * 8;
    if (a.shuffle && a.rng_mode == V2E_RNG_PHILOX) {
        const uint32_t fbase = ea.fidx_base ? *ea.fidx_base : 0u;
        for (int i = lane; i < M; i += WAVE) {
            uint32_t pk[4], sh, aa, amask;
            const uint32_t n_i = s_T[2 + 2 * i] + s_T[3 + 2 * i];
            v2e_perm_shape(n_i, &sh, &aa, &amask);
            v2e_perm_keys(a.seed, (uint32_t)clip, fbase + (uint32_t)fe, (uint32_t)i, pk);
            uint4 *pp = (uint4 *)(perm + (size_t)i * 8);
            pp[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            pp[1] = make_uint4(sh, aa, amask, n_i);
        }
    }
    if (lane == 0) {
        const uint32_t son = a.do_shot ? s_T[0] : 0u, soff = a.do_shot ? s_T[1] : 0u;
        cf->M = M; cf->n_signal = n_signal; cf->n_events = n_signal + son + soff; cf->discarded = 0u;
        rec->max_events = M;
        rec->n_signal = n_signal;
        rec->n_events = n_signal + son + soff;
        rec->n_on = sum_on + son;
        rec->n_off = sum_off + soff;
    }
}

// Per frame: M, per key the prefix over waves and the total, prefix over keys, shuffle parameters.  One workgroup per
// (key row, frame, clip): a row's exclusive prefix over the frame's waves is a segmented scan by the workgroup's 16 waves
// (each takes a contiguous sixteenth of the row).  The workgroup that finishes LAST for a frame (a counter per frame)
// reads the rows' totals and builds what k_cemit needs per key and iteration.  CFRAME_ROWS rows have a workgroup each;
// frames with more rows have the last-row workgroup walk the rest.  (Large grids; small ones: k_cframe1.)
constexpr int CFRAME_ROWS = 24; // M <= 11 fully parallel

__global__ __launch_bounds__(CFRAME_THREADS) void k_cframe(KArgs a, CEmitArgs ea)
{
    constexpr int NW = CFRAME_THREADS / WAVE;
    __shared__ int s_red[NW];
    __shared__ uint32_t s_seg[NW];
    __shared__ uint32_t s_T[2 * CHAIN_MAX_ITERS + 2];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int r0 = blockIdx.x, clip = blockIdx.y, z = blockIdx.z, fe = ea.f0 + z;
    const size_t zc = (size_t)z * ea.n_clips + clip;
    const uint16_t *wm = ea.wmax + ((size_t)z * ea.n_clips + clip) * ea.nwp;
    int m = 0;
    for (int k = tid; k < ea.nwaves; k += CFRAME_THREADS) m = max(m, (int)wm[k]);
    m = wave_max_i32(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    int M = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) M = max(M, s_red[q]);
    M = __builtin_amdgcn_readfirstlane(M);
    CFrame *cf = ea.cf + zc;
    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips + clip;
    const bool discard = M > a.max_iters; // the frame is not emitted; the caller sees the flag
    const int nk = discard ? 0 : 2 + 2 * M;
    const uint16_t *tot = ea.wtot + ((size_t)z * ea.n_clips + clip) * a.nkeys_cap * ea.nwp;
    uint32_t *pre = ea.cpre + zc * a.nkeys_cap * ea.nwp;
    uint32_t *cT = ea.cT + zc * a.nkeys_cap;
    // this workgroup's row(s): r0, and for the last row workgroup every row beyond the grid
    const int seg = ((ea.nwp + NW - 1) / NW + 15) / 16 * 16; // entries per wave, a multiple of the 16 a lane takes
    for (int k = r0; k < nk; k += (r0 == (int)gridDim.x - 1 ? 1 : nk)) {
        const int lo = wave * seg, hi = min(lo + seg, ea.nwp);
        uint32_t wsum = 0; // pass 1: this wave's segment total
        for (int w0 = lo; w0 < hi; w0 += 16 * WAVE) {
            const int wi = w0 + lane * 16;
            uint32_t lane_tot = 0;
            if (wi < hi) {
                uint32_t v[16];
                key_row16(tot + (size_t)k * ea.nwp + wi, wm + wi, (uint32_t)k, v);
#pragma unroll
                for (int j = 0; j < 16; ++j) lane_tot += v[j];
            }
            wsum += wave_sum_u32(lane_tot);
        }
        __syncthreads(); // s_seg of the previous row is no longer read
        if (lane == 0) s_seg[wave] = wsum;
        __syncthreads();
        uint32_t carry = 0, total = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            carry += q < wave ? s_seg[q] : 0u;
            total += s_seg[q];
        }
        for (int w0 = lo; w0 < hi; w0 += 16 * WAVE) { // pass 2: prefixes (the row is L2-resident by now)
            const int wi = w0 + lane * 16;
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0u;
            if (wi < hi) {
                key_row16(tot + (size_t)k * ea.nwp + wi, wm + wi, (uint32_t)k, v);
            }
            uint32_t lane_tot = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) lane_tot += v[j];
            uint32_t run = carry + wave_excl_scan_u32(lane_tot, lane);
            carry += wave_sum_u32(lane_tot);
            if (wi < hi) {
                uint32_t o[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { o[j] = run; run += v[j]; }
                uint4 *dst = (uint4 *)(pre + (size_t)k * ea.nwp + wi);
                if (ea.cpre16) ea.cpre16[(((size_t)z * ea.n_clips + clip) * a.nkeys_cap + k) * (ea.nwp / 16) + wi / 16] = o[0];
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
                dst[2] = make_uint4(o[8], o[9], o[10], o[11]);
                dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
            }
        }
        if (tid == 0) cT[k] = total;
    }
    // ---- last workgroup of the frame: everything per key / per iteration
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(ea.cdone + zc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == gridDim.x - 1;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(ea.cdone + zc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next batch
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (discard) {
        if (tid == 0) {
            cf->M = M; cf->n_events = 0u; cf->n_signal = 0u; cf->discarded = 1u;
            rec->max_events = M;
            atomicOr(&rec->flags, V2E_FLAG_ITERS_CLAMPED);
        }
        return;
    }
    for (int k = tid; k < nk; k += CFRAME_THREADS) s_T[k] = __hip_atomic_load(cT + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (wave != 0) return;
    uint32_t *ckb = ea.ckbase + zc * a.nkeys_cap;
    uint32_t carry = 0, sum_on = 0, sum_off = 0;
    for (int kb = 0; kb < nk; kb += WAVE) {
        const int key = kb + lane;
        const uint32_t T_k = key < nk ? s_T[key] : 0u;
        const uint32_t sig = key >= 2 ? T_k : 0u;
        const uint32_t kbase = carry + wave_excl_scan_u32(sig, lane);
        if (key < nk) ckb[key] = kbase;
        sum_on += wave_sum_u32((lane & 1) ? 0u : sig);
        sum_off += wave_sum_u32((lane & 1) ? sig : 0u);
        carry += wave_sum_u32(sig);
    }
    const uint32_t n_signal = carry;
    uint32_t *perm = ea.cperm + zc * a.max_iters * 8;
    if (a.shuffle && a.rng_mode == V2E_RNG_PHILOX) {
        const uint32_t fbase = ea.fidx_base ? *ea.fidx_base : 0u;
        for (int i = lane; i < M; i += WAVE) {
            uint32_t pk[4], sh, aa, amask;
            const uint32_t n_i = s_T[2 + 2 * i] + s_T[3 + 2 * i];
            v2e_perm_shape(n_i, &sh, &aa, &amask);
            v2e_perm_keys(a.seed, (uint32_t)clip, fbase + (uint32_t)fe, (uint32_t)i, pk);
            uint4 *pp = (uint4 *)(perm + (size_t)i * 8);
            pp[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            pp[1] = make_uint4(sh, aa, amask, n_i);
        }
    }
    if (lane == 0) {
        const uint32_t son = a.do_shot ? s_T[0] : 0u, soff = a.do_shot ? s_T[1] : 0u;
        cf->M = M; cf->n_signal = n_signal; cf->n_events = n_signal + son + soff; cf->discarded = 0u;
        rec->max_events = M;
        rec->n_signal = n_signal;
        rec->n_events = n_signal + son + soff;
        rec->n_on = sum_on + son;
        rec->n_off = sum_off + soff;
    }
}

// Event offset of the next batch: this batch's plus its frames' event counts (one wave per clip; behind k_cframe on the tables
// stream, so that k_cemit of batch b + 1 needs nothing from k_cemit of batch b).
__global__ __launch_bounds__(WAVE) void k_coff(CEmitArgs ea)
{
    const int clip = blockIdx.x, lane = threadIdx.x;
    unsigned long long tot = 0;
    for (int z = lane; z < ea.nE; z += WAVE) tot += ea.cf[(size_t)z * ea.n_clips + clip].n_events;
    const uint32_t lo = wave_sum_u32((uint32_t)(tot & 0xFFFFFFu)), hi = wave_sum_u32((uint32_t)(tot >> 24));
    if (lane == 0) ea.off_out[clip] = ea.off_in[clip] + lo + ((unsigned long long)hi << 24);
}

// Event rows of one frame, every group (256 pixels, one wave) on its own: which of its pixels' iterations pass (the
// refractory recurrence against tsold on rule-on frames), ballot ranks, one 4-byte record per event in LDS; then one event per
// lane: row = frame offset + iteration base + shuffle(ON/OFF block offset + prefix over earlier groups + rank in group).
__global__ __launch_bounds__(BLOCK) void k_cemit(KArgs a, CEmitArgs ea)
{
    extern __shared__ uint32_t s_crec[]; // [BLOCK / WAVE][capw + WAVE]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, z = blockIdx.z, fe = ea.f0 + z;
    const int slot = ea.slot0 + z >= ea.D ? ea.slot0 + z - ea.D : ea.slot0 + z; // nE <= D
    const int grp = blockIdx.x * (BLOCK / WAVE) + wave;
    if (grp >= ea.nwaves) return;
    const size_t zc = (size_t)z * ea.n_clips + clip;
    const CFrame *cf = ea.cf + zc;
    const int p0 = grp * GROUP_PX + lane; // this lane's pixels: p0 + 64 j
    const size_t sp0 = ((size_t)slot * ea.n_clips + clip) * a.npx_pad + p0;
    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips;
    const FrameCtl *c = ea.ctl + (size_t)fe * ea.n_clips + clip;
    const uint32_t *cT = ea.cT + zc * a.nkeys_cap, *ckb = ea.ckbase + zc * a.nkeys_cap;
    const uint32_t *pre = ea.cpre + zc * a.nkeys_cap * ea.nwp + grp;
    const uint32_t *perm = ea.cperm + zc * a.max_iters * 8;
    const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
    const int ICH = ea.ich; // iterations per pass: their 2 * ICH keys in the wave's lanes, at most GROUP_PX * ICH records
    // ------------------------------------------------------------ every load of the common case, issued back to back (one
    // memory round trip per wave instead of one per early-exit test): frame table, offsets, count words, group max, timestamp
    // tables, and the first pass's per-key totals / prefixes / shuffle parameters (rows beyond the frame's keys hold older
    // frames' values: masked once M is known)
    const uint32_t off_lo = (uint32_t)ea.off_in[clip], off_hi = (uint32_t)(ea.off_in[clip] >> 32);
    const uint32_t nj = lane < z ? ea.cf[(size_t)lane * ea.n_clips + clip].n_events : 0u; // lane j: frame j of the batch (E <= 64)
    const int M_v = cf->M;
    const uint32_t nsig_v = cf->n_signal, nev_v = cf->n_events, disc_v = cf->discarded;
    uint32_t cw[GPX];
#pragma unroll
    for (int j = 0; j < GPX; ++j) cw[j] = p0 + j * WAVE < a.npx ? ea.cnt[sp0 + j * WAVE] : 0u;
    const int wm_v = (int)ea.wmax[zc * ea.nwp + grp];
    const FrameTab ftb(c, lane);
    uint32_t T_0 = 0, kbase_0 = 0, P_0 = 0;
    if (lane < 2 * ICH && 2 + lane < a.nkeys_cap) {
        T_0 = cT[2 + lane];
        kbase_0 = ckb[2 + lane];
        P_0 = pre[(size_t)(2 + lane) * ea.nwp];
    }
    uint4 pa_0 = make_uint4(0u, 0u, 0u, 0u), pb_0 = make_uint4(0u, 1u, 0u, 0u);
    if (shuf && lane < ICH && lane < a.max_iters) {
        pa_0 = ((const uint4 *)(perm + (size_t)lane * 8))[0];
        pb_0 = ((const uint4 *)(perm + (size_t)lane * 8))[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long ev0 = ((unsigned long long)off_hi << 32 | off_lo) + (unsigned long long)wave_sum_u32(nj & 0xFFFFFFu) +
                             ((unsigned long long)wave_sum_u32(nj >> 24) << 24); // + the events of the batch's earlier frames
    const int M = __builtin_amdgcn_readfirstlane(M_v);
    const uint32_t n_signal = __builtin_amdgcn_readfirstlane((int)nsig_v);
    const uint32_t n_events = __builtin_amdgcn_readfirstlane((int)nev_v);
    if (grp == 0 && lane == 0) {
        rec[clip].ev_offset = ev0;
        if (ea.coff_in_cemit && z == ea.nE - 1) ea.off_out[clip] = ev0 + n_events; // else k_coff's
        // rows beyond the capacity are dropped by the stores' bounds check (below): the frame says so here, once
        if (ev0 + n_events > ea.cap || n_events > 0x7FFFFFFu) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
    }
    if (__builtin_amdgcn_readfirstlane((int)disc_v)) return;
    const int wmw = __builtin_amdgcn_readfirstlane(wm_v);
    unsigned long long so[GPX], sf[GPX];
    unsigned long long any_shot = 0ull;
#pragma unroll
    for (int j = 0; j < GPX; ++j) {
        so[j] = __ballot((cw[j] & CNT_SHOT_ON) != 0);
        sf[j] = __ballot((cw[j] & CNT_SHOT_OFF) != 0);
        any_shot |= so[j] | sf[j];
    }
    if (wmw == 0 && any_shot == 0ull) return; // nothing of this group in the frame
    const int n = M > 0 ? M : 1;
    bool use_refr;
    const TsGen tg = frame_tsgen(a, c, ftb, n, use_refr);
    int mag[GPX];
    bool neg[GPX];
    unsigned long long negb[GPX]; // the sub-group's OFF lanes
    float tsm[GPX];
#pragma unroll
    for (int j = 0; j < GPX; ++j) {
        mag[j] = (int)(cw[j] & CNT_MASK);
        neg[j] = (cw[j] & CNT_NEG) != 0;
        negb[j] = __ballot(neg[j]);
        tsm[j] = 0.f;
    }
    if (use_refr && ea.tsold) { // ts_mem as it was before the frame (rule-on frames only)
#pragma unroll
        for (int j = 0; j < GPX; ++j)
            if (p0 + j * WAVE < a.npx) tsm[j] = ea.tsold[sp0 + j * WAVE];
    }
    // The frame's rows through a buffer resource over exactly the rows it may write -- base = the frame's first row, size = its
    // rows that fit below the capacity: a row beyond the capacity is dropped by the bounds check of the store itself (no compare,
    // no exec-mask branch per store site; k_cemit is bound by its scalar instructions at 1280x720), offsets are frame-relative
    // 32-bit numbers, and "events dropped" is one comparison per frame (group 0).
    const uint32_t ev0_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ev0), ev0_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ev0 >> 32));
    const unsigned long long ev0u = ((unsigned long long)ev0_hi << 32) | ev0_lo;
    const unsigned long long room = ev0u < ea.cap ? ea.cap - ev0u : 0ull;
    const unsigned long long frows = room < (unsigned long long)n_events ? room : (unsigned long long)n_events;
    const uint32_t fbytes = frows > 0x7FFFFFFull ? 0x7FFFFFF0u : (uint32_t)frows * 16u;
    const __amdgpu_buffer_rsrc_t ev_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(ea.events + (size_t)clip * ea.cap + ev0u), 0, (int)fbytes, 0x00020000);
    auto store_row = [&](const uint32_t rel, const float t, const float x, const float y, const float pol) __attribute__((always_inline)) {
        const v2e_f4 v = {t, x, y, pol};
        __builtin_amdgcn_raw_buffer_store_b128(v, ev_rsrc, (int)(rel * 16u), 0, 17); // sc0 sc1: written through (see store_event_wt)
    };
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float rcpW = 1.0f / (float)a.W;
    auto pixel_xy = [&](const uint32_t p, float &x, float &y) __attribute__((always_inline)) { // y = p / W, x = p % W, exactly
        uint32_t q;
        if (a.npx < (1 << 24)) { // float32 holds p exactly: the quotient estimate is the true one or one off
            q = (uint32_t)((float)p * rcpW);
            if (q * (uint32_t)a.W > p) --q;
            else if ((q + 1u) * (uint32_t)a.W <= p) ++q;
        } else {
            q = p / (uint32_t)a.W;
        }
        y = (float)q;
        x = (float)(p - q * (uint32_t)a.W);
    };
    uint32_t *rec_w = s_crec + (size_t)wave * (ea.capw + WAVE); // capw records + one dump word per lane
    const int iters = min(wmw, M);
    for (int i0 = 0; i0 < iters; i0 += ICH) {
        const int i1 = min(i0 + ICH, iters);
        const int key = 2 + 2 * i0 + lane; // lanes 0 .. 2 ICH - 1: the ON / OFF keys of iterations i0 .. i0 + ICH - 1
        const bool key_on = lane < 2 * ICH && key < 2 + 2 * i1;
        uint32_t T_k = 0, kbase_k = 0, P_k = 0;
        uint4 pa = make_uint4(0u, 0u, 0u, 0u), pb = make_uint4(0u, 1u, 0u, 0u);
        if (i0 == 0) { // prefetched
            if (key_on) { T_k = T_0; kbase_k = kbase_0; P_k = P_0; }
            if (shuf && lane < ICH && lane < i1) { pa = pa_0; pb = pb_0; }
        } else {
            if (key_on) {
                T_k = cT[key];
                kbase_k = ckb[key];
                P_k = pre[(size_t)key * ea.nwp];
            }
            if (shuf && lane < ICH && i0 + lane < i1) {
                pa = ((const uint4 *)(perm + (size_t)(i0 + lane) * 8))[0];
                pb = ((const uint4 *)(perm + (size_t)(i0 + lane) * 8))[1];
            }
        }
        // pass 1: a record per passing (pixel, iteration): source pixel of the group, iteration, polarity, rank within the group's
        // (iteration, polarity) block in pixel order (sub-group by sub-group)
        uint32_t nrec = 0;
        bool alive = true;
        for (int i = i0; i < i1; ++i) {
            unsigned long long cb[GPX], cand_any = 0ull;
#pragma unroll
            for (int j = 0; j < GPX; ++j) { cb[j] = __ballot(mag[j] > i); cand_any |= cb[j]; }
            if (cand_any == 0ull) { alive = false; break; }
            uint32_t run_on = 0, run_off = 0;
#pragma unroll
            for (int j = 0; j < GPX; ++j) {
                // a sub-group without a candidate at this iteration (most of them beyond the first two: a frame's large counts
                // sit on a few pixels) takes one scalar test; with the rule on a non-candidate is still put to the test below,
                // as the reference's `pos_cord * ts - timestamp_mem > refractory_period_s` does
                if (!use_refr && cb[j] == 0ull) continue;
                const bool cand = mag[j] > i;
                bool pass = cand;
                unsigned long long bo, bf;
                if (use_refr) {
                    const float t = tg(i);
                    const float pt = (cand ? 1.0f : 0.0f) * t - tsm[j];
                    pass = pt > a.refr_f;
                    if (pass) tsm[j] = t;
                    bo = __ballot(pass && !neg[j]);
                    bf = __ballot(pass && neg[j]);
                } else { // the polarity split of the candidates on the scalar unit
                    bf = cb[j] & negb[j];
                    bo = cb[j] & ~negb[j];
                }
                {   // every lane writes: a lane without an event to its own dump word behind the records (no exec-mask branch)
                    const uint32_t rank = neg[j] ? run_off + (uint32_t)__popcll(bf & lt) : run_on + (uint32_t)__popcll(bo & lt);
                    const uint32_t pos = pass ? nrec + (uint32_t)__popcll((bo | bf) & lt) : (uint32_t)(ea.capw + lane);
                    rec_w[pos] = (uint32_t)(j * WAVE + lane) | ((uint32_t)(i - i0) << GROUP_SRC_BITS) | ((neg[j] ? 1u : 0u) << (GROUP_SRC_BITS + 5)) |
                                 (rank << (GROUP_SRC_BITS + 6));
                }
                nrec += (uint32_t)__popcll(bo | bf);
                run_on += (uint32_t)__popcll(bo);
                run_off += (uint32_t)__popcll(bf);
            }
        }
        // pass 2: one event per lane
        for (uint32_t e0 = 0; e0 < nrec; e0 += WAVE) {
            const bool has = e0 + lane < nrec;
            const uint32_t r = has ? rec_w[e0 + lane] : 0u;
            const uint32_t src = r & ((1u << GROUP_SRC_BITS) - 1u);
            const int il = (int)((r >> GROUP_SRC_BITS) & 31u);
            const bool eneg = (r >> (GROUP_SRC_BITS + 5)) & 1u;
            const uint32_t rank = r >> (GROUP_SRC_BITS + 6);
            const int kl = 2 * il;
            const uint32_t it_base = (uint32_t)__shfl((int)kbase_k, kl);
            const uint32_t tot_on = (uint32_t)__shfl((int)T_k, kl);
            const uint32_t off = (uint32_t)__shfl((int)P_k, kl + (eneg ? 1 : 0));
            float ex, ey;
            pixel_xy((uint32_t)(grp * GROUP_PX) + src, ex, ey);
            uint32_t cidx = (eneg ? tot_on : 0u) + off + rank;
            if (shuf) {
                v2e_perm_t pm;
                pm.k[0] = (uint32_t)__shfl((int)pa.x, il); pm.k[1] = (uint32_t)__shfl((int)pa.y, il);
                pm.k[2] = (uint32_t)__shfl((int)pa.z, il); pm.k[3] = (uint32_t)__shfl((int)pa.w, il);
                pm.sh = (uint32_t)__shfl((int)pb.x, il); pm.a = (uint32_t)__shfl((int)pb.y, il);
                pm.amask = 0u; pm.n = (uint32_t)__shfl((int)pb.w, il);
                pm.rmask = (1u << pm.sh) - 1u;
                if (has) cidx = v2e_perm_apply(&pm, cidx);
            }
            if (has) store_row(it_base + cidx, tg(i0 + il), ex, ey, eneg ? -1.0f : 1.0f);
        }
        if (!alive) break;
    }
    // shot-noise events after all signal events (ON block, OFF block), ts[-1], unshuffled; a sub-group without one (most: a frame
    // has a couple per 256 pixels) costs one scalar test
    if (a.do_shot && any_shot) {
        const uint32_t son_tot = cT[0];
        uint32_t son_off = n_signal + pre[0], soff_off = n_signal + son_tot + pre[(size_t)ea.nwp];
        const float tl = tg(n - 1);
#pragma unroll
        for (int j = 0; j < GPX; ++j) {
            if ((so[j] | sf[j]) != 0ull) {
                float fx, fy;
                pixel_xy((uint32_t)(p0 + j * WAVE), fx, fy);
                if (cw[j] & CNT_SHOT_ON) store_row(son_off + (uint32_t)__popcll(so[j] & lt), tl, fx, fy, 1.0f);
                if (cw[j] & CNT_SHOT_OFF) store_row(soff_off + (uint32_t)__popcll(sf[j] & lt), tl, fx, fy, -1.0f);
                son_off += (uint32_t)__popcll(so[j]);
                soff_off += (uint32_t)__popcll(sf[j]);
            }
        }
    }
}

// ------------------------------------------------------------------ the event writer as a PULL (round 5)
// k_cemit pushes: a wave per 256-pixel group finds the group's events and scatters their 16-byte rows through the keyed bijection --
// partial lines (1.5x the row bytes written), and a per-group control flow that is bound by the scalar unit.  k_cpull turns it
// round: a thread per OUTPUT row.  Row j of iteration i holds the event of canonical index c = sigma_i^-1(j) (v2e_perm_invert;
// emulator.py:861-870: the iteration's ON block then its OFF block, shuffled together); c falls into the group g with
// pre[key][g] <= c' < pre[key][g + 1] (binary search over the key's exclusive prefix row, staged in LDS), and is that group's
// r-th pixel of the (iteration, polarity) block = the r-th set bit of the four ballots k_ctot left (v2e_nth_set_bit_256).  Rows
// are written in order: whole lines.  A frame's rows are split into `wpf` contiguous ranges, one workgroup each; the workgroups of
// one frame share blockIdx.x mod 8, i.e. an XCD and its L2 (prefix rows and masks of the frame are read from HBM once).
static_assert(GPX == 4, "k_cpull selects a pixel of a 256-pixel group (v2e_nth_set_bit_256)");
// TWO (large frames: a prefix row of 1280x720 is 14 KB): two levels -- every 16th entry in LDS, then the 16 entries (one line,
// from L2) searched in registers.
template <bool TWO>
__global__ __launch_bounds__(BLOCK) void k_cpull(KArgs a, CEmitArgs ea)
{
    extern __shared__ uint32_t s_pre[]; // [2][p2]: the ON and the OFF key's prefix row (TWO: its coarse level) of the iteration at hand
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int clip = blockIdx.y;
    const int blk = (int)blockIdx.x >> 3;
    const int z = (blk / ea.wpf) * 8 + ((int)blockIdx.x & 7), q = blk % ea.wpf;
    if (z >= ea.nE) return;
    const int fe = ea.f0 + z;
    const size_t zc = (size_t)z * ea.n_clips + clip;
    const CFrame *cf = ea.cf + zc;
    v2e_frame_rec *rec = ea.recs + (size_t)fe * ea.n_clips;
    const FrameCtl *c = ea.ctl + (size_t)fe * ea.n_clips + clip;
    const uint32_t *cT = ea.cT + zc * a.nkeys_cap, *ckb = ea.ckbase + zc * a.nkeys_cap;
    const uint32_t *pre = ea.cpre + zc * a.nkeys_cap * ea.nwp;
    const uint32_t *perm = ea.cperm + zc * a.max_iters * 8;
    const uint32_t *cmask = ea.cmask + zc * a.nkeys_cap * ea.nwp * (2 * GPX);
    const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
    const uint32_t off_lo = (uint32_t)ea.off_in[clip], off_hi = (uint32_t)(ea.off_in[clip] >> 32);
    const uint32_t nj = lane < z ? ea.cf[(size_t)lane * ea.n_clips + clip].n_events : 0u; // lane j: frame j of the batch (E <= 64)
    const int M = __builtin_amdgcn_readfirstlane(cf->M);
    const uint32_t n_signal = (uint32_t)__builtin_amdgcn_readfirstlane((int)cf->n_signal);
    const uint32_t n_events = (uint32_t)__builtin_amdgcn_readfirstlane((int)cf->n_events);
    const uint32_t disc = (uint32_t)__builtin_amdgcn_readfirstlane((int)cf->discarded);
    const FrameTab ftb(c, lane);
    const unsigned long long ev0 = ((unsigned long long)off_hi << 32 | off_lo) + (unsigned long long)wave_sum_u32(nj & 0xFFFFFFu) +
                                   ((unsigned long long)wave_sum_u32(nj >> 24) << 24); // + the events of the batch's earlier frames
    if (q == 0 && tid == 0) {
        rec[clip].ev_offset = ev0;
        if (ea.coff_in_cemit && z == ea.nE - 1) ea.off_out[clip] = ev0 + n_events; // else k_coff's
        if (ev0 + n_events > ea.cap || n_events > 0x7FFFFFFu) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
    }
    if (disc) return;
    const int n = M > 0 ? M : 1;
    bool use_refr;
    const TsGen tg = frame_tsgen(a, c, ftb, n, use_refr); // (the refractory filter is in k_ctot's masks already)
    const uint32_t ev0_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ev0), ev0_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ev0 >> 32));
    const unsigned long long ev0u = ((unsigned long long)ev0_hi << 32) | ev0_lo;
    const unsigned long long room = ev0u < ea.cap ? ea.cap - ev0u : 0ull;
    unsigned long long frows64 = room < (unsigned long long)n_events ? room : (unsigned long long)n_events; // rows that fit below the capacity
    if (frows64 > 0x7FFFFFFull) frows64 = 0x7FFFFFFull;
    const uint32_t frows = (uint32_t)frows64;
    const uint32_t per = ((frows + (uint32_t)ea.wpf - 1u) / (uint32_t)ea.wpf + (uint32_t)WAVE - 1u) & ~(uint32_t)(WAVE - 1);
    const uint32_t lo = (uint32_t)q * per, hi = min(lo + per, frows);
    if (lo >= hi) return;
    const __amdgpu_buffer_rsrc_t ev_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(ea.events + (size_t)clip * ea.cap + ev0u), 0, (int)(frows * 16u), 0x00020000);
    const float rcpW = 1.0f / (float)a.W;
    const int P2 = ea.p2;
    bool staged = false;
    for (int i = 0; i <= M; ++i) { // i == M: the shot-noise rows (ON block, OFF block, unshuffled, ts[-1]) behind the signal rows
        if (i == M && !a.do_shot) break;
        const int key0 = i < M ? 2 + 2 * i : 0;
        const uint32_t base = i < M ? ckb[key0] : n_signal;
        const uint32_t T_on = cT[key0], T_off = cT[key0 + 1];
        const uint32_t r_lo = max(lo, base), r_hi = min(hi, base + T_on + T_off);
        if (r_lo >= r_hi) continue; // (uniform: none of this iteration's rows in the workgroup's range)
        if (staged) __syncthreads(); // the readers of the rows staged before
        if (TWO) {
            const int nc16 = ea.nwp / 16;
            const uint32_t *pre16 = ea.cpre16 + (zc * a.nkeys_cap + key0) * nc16;
            for (int k = tid; k < 2 * P2; k += BLOCK) {
                const int kk = k & (P2 - 1);
                s_pre[k] = kk < nc16 ? pre16[(size_t)(k >= P2 ? nc16 : 0) + kk] : 0xFFFFFFFFu;
            }
        } else { // four entries per load (nwp is a multiple of 16; entries nwaves .. nwp - 1 hold the key's total: never <= cp)
            for (int k = tid * 4; k < 2 * P2; k += BLOCK * 4) {
                const int kk = k & (P2 - 1);
                uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                if (kk < ea.nwp) v = *(const uint4 *)(pre + (size_t)(key0 + (k >= P2 ? 1 : 0)) * ea.nwp + kk);
                *(uint4 *)(s_pre + k) = v;
            }
        }
        __syncthreads();
        staged = true;
        v2e_perm_t pm;
        const bool sh_i = shuf && i < M;
        if (sh_i) {
            const uint32_t *pp = perm + (size_t)i * 8;
            pm.k[0] = pp[0]; pm.k[1] = pp[1]; pm.k[2] = pp[2]; pm.k[3] = pp[3];
            pm.sh = pp[4]; pm.a = pp[5]; pm.amask = 0u; pm.n = pp[7];
            pm.rmask = (1u << pm.sh) - 1u;
        }
        const float t = tg(i < M ? i : n - 1);
        for (uint32_t row = r_lo + (uint32_t)tid; row < r_hi; row += BLOCK) {
            const uint32_t jj = row - base;
            const uint32_t ci = sh_i ? v2e_perm_invert(&pm, jj) : jj;
            const bool eneg = ci >= T_on;
            const uint32_t cp = eneg ? ci - T_on : ci;
            const uint32_t *tab = s_pre + (eneg ? P2 : 0);
            uint32_t g = 0, gbase = 0; // the last group whose prefix is <= cp (an empty group shares its prefix with the next one)
            if (!TWO) {
                for (int st = P2 >> 1; st >= 1; st >>= 1) {
                    const uint32_t v = tab[g + st];
                    if (v <= cp) { g += st; gbase = v; }
                }
            } else {
                for (int st = P2 >> 1; st >= 1; st >>= 1) {
                    const uint32_t v = tab[g + st];
                    if (v <= cp) g += st;
                }
                // the line of 16 entries whose first one is <= cp: four halving steps in registers
                const uint4 *lp = (const uint4 *)(pre + (size_t)(key0 + (eneg ? 1 : 0)) * ea.nwp + g * 16u);
                const uint4 q0 = lp[0], q1 = lp[1], q2 = lp[2], q3 = lp[3];
                const bool h8 = q2.x <= cp;
                const uint4 a0 = h8 ? q2 : q0, a1 = h8 ? q3 : q1;
                const bool h4 = a1.x <= cp;
                const uint4 b = h4 ? a1 : a0;
                const bool h2 = b.z <= cp;
                const uint32_t c0 = h2 ? b.z : b.x, c1 = h2 ? b.w : b.y;
                const bool h1 = c1 <= cp;
                gbase = h1 ? c1 : c0;
                g = g * 16u + (h8 ? 8u : 0u) + (h4 ? 4u : 0u) + (h2 ? 2u : 0u) + (h1 ? 1u : 0u);
            }
            const uint4 *mp = (const uint4 *)(cmask + ((size_t)(key0 + (eneg ? 1 : 0)) * ea.nwp + g) * (2 * GPX));
            const uint4 ma = mp[0], mb = mp[1];
            const uint32_t bit = v2e_nth_set_bit_256((unsigned long long)ma.x | ((unsigned long long)ma.y << 32), (unsigned long long)ma.z | ((unsigned long long)ma.w << 32),
                                                     (unsigned long long)mb.x | ((unsigned long long)mb.y << 32), (unsigned long long)mb.z | ((unsigned long long)mb.w << 32),
                                                     cp - gbase);
            const uint32_t p = g * (uint32_t)GROUP_PX + bit;
            uint32_t qy; // y = p / W, x = p % W, exactly
            if (a.npx < (1 << 24)) { // float32 holds p exactly: the quotient estimate is the true one or one off
                qy = (uint32_t)((float)p * rcpW);
                if (qy * (uint32_t)a.W > p) --qy;
                else if ((qy + 1u) * (uint32_t)a.W <= p) ++qy;
            } else {
                qy = p / (uint32_t)a.W;
            }
            const v2e_f4 v = {t, (float)(p - qy * (uint32_t)a.W), (float)qy, eneg ? -1.0f : 1.0f};
            __builtin_amdgcn_raw_buffer_store_b128(v, ev_rsrc, (int)(row * 16u), 0, 19); // sc0 nt sc1: written through, non-temporal (see store_event_wt)
        }
    }
}
