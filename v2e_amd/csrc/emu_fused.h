// emu_fused.h -- fused device-resident pipeline of the DVS pixel model (included by emu.hip).
//
// Round-1 profiling (profiles/r01_*) showed that at 346x260 a frame is one wave per SIMD, so
// what a kernel costs is its chain of DEPENDENT memory round trips plus launch latency, not its
// arithmetic.  This pipeline therefore
//   * needs only the two unavoidable grid-wide dependencies per frame (the global max M and the
//     per-(iteration,polarity) totals), both read back through plain memory at the next launch:
//       k_main(f) = emit(f-1) + count(f) for the same pixel in one thread,
//       k_refr(f) = refractory re-count of frame f's workgroup totals, a no-op launch unless the
//                   rule is active for that frame (device-side test on M);
//   * issues every global load of both phases at the top of k_main, before any barrier;
//   * has no atomics: each workgroup publishes its own max and key totals, consumers reduce
//     them (M = max over workgroups; row prefix = sum over earlier workgroups, lane = key);
//   * reads per-frame scalars from the host-filled FrameCtl and lin_log/inten01 of uint8 frames
//     from a 256-entry table staged in LDS.
//
// Keys: 0 = shot ON, 1 = shot OFF, 2+2i = iteration i ON, 3+2i = iteration i OFF.
// Scratch is double-buffered by frame parity (cnt, gtT, gmax): k_main reads frame f-1's while
// writing frame f's.  Invariant: gtT[k][g] == 0 for k >= 2 + 2*gmax[g] (and for g >= ngroups).
#pragma once

struct FusedArgs {
    const void *frame;           // frame f (count phase)
    const FrameCtl *ctl_c;       // times of frame f
    const FrameCtl *ctl_e;       // times of frame f-1
    v2e_frame_rec *rec_e;        // record of frame f-1 (written here)
    const v2e_frame_rec *rec_ee; // record of frame f-2 (event offset chain) or nullptr
    const uint32_t *fidx_base;
    uint32_t fidx_c, fidx_e;
    int do_emit, do_count;
    int par_c, par_e;
    int ngroups;
    uint32_t *cnt2[2];
    uint16_t *gtT2[2];  // [n_clips][nkeys_cap][ngp] per-workgroup key totals (<= 256 each), key-major
    int ngp;            // ngroups rounded up to 512 (one 16-byte load per lane covers 512 workgroups)
    int *gmax2[2];      // [n_clips][ngroups], true (unclamped) per-workgroup max count
    float4 *events;
    unsigned long long cap;
    unsigned long long *dbg; // dev tool: per-workgroup s_memrealtime stamps [ngroups][16] or nullptr
    // In-kernel refractory fix-up (small grids whose workgroups are all co-resident): instead of the
    // k_refr launch, frames on which the rule is active re-count inside k_main and synchronise the
    // clip's workgroups with one counter per (frame, clip).  nullptr: k_refr launches are used.
    unsigned *bar;      // [n_frames_of_run][n_clips], zeroed at run start
    // Large grids (thousands of workgroups): a k_scan2 launch turns the per-workgroup key totals
    // into exclusive prefixes once, instead of every workgroup re-reducing all of them.
    const uint32_t *pre32; // [n_clips][nkeys_cap][ngp] or nullptr
    const uint32_t *tot32; // [n_clips][nkeys_cap]
    // Event records (dynamic LDS, capw 4-byte records per wave; 0: off): in the first 64-key chunk every passing
    // (lane, iteration) leaves a record and the wave then writes its events one per lane, instead of iteration by
    // iteration with the shuffle arithmetic running for the few lanes that fire.
    int capw;
};

#define V2E_STAMP(i) do { if (fa.dbg && tid == 0) fa.dbg[(size_t)g * 16 + (i)] = wall_clock64(); } while (0)

constexpr int KPRE = 12; // keys of chunk 0 fetched before M is known (covers M <= 5)
constexpr int KPW = KPRE / (BLOCK / WAVE); // of those, keys per wave

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

// max over all workgroups' published maxima (block-uniform result). One barrier pair.
__device__ __forceinline__ int block_max_of_groups(const int *__restrict__ gm, int ngroups, int *s_red, int tid, int lane, int wave)
{
    int m = 0;
    for (int k = tid; k < ngroups; k += BLOCK) m = max(m, gm[k]);
    m = wave_max_i32(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(m);
}

__device__ __forceinline__ int block_max_finish(int m, int *s_red, int lane, int wave)
{
    m = wave_max_i32(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(m); // make the uniformity visible: loops over M stay scalar
}

// One wave, one key: total over all workgroups and over the workgroups before g.  A 16-byte load
// per lane covers 512 workgroups, so this is one load instruction per key for sensors up to
// 512*256 pixels.  (tot, pre) are returned wave-uniform.
__device__ __forceinline__ void key_totals(const uint16_t *__restrict__ row, int ngp, int g, int lane, uint32_t &tot_o,
                                           uint32_t &pre_o)
{
    uint32_t tot = 0, pre = 0;
    for (int c0 = 0; c0 < ngp; c0 += 512) {
        const int gb = c0 + lane * 8;
        const uint4 v = *(const uint4 *)(row + gb);
        // two u16 counts per dword, each <= 256: add the four dwords lane-wise, then fold halves
        const uint32_t s2 = v.x + v.y + v.z + v.w;
        const uint32_t lane_tot = (s2 & 0xFFFFu) + (s2 >> 16);
        tot += lane_tot;
        if (gb + 8 <= g) {
            pre += lane_tot;
        } else if (gb < g) { // the one lane whose 8 workgroups straddle g
            const uint32_t w[8] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16, v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
#pragma unroll
            for (int j = 0; j < 8; ++j) pre += (gb + j < g) ? w[j] : 0u;
        }
    }
    tot_o = wave_sum_u32(tot);
    pre_o = wave_sum_u32(pre);
}

// the reduction half of key_totals for one already-loaded 16-byte slice (ngp == 512: one slice per key)
__device__ __forceinline__ void key_totals_loaded(const uint4 v, int g, int lane, uint32_t &tot_o, uint32_t &pre_o)
{
    const int gb = lane * 8;
    const uint32_t s2 = v.x + v.y + v.z + v.w;
    const uint32_t lane_tot = (s2 & 0xFFFFu) + (s2 >> 16);
    uint32_t pre = 0;
    if (gb + 8 <= g) {
        pre = lane_tot;
    } else if (gb < g) {
        const uint32_t w[8] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16, v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
#pragma unroll
        for (int j = 0; j < 8; ++j) pre += (gb + j < g) ? w[j] : 0u;
    }
    tot_o = wave_sum_u32(lane_tot);
    pre_o = wave_sum_u32(pre);
}

// Per-workgroup key totals of this frame's candidates.  REFR = false: every candidate counts
// (speculative, right whenever the refractory rule is off); REFR = true: apply the rule against
// a private copy of ts_mem.  Called by all 256 threads (barriers inside).  Keeps the row invariant.
template <bool REFR>
__device__ __forceinline__ void group_key_totals(const KArgs &a, uint32_t cw, float tsm, const TsGen &tg, int gmax,
                                                 uint16_t *__restrict__ gcol, int ngp, uint32_t (*s_wcnt)[WAVE], int lane,
                                                 int wave, int gmax_old, uint32_t *wg_events = nullptr)
{
    // gcol = &gtT[clip][0][g]; key k of this workgroup lives at gcol[k * ngp]
    for (int k = 2 + 2 * gmax + (int)threadIdx.x; k < 2 + 2 * gmax_old; k += BLOCK) gcol[(size_t)k * ngp] = 0;
    const int mag = (int)(cw & CNT_MASK);
    const bool neg = (cw & CNT_NEG) != 0;
    const int nk = 2 + 2 * gmax;
    bool alive = true;
    for (int kb = 0; kb < nk; kb += WAVE) {
        uint32_t mine = 0;
        const int i_lo = kb == 0 ? 0 : (kb - 2) / 2;
        const int i_hi = (kb + WAVE - 2) / 2; // exclusive
        for (int i = i_lo; i < i_hi && i < gmax && alive; ++i) {
            const bool cand = mag > i;
            if (__ballot(cand) == 0ull) { alive = false; break; }
            bool pass = cand;
            if (REFR) {
                const float t = tg(i);
                const float pt = (cand ? 1.0f : 0.0f) * t - tsm;
                pass = pt > a.refr_f;
                if (pass) tsm = t;
            }
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
        if (wave == 0 && kb + lane < nk) {
            const uint32_t v = s_wcnt[0][lane] + s_wcnt[1][lane] + s_wcnt[2][lane] + s_wcnt[3][lane];
            gcol[(size_t)(kb + lane) * ngp] = (uint16_t)v;
            if (wg_events) *wg_events += v; // wave 0, lane-wise partial sums of this workgroup's events
        }
        __syncthreads();
    }
}

// Grid-wide rendezvous of the `target` workgroups of one clip (MI355X guide, Guideline 16 hand-off
// in its counter form): every wave drains its stores, one lane does the agent-scope release, the
// relaxed arrive, a relaxed bounded poll, and the agent-scope acquire; __syncthreads() extends it
// to the workgroup.  Requires every workgroup of the grid to be resident (checked by the host).
__device__ __forceinline__ bool clip_barrier(unsigned *ctr, unsigned target)
{
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 2000000u) { ok = 0; break; } // bounded: never hang the GPU
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <typename R, typename FT>
__global__ __launch_bounds__(BLOCK) void k_main(KArgs a, FusedArgs fa)
{
    extern __shared__ uint32_t s_dynrec[]; // [BLOCK / WAVE][fa.capw] event records
    __shared__ uint32_t s_T[WAVE], s_P[WAVE]; // per key of the current 64-key chunk: total / prefix over workgroups
    __shared__ uint32_t s_wcnt[BLOCK / WAVE][WAVE];
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ float s_lutL[256];
    __shared__ double s_lutI[256];
    constexpr bool U8 = sizeof(FT) == 1;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int clip = blockIdx.y, g = blockIdx.x;
    const int p = g * BLOCK + tid;
    const bool valid = p < a.npx;
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const uint32_t fbase = fa.fidx_base ? *fa.fidx_base : 0u;

    V2E_STAMP(0);
    // ------------------------------------------------------------ all independent loads first
    R b = (R)0, lp_old = (R)0;
    float thp = 1.f, thn = 1.f, nr = 0.f, tsm = 0.f;
    uint32_t cw_e = 0;
    FT px = (FT)0;
    if (valid) {
        b = ((R *)a.base)[sp];
        thp = a.pos_thres[sp];
        thn = a.neg_thres[sp];
        if (a.has_cutoff || (fa.do_emit && a.do_shot)) lp_old = ((R *)a.lp)[sp];
        if (a.do_leak && fa.do_count) nr = a.noise_rate[sp];
        if (a.has_refr && fa.do_emit) tsm = a.ts_mem[sp];
        if (fa.do_emit) cw_e = fa.cnt2[fa.par_e][sp];
        if (fa.do_count) px = ((const FT *)fa.frame)[(size_t)clip * a.npx + p];
    }
    if (U8 && fa.do_count) {
        s_lutL[tid] = a.lut_L[tid];
        s_lutI[tid] = a.lut_I[tid];
    }
    // key rows of this wave's KPW keys and the workgroup maxima: issue the loads now, reduce later
    const uint16_t *gt = fa.gtT2[fa.par_e] + (size_t)clip * a.nkeys_cap * fa.ngp;
    const bool krow_fast = fa.do_emit && !fa.pre32 && fa.ngp == 512;
    uint4 kv[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) kv[j] = make_uint4(0u, 0u, 0u, 0u);
    int gm_part = 0;
    if (fa.do_emit) {
        if (krow_fast) {
#pragma unroll
            for (int j = 0; j < KPW; ++j) kv[j] = *(const uint4 *)(gt + (size_t)(wave + (BLOCK / WAVE) * j) * 512 + lane * 8);
        }
        const int *gmv = fa.gmax2[fa.par_e] + (size_t)clip * fa.ngroups;
        for (int k = tid; k < fa.ngroups; k += BLOCK) gm_part = max(gm_part, gmv[k]);
    }
    // keep every load above issued here (the scheduler otherwise sinks them next to their first use,
    // which puts a full memory round trip back on the critical path); waits stay at the uses
    __builtin_amdgcn_sched_barrier(0);
    // ---- arithmetic that depends on no memory, done while those loads are in flight
    float rng_r = 0.f, rng_u = 0.f; // count(f): leak normal / shot uniform of this pixel
    if (fa.do_count && valid && ((a.do_leak && a.jit_f != 0.f) || a.do_shot))
        v2e_draw_frame(a.seed, (uint32_t)clip, fbase + fa.fidx_c, (uint32_t)p, &rng_r, &rng_u);
    uint32_t pk[4] = {0, 0, 0, 0};  // emit(f-1): shuffle round keys of iteration `lane` (first chunk)
    float tab_start = 0.f, tab_step = 0.f, tab_end = 0.f;
    uint32_t refr_mask = 0;
    if (fa.do_emit) {
        if (a.shuffle && a.rng_mode == V2E_RNG_PHILOX) v2e_perm_keys(a.seed, (uint32_t)clip, fbase + fa.fidx_e, (uint32_t)lane, pk);
        const FrameCtl *ce = fa.ctl_e + clip;
        tab_start = ce->ts_start[lane & 31];
        tab_step = ce->ts_stepf[lane & 31];
        refr_mask = ce->refr_mask;
        tab_end = ce->ts_end;
    }
    unsigned long long ev0 = 0;
    int M = 0, gmax_own_e = 0;
    if (fa.do_emit) {
        if (fa.bar) gmax_own_e = __builtin_amdgcn_readfirstlane(fa.gmax2[fa.par_e][(size_t)clip * fa.ngroups + g]);
        if (fa.pre32) { // prefixes already computed by k_scan2: two loads per key, lane = key
            if (wave == 0) {
                s_T[lane] = fa.tot32[(size_t)clip * a.nkeys_cap + lane];
                s_P[lane] = fa.pre32[((size_t)clip * a.nkeys_cap + lane) * fa.ngp + g];
            }
        } else if (krow_fast) {
#pragma unroll
            for (int j = 0; j < KPW; ++j) { // the first KPRE keys, loaded above before M is known
                uint32_t t, q;
                key_totals_loaded(kv[j], g, lane, t, q);
                if (lane == 0) { s_T[wave + (BLOCK / WAVE) * j] = t; s_P[wave + (BLOCK / WAVE) * j] = q; }
            }
        } else {
            for (int k = wave; k < KPRE && k < a.nkeys_cap; k += BLOCK / WAVE) {
                uint32_t t, q;
                key_totals(gt + (size_t)k * fa.ngp, fa.ngp, g, lane, t, q);
                if (lane == 0) { s_T[k] = t; s_P[k] = q; }
            }
        }
        if (fa.rec_ee) ev0 = fa.rec_ee[clip].ev_offset + fa.rec_ee[clip].n_events;
        V2E_STAMP(1);
        M = block_max_finish(gm_part, s_red, lane, wave);
        V2E_STAMP(2);
    } else if (U8 && fa.do_count) {
        __syncthreads(); // LUT visible
    }
    bool b_dirty = false;

    // ------------------------------------------------------------ emit(f-1)
    if (fa.do_emit) {
        v2e_frame_rec *rec = fa.rec_e;
        if (M > a.max_iters) {
            if (g == 0 && tid == 0) {
                rec[clip].max_events = M;
                rec[clip].flags |= V2E_FLAG_ITERS_CLAMPED;
                rec[clip].ev_offset = ev0;
            }
        } else {
            const uint32_t frame_idx = fbase + fa.fidx_e;
            const int n = M > 0 ? M : 1;
            const FrameCtl *ce = fa.ctl_e + clip;
            TsGen tg(0.f, 0.f, 0.f, n);
            bool use_refr;
            if (n <= 32) { // host-filled tables: no float64 division on the critical path
                tg = TsGen(__uint_as_float(lane_value(__float_as_uint(tab_start), n - 1)), tab_end,
                           __uint_as_float(lane_value(__float_as_uint(tab_step), n - 1)), n);
                use_refr = a.has_refr && ((refr_mask >> (n - 1)) & 1u);
            } else {
                const FrameCtl c = *ce;
                tg = TsGen(c, n, nullptr);
                use_refr = a.has_refr && (a.refr > (c.t_frame - c.t_prev) / (double)n);
            }
            V2E_STAMP(10);
            const uint32_t cw = cw_e;
            const int mag = (int)(cw & CNT_MASK);
            const bool neg = (cw & CNT_NEG) != 0;
            if (use_refr && fa.bar) {
                // rare path: the totals published by count(f-1) ignore the refractory rule; re-count this
                // workgroup with it, publish, wait for the clip's other workgroups, re-read the totals
                const int gm = min(gmax_own_e, a.max_iters);
                uint16_t *gcol_e = fa.gtT2[fa.par_e] + (size_t)clip * a.nkeys_cap * fa.ngp + g;
                group_key_totals<true>(a, cw, tsm, tg, gm, gcol_e, fa.ngp, s_wcnt, lane, wave, gm);
                const bool ok = clip_barrier(fa.bar + (size_t)fa.fidx_e * gridDim.y + clip, (unsigned)fa.ngroups);
                if (!ok && tid == 0) atomicOr(&rec[clip].flags, V2E_FLAG_SYNC_TIMEOUT);
                for (int k = wave; k < KPRE && k < a.nkeys_cap; k += BLOCK / WAVE) {
                    uint32_t t, q;
                    key_totals(gt + (size_t)k * fa.ngp, fa.ngp, g, lane, t, q);
                    if (lane == 0) { s_T[k] = t; s_P[k] = q; }
                }
            }
            float4 *ev = fa.events + (size_t)clip * fa.cap;
            const unsigned long long lt = (1ull << lane) - 1ull;
            const float fx = (float)(p % a.W), fy = (float)(p / a.W);
            const bool shuf = (a.rng_mode == V2E_RNG_PHILOX) && a.shuffle;
            const int nk = 2 + 2 * M;
            uint32_t carry = 0, sum_on = 0, sum_off = 0;
            uint32_t son_tot = 0, soff_tot = 0, son_off = 0, soff_off = 0;
            uint32_t nrec = 0; // event records of this wave (first chunk)
            uint32_t *rec_w = s_dynrec + (size_t)wave * fa.capw;
            int fcount = 0;
            bool dropped = false, alive = true;
            V2E_STAMP(11);
            for (int kb = 0; kb < nk; kb += WAVE) {
                const int key = kb + lane;
                // totals over all workgroups / over earlier workgroups for the keys not fetched yet
                if (fa.pre32) {
                    if (kb > 0 && wave == 0 && key < nk) {
                        s_T[lane] = fa.tot32[(size_t)clip * a.nkeys_cap + key];
                        s_P[lane] = fa.pre32[((size_t)clip * a.nkeys_cap + key) * fa.ngp + g];
                    }
                } else {
                    for (int k = (kb == 0 ? KPRE : 0) + wave; k < WAVE && kb + k < nk; k += BLOCK / WAVE) {
                        uint32_t t, q;
                        key_totals(gt + (size_t)(kb + k) * fa.ngp, fa.ngp, g, lane, t, q);
                        if (lane == 0) { s_T[k] = t; s_P[k] = q; }
                    }
                }
                V2E_STAMP(12);
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
                    if (pass) { mymask |= 1u << (i - i_lo); ++fcount; }
                    const unsigned long long bo = __ballot(pass && !neg);
                    const unsigned long long bf = __ballot(pass && neg);
                    const int kl = 2 + 2 * i - kb;
                    if (lane == kl) mine = (uint32_t)__popcll(bo);
                    if (lane == kl + 1) mine = (uint32_t)__popcll(bf);
                    if (kb == 0 && fa.capw > 0) {
                        if (pass) {
                            const uint32_t rank = (uint32_t)__popcll((neg ? bf : bo) & lt);
                            const uint32_t pos = nrec + (uint32_t)__popcll((bo | bf) & lt);
                            if (pos < (uint32_t)fa.capw) rec_w[pos] = (uint32_t)lane | ((uint32_t)i << 6) | ((neg ? 1u : 0u) << 11) | (rank << 12);
                        }
                        nrec += (uint32_t)__popcll(bo | bf);
                    }
                }
                if (kb == 0) {
                    const unsigned long long so = __ballot((cw & CNT_SHOT_ON) != 0);
                    const unsigned long long sf = __ballot((cw & CNT_SHOT_OFF) != 0);
                    if (lane == 0) mine = (uint32_t)__popcll(so);
                    if (lane == 1) mine = (uint32_t)__popcll(sf);
                }
                s_wcnt[wave][lane] = mine;
                V2E_STAMP(3);
                __syncthreads();
                V2E_STAMP(4);
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
                if (kb == 0 && fa.capw > 0 && nrec <= (uint32_t)fa.capw) { // one event per lane from the wave's records
                    for (uint32_t e0 = 0; e0 < nrec; e0 += WAVE) {
                        const bool has = e0 + lane < nrec;
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
                            pm.k[0] = (uint32_t)__shfl((int)pk[0], i); pm.k[1] = (uint32_t)__shfl((int)pk[1], i);
                            pm.k[2] = (uint32_t)__shfl((int)pk[2], i); pm.k[3] = (uint32_t)__shfl((int)pk[3], i);
                            pm.sh = (uint32_t)__shfl((int)ps_sh, i); pm.a = (uint32_t)__shfl((int)ps_a, i);
                            pm.amask = (uint32_t)__shfl((int)ps_amask, i); pm.n = (uint32_t)__shfl((int)ps_n, i);
                            pm.rmask = (1u << pm.sh) - 1u;
                            if (has) cidx = v2e_perm_apply(&pm, cidx);
                        }
                        if (has) {
                            const unsigned long long row = ev0 + it_base + cidx;
                            if (row < fa.cap) ev[row] = make_float4(tg(i), ex, ey, eneg ? -1.0f : 1.0f);
                            else dropped = true;
                        }
                    }
                } else if (__ballot(mymask != 0u)) {
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
                            uint32_t cidx = neg ? tot_on + off_off + (uint32_t)__popcll(bf & lt)
                                                : off_on + (uint32_t)__popcll(bo & lt);
                            if (shuf) cidx = v2e_perm_apply(&pm, cidx);
                            const unsigned long long row = ev0 + it_base + cidx;
                            if (row < fa.cap) ev[row] = make_float4(tg(i), fx, fy, neg ? -1.0f : 1.0f);
                            else dropped = true;
                        }
                    }
                }
                carry += chunk_total;
                V2E_STAMP(5);
                __syncthreads();
            }
            V2E_STAMP(6);
            // shot-noise events after all signal events (ON block, OFF block), ts[-1], unshuffled
            if (a.do_shot) {
                const bool s_on = (cw & CNT_SHOT_ON) != 0, s_off = (cw & CNT_SHOT_OFF) != 0;
                const unsigned long long so = __ballot(s_on), sf = __ballot(s_off);
                if (so | sf) {
                    const float tl = tg(n - 1);
                    if (s_on) {
                        const unsigned long long row = ev0 + carry + son_off + (uint32_t)__popcll(so & lt);
                        if (row < fa.cap) ev[row] = make_float4(tl, fx, fy, 1.0f);
                        else dropped = true;
                    }
                    if (s_off) {
                        const unsigned long long row = ev0 + carry + son_tot + soff_off + (uint32_t)__popcll(sf & lt);
                        if (row < fa.cap) ev[row] = make_float4(tl, fx, fy, -1.0f);
                        else dropped = true;
                    }
                }
            }
            if (valid) { // emulator.py:936-942
                const bool shot = a.do_shot && (cw & (CNT_SHOT_ON | CNT_SHOT_OFF));
                if (fcount > 0 || shot) {
                    const float dp = (float)(neg ? 0 : fcount) * thp;
                    const float dn = (float)(neg ? fcount : 0) * thn;
                    b = b + (R)dp;
                    b = b - (R)dn;
                    if (shot) b = lp_old;
                    b_dirty = true;
                }
                if (use_refr && fcount > 0) a.ts_mem[sp] = tsm;
            }
            if (__ballot(dropped) != 0ull && lane == 0) atomicOr(&rec[clip].flags, V2E_FLAG_EVENTS_DROPPED);
            if (g == 0 && tid == 0) {
                rec[clip].max_events = M;
                rec[clip].n_signal = carry;
                rec[clip].n_events = carry + (a.do_shot ? son_tot + soff_tot : 0u);
                rec[clip].n_on = sum_on + (a.do_shot ? son_tot : 0u);
                rec[clip].n_off = sum_off + (a.do_shot ? soff_tot : 0u);
                rec[clip].ev_offset = ev0;
            }
        }
    }

    V2E_STAMP(7);
    // ------------------------------------------------------------ count(f)
    if (fa.do_count) {
        const FrameCtl c = fa.ctl_c[clip];
        const uint32_t frame_idx = fbase + fa.fidx_c;
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
            fa.cnt2[fa.par_c][sp] = cw;
            m = pc > nc ? pc : nc;
        }
        V2E_STAMP(8);
        m = wave_max_i32(m);
        int *gmp = fa.gmax2[fa.par_c] + (size_t)clip * fa.ngroups + g;
        const int gmax_old_raw = __builtin_amdgcn_readfirstlane(*gmp); // what this row was last written with (two frames ago)
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        const int gmax_raw = __builtin_amdgcn_readfirstlane(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
        if (tid == 0) *gmp = gmax_raw;
        const int gmax = min(gmax_raw, a.max_iters), gmax_old = min(gmax_old_raw, a.max_iters);
        uint16_t *gcol = fa.gtT2[fa.par_c] + (size_t)clip * a.nkeys_cap * fa.ngp + g;
        const TsGen tg_unused(c, 1, nullptr);
        group_key_totals<false>(a, cw, 0.f, tg_unused, gmax, gcol, fa.ngp, s_wcnt, lane, wave, gmax_old);
        V2E_STAMP(9);
    } else if (valid && b_dirty) {
        ((R *)a.base)[sp] = b;
    }
}

// Refractory re-count of frame f's workgroup totals; every workgroup exits at once unless the
// rule is active for this frame (emulator.py:830: refractory_period_s > ts_step).
__global__ __launch_bounds__(BLOCK) void k_refr(KArgs a, const FrameCtl *__restrict__ ctl, const uint32_t *__restrict__ cnt,
                                                uint16_t *__restrict__ gtT, int ngp, const int *__restrict__ gmaxv, int ngroups)
{
    __shared__ uint32_t s_wcnt[BLOCK / WAVE][WAVE];
    __shared__ int s_red[BLOCK / WAVE];
    const int clip = blockIdx.y, g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int p = g * BLOCK + tid;
    const size_t sp = (size_t)clip * a.npx_pad + p;
    const bool valid = p < a.npx;
    const int *gm = gmaxv + (size_t)clip * ngroups;
    const int gmax = __builtin_amdgcn_readfirstlane(min(gm[g], a.max_iters));
    const uint32_t cw = valid ? cnt[sp] : 0u; // issued before M is known: one round trip
    const float tsm = valid ? a.ts_mem[sp] : 0.f;
    const int M = block_max_of_groups(gm, ngroups, s_red, tid, lane, wave);
    if (M <= 0 || M > a.max_iters) return;
    const FrameCtl c = ctl[clip];
    if (!(a.refr > (c.t_frame - c.t_prev) / (double)M)) return;
    if (gmax == 0) return;
    const TsGen tg(c, M, nullptr);
    group_key_totals<true>(a, cw, tsm, tg, gmax, gtT + (size_t)clip * a.nkeys_cap * ngp + g, ngp, s_wcnt, lane, wave, gmax);
}

// Large grids: exclusive prefix over workgroups of every key row (u16 counts -> u32 prefixes) and
// the row totals.  One workgroup per key (grid-stride over keys < 2 + 2M).
__device__ __forceinline__ void scan2_body(const KArgs &a, const uint16_t *__restrict__ gtT, int ngp, const int *__restrict__ gmaxv,
                                           int ngroups, uint32_t *__restrict__ pre32, uint32_t *__restrict__ tot32)
{
    __shared__ int s_red[BLOCK / WAVE];
    __shared__ uint32_t s_w[BLOCK / WAVE];
    const int clip = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int M = block_max_of_groups(gmaxv + (size_t)clip * ngroups, ngroups, s_red, tid, lane, wave);
    if (M > a.max_iters) return;
    const int nk = 2 + 2 * M;
    const int ipt = ngp / BLOCK; // ngp is a multiple of 512
    for (int key = blockIdx.x; key < nk; key += gridDim.x) {
        const uint16_t *row = gtT + ((size_t)clip * a.nkeys_cap + key) * ngp + (size_t)tid * ipt;
        uint32_t *orow = pre32 + ((size_t)clip * a.nkeys_cap + key) * ngp + (size_t)tid * ipt;
        uint32_t s = 0;
        for (int j = 0; j < ipt; ++j) s += row[j];
        const uint32_t ex = wave_excl_scan_u32(s, lane);
        if (lane == WAVE - 1) s_w[wave] = ex + s;
        __syncthreads();
        uint32_t base = ex, total = 0;
#pragma unroll
        for (int q = 0; q < BLOCK / WAVE; ++q) {
            if (q < wave) base += s_w[q];
            total += s_w[q];
        }
        for (int j = 0; j < ipt; ++j) {
            orow[j] = base;
            base += row[j];
        }
        if (tid == 0) tot32[(size_t)clip * a.nkeys_cap + key] = total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void k_scan2(KArgs a, const uint16_t *__restrict__ gtT, int ngp, const int *__restrict__ gmaxv,
                                                 int ngroups, uint32_t *__restrict__ pre32, uint32_t *__restrict__ tot32)
{
    scan2_body(a, gtT, ngp, gmaxv, ngroups, pre32, tot32);
}
