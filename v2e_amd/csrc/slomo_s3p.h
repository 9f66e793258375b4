// slomo_s3p.h -- the 3x3 split-bf16 convolution as ONE software-pipelined wave per SIMD (included by slomo.hip after
// slomo_s3.h, whose operand layouts, weight packing and arithmetic it shares: the same six piece products in the same
// order per (tap, 16-channel chunk), chunks and taps walked in the same order, so the output is bit-identical to
// k_conv_s3's).
//
// Why: k_conv_s3's 32-channel x 64-pixel register tile reads 9 LDS operands per 12 multiplies -- with every SIMD
// multiplying that is 96 B/clk of the CU's 128 B/clk of LDS bandwidth, before the staging stores; the matrix pipe waits on
// LDS.  A 64 x 64 register tile reads 12 operands per 24 multiplies (half the bytes per multiply), but needs four
// accumulator chains per wave, and four chains x TWO waves per SIMD is the one shape the matrix pipe runs slowly
// (scripts/ubench_mfma.hip: 66 % of peak on zeros where 4 x 1, 2 x 2 and 8 x 1 reach 98 %; profiles/r03_mfma_bare.txt).
// So: one 4-wave workgroup per CU (one wave per SIMD), and since no second wave is there to multiply while this one
// stages, the staging of what comes next is interleaved with the multiplies of the wave itself:
//   * LDS holds TWO patches (16 input channels x (TH+2) x (TW+2) pixels x 3 pieces, 32 KB each) and TWO kernel-row weight
//     blocks (3 taps x 16 channels x 64 output channels x 3 pieces, 18 KB each): 101 KB;
//   * a step = one kernel row of one chunk = 3 taps x 6 piece products x 4 tiles = 72 multiplies; while it runs, the wave
//     stores the next step's weights (fetched two steps ago) and a third of the next chunk's patch (fetched a chunk ago,
//     split into bf16 pieces here), and re-issues the global loads for three steps / two chunks further on;
//   * ONE barrier per step (three per chunk): what was written during step s is read from step s + 1 on, what step s - 1
//     read is overwritten during step s;
//   * the order of issue is fixed in the source: after every piece product (4 multiplies = 128 matrix-pipe cycles) comes a
//     slot of side work (two operand reads of the next tap, one weight store + reload, a quarter of the splitting, ...)
//     fenced by sched_barrier, so the compiler neither hoists the side work into one block nor sinks it behind the
//     multiplies.
// Shapes it takes: 3x3, cout % 64 == 0, cin % 32 == 0 (an even number of chunks: the body is unrolled over a chunk
// pair so that every buffer and register-set index is a compile-time constant), plain f32 NCHW input (one or two concat
// sources with channel counts that are multiples of 16).  Everything else stays on k_conv_s3.

// DBG (the ablation of round 3: instantiate launch_conv_s3p<16, DBG> in conv_dispatch_s3p to repeat it -- the environment switch that
// selected it is gone; results are wrong by construction): 1 = no side work
// (what the multiplies, their operand reads and the barriers cost alone), bit 2 (4) = no global reloads, bit 3 (8) = no LDS
// stores, 16 = no splitting arithmetic, 32 = no operand reads of the next tap, 64 / 128 = no patch / weight reloads; 2 = no
// multiplies.  The per-step timeline below works in every mode.  Measured (profiles/r03_slomo_s3p_ablation.txt): the kernel is
// at the chip's POWER limit -- fewer clocks per step lower the shader clock (1.63-1.86 GHz of 2.4) and leave the step's wall
// time where it was; what saves time is what saves energy per multiply (LDS and global bytes), not issue slots.
__device__ unsigned long long g_s3p_timeline[2 * 512]; // dev: (shader clock, 100 MHz wall clock) at every step of workgroup 0
static int g_s3p_timeline_on = 0; // host: passed to the kernel as an argument (ConvArgs::tl_on)

template <int TW, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_conv_s3p(ConvArgs a)
{
    constexpr int CT = 2, PT = 2, WP = 4, NT = 256, KS = 3, PAD = 1;
    constexpr int NPX = WP * PT * 32;
    constexpr int TH = NPX / TW;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PP = PH * PW;
    constexpr int COT = CT * 32;
    constexpr int NPI = (2 * PP + NT - 1) / NT; // patch items per thread; item = 8 channels of one patch pixel
    constexpr int WU = 3 * 6 * COT;             // 16-byte weight units of one kernel row
    constexpr int NWU = (WU + NT - 1) / NT;
    static_assert(NPX % TW == 0 && NPI == 3 && NWU == 5, "the slot schedule below is written for 3 patch items and 5 weight units a thread");
    extern __shared__ u32x4 s3_smem[];
    u32x4 *sp = s3_smem;              // [2][3][2][PP]
    u32x4 *sw = s3_smem + 2 * 6 * PP; // [2][3 taps][3][2][COT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hsel = lane >> 5, l31 = lane & 31;
    int bx, cblk; // (see k_conv_s3: channel blocks of one pixel tile consecutive on one XCD)
    if (a.ncb > 0) {
        const int L = blockIdx.x, j = L >> 3;
        cblk = j % a.ncb;
        bx = (j / a.ncb) * 8 + (L & 7);
        if (bx >= a.n * a.tiles_x * a.tiles_y) return;
    } else {
        bx = blockIdx.x;
        cblk = blockIdx.y;
    }
    const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty_i = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int oy0 = ty_i * TH, ox0 = tx_i * TW;
    const int cobase = cblk * COT;
    const int hw = a.h * a.w_;
    const u32x4 *wsrc = (const u32x4 *)a.ws3;
    const int nchunks = a.cin >> 4;

    int bofs[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = (wave * PT + pt) * 32 + l31;
        bofs[pt] = hsel * PP + (m / TW) * PW + (m % TW);
    }
    const int aofs = hsel * COT + l31;
    int pinfo[NPI];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int i = tid + j * NT;
        int v = -1;
        if (i < 2 * PP) {
            const int cig = i / PP, r = i - cig * PP;
            const int py = r / PW, px = r - py * PW;
            const int gy = oy0 + py - PAD, gx = ox0 + px - PAD;
            if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w_) v = cig * 8 * hw + gy * a.w_ + gx;
        }
        pinfo[j] = v;
    }
    int woff[NWU];
#pragma unroll
    for (int j = 0; j < NWU; ++j) {
        const int u = tid + j * NT;
        const int uu = u < WU ? u : WU - 1;
        const int r = uu / COT, c = uu - r * COT;
        woff[j] = r * a.cout + cobase + c;
    }

    // byte offsets of this thread's loads from a wave-uniform base: 32-bit, so that a load is `global_load ... v_off, s[base]`
    // with no per-load 64-bit vector address arithmetic (which costs the wave's issue slot ~20 clocks a load)
    uint32_t pbyte[NPI], wbyte[NWU];
#pragma unroll
    for (int j = 0; j < NPI; ++j) pbyte[j] = (uint32_t)(pinfo[j] < 0 ? 0 : pinfo[j]) * 4u;
#pragma unroll
    for (int j = 0; j < NWU; ++j) wbyte[j] = (uint32_t)woff[j] * 16u;

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    float pv[2][NPI][8]; // two chunks of patch loads in flight
#ifndef V2E_S3P_WD
#define V2E_S3P_WD 3
#endif
    constexpr int WD = V2E_S3P_WD; // weight register sets = steps a weight load has to arrive (3 or 6: 6 % WD == 0)
    u32x4 wv[WD][NWU];

    // chunk index beyond the layer: clamped (the loads stay unconditional; what they fetch is staged into a buffer nobody reads)
    auto patch_base = [&](int chunk) -> const float * {
        const int cb = (chunk < nchunks ? chunk : nchunks - 1) << 4;
        const float *src;
        int cs, C;
        if (cb < a.c0) { src = a.x0; cs = cb; C = a.c0; }
        else { src = a.x1; cs = cb - a.c0; C = a.c1; }
        return src + ((size_t)n * C + cs) * hw;
    };
    auto load_item_half = [&](int set, int j, const float *base, int half) { // 4 of an item's 8 channels
        const int pi = pinfo[j];
        const float *q = base + (pi < 0 ? 0 : pi);
#pragma unroll
        for (int e = 4 * half; e < 4 * half + 4; ++e) pv[set][j][e] = q[(size_t)e * hw];
    };
    auto weight_base = [&](int step) -> const u32x4 * { // step = chunk * 3 + kernel row
        const int st = step < 3 * nchunks ? step : 3 * nchunks - 1;
        return wsrc + (size_t)st * 3 * 6 * a.cout; // [chunk][tap][6][cout]: a kernel row is 3 consecutive taps
    };
    auto stage_item_piece = [&](u32x4 *P, int j, const uint32_t (&q)[3][4], int p) {
        const int i = tid + j * NT;
        if (i < 2 * PP) P[p * 2 * PP + i] = u32x4{q[p][0], q[p][1], q[p][2], q[p][3]};
    };
    auto split_pair = [&](int set, int j, int e, uint32_t (&q)[3][4]) {
        const bool ok = pinfo[j] >= 0;
        split3_pair(ok ? pv[set][j][2 * e] : 0.f, ok ? pv[set][j][2 * e + 1] : 0.f, q[0][e], q[1][e], q[2][e]);
    };

    // ---- prologue: weights of steps 0 and 1, patches of chunks 0 and 1 in flight; step 0's operands into LDS
    {
#pragma unroll
        for (int d = 0; d < WD; ++d) {
            const u32x4 *wd = weight_base(d);
#pragma unroll
            for (int j = 0; j < NWU; ++j) wv[d][j] = wd[woff[j]];
        }
        const float *b0 = patch_base(0), *b1 = patch_base(1);
#pragma unroll
        for (int j = 0; j < NPI; ++j) { load_item_half(0, j, b0, 0); load_item_half(0, j, b0, 1); }
#pragma unroll
        for (int j = 0; j < NPI; ++j) { load_item_half(1, j, b1, 0); load_item_half(1, j, b1, 1); }
#pragma unroll
        for (int j = 0; j < NWU; ++j) {
            const int u = tid + j * NT;
            if (u < WU) sw[u] = wv[0][j];
        }
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            uint32_t q[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_pair(0, j, e, q);
#pragma unroll
            for (int p = 0; p < 3; ++p) stage_item_piece(sp, j, q, p);
        }
        const u32x4 *wn = weight_base(WD);
#pragma unroll
        for (int j = 0; j < NWU; ++j) wv[0][j] = wn[woff[j]];
        const float *b2 = patch_base(2);
#pragma unroll
        for (int j = 0; j < NPI; ++j) { load_item_half(0, j, b2, 0); load_item_half(0, j, b2, 1); }
    }

    bf16x8 av[2][3][CT], bv[2][3][PT];
    // One step.  S = position inside the chunk pair (0..5): chunk parity S / 3, kernel row S % 3, weight buffer S & 1.
    auto step = [&](auto S_tag, int pair) {
        constexpr int S = decltype(S_tag)::value;
        constexpr int cp = S / 3, g = S % 3, wp = S & 1;
        const int chunk = 2 * pair + cp, s = 6 * pair + S;
        const u32x4 *Pc = sp + cp * 6 * PP;
        u32x4 *Pn = sp + (cp ^ 1) * 6 * PP;
        const u32x4 *Wc = sw + wp * WU;
        u32x4 *Wn = sw + (wp ^ 1) * WU;
        constexpr int ws = (S + 1) % WD;               // the register set holding step s + 1's weights (6 % WD == 0)
        const u32x4 *wnext = weight_base(s + 1 + WD);  // refills it once stored
        const float *pnext = patch_base(chunk + 3);    // refills patch item g of the set stored during this step
        uint32_t q[3][4];
        __syncthreads();
        if (a.tl_on && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && s < 511) {
            g_s3p_timeline[2 * s] = clock64();
            g_s3p_timeline[2 * s + 1] = wall_clock64();
            g_s3p_timeline[2 * (s + 1)] = 0;
        }
        auto load_a1 = [&](int t, int buf, int p, int ct) { av[buf][p][ct] = __builtin_bit_cast(bf16x8, Wc[aofs + (t * 6 + p * 2) * COT + ct * 32]); };
        auto load_b1 = [&](int t, int buf, int p, int pt) { bv[buf][p][pt] = __builtin_bit_cast(bf16x8, Pc[bofs[pt] + p * 2 * PP + g * PW + t]); };
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { load_a1(0, 0, p, c2); load_b1(0, 0, p, c2); }
        }
        // The matrix pipe takes a multiply every 32 clocks and the wave issues in order, so side work only hides in the gap
        // after EACH multiply (a handful of instructions), not in a block after four of them.  Micro-slot m = the gap after
        // the m-th multiply of the step (72 of them; tap = m / 24):
        //   taps 0, 1, gaps 0-11   one operand read of the next tap each, in the order its products need them
        //   tap 0, gaps 12-23      the split of patch item g of the next chunk: 4 channel pairs x 3 stages
        //   tap 1, gaps 12-14      its three pieces stored;  gaps 15-22: its 8 registers reloaded (two chunks on)
        //   tap 2, gaps 0-9        the next step's 5 weight units stored and their registers reloaded (WD steps on)
        float ra = 0.f, rb = 0.f; // the split's running residuals
        auto micro = [&](auto M_tag) {
            constexpr int m = decltype(M_tag)::value;
            constexpr int t = m / 24, i = m % 24;
            if constexpr (DBG == 1) return;
            if constexpr (t < 2 && i < 12) {
                if constexpr (!(DBG & 32)) {
                    constexpr int nb = (t + 1) & 1, c2 = i & 1;
                    constexpr int piece_of[6] = {2, 0, 0, 2, 1, 1}; // A2 B0 A0 B2 A1 B1: what (2,0) (0,2) (1,1) need, in that order
                    if constexpr ((i / 2) % 2 == 0) load_a1(t + 1, nb, piece_of[i / 2], c2);
                    else load_b1(t + 1, nb, piece_of[i / 2], c2);
                }
            } else if constexpr (t == 0) {
                constexpr int e = (i - 12) / 3, st = (i - 12) % 3;
                if constexpr (DBG & 16) {
                    if constexpr (st == 0) { q[0][e] = __float_as_uint(pv[cp ^ 1][g][2 * e]); q[1][e] = __float_as_uint(pv[cp ^ 1][g][2 * e + 1]); q[2][e] = q[0][e]; }
                } else if constexpr (st == 0) { // p0 = bf16(x)
                    const bool ok = pinfo[g] >= 0;
                    ra = ok ? pv[cp ^ 1][g][2 * e] : 0.f;
                    rb = ok ? pv[cp ^ 1][g][2 * e + 1] : 0.f;
                    const f32x2 v0 = {ra, rb};
                    q[0][e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2));
                } else if constexpr (st == 1) { // p1 = bf16(x - p0)
                    ra = ra - __uint_as_float(q[0][e] << 16);
                    rb = rb - __uint_as_float(q[0][e] & 0xFFFF0000u);
                    const f32x2 v1 = {ra, rb};
                    q[1][e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2));
                } else { // p2 = x - p0 - p1 (exact, <= 8 bits)
                    ra = ra - __uint_as_float(q[1][e] << 16);
                    rb = rb - __uint_as_float(q[1][e] & 0xFFFF0000u);
                    const f32x2 v2 = {ra, rb};
                    q[2][e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2));
                }
            } else if constexpr (t == 1 && i < 15) {
                if constexpr (!(DBG & 8)) stage_item_piece(Pn, g, q, i - 12);
            } else if constexpr (t == 1 && i < 23) {
                if constexpr (!(DBG & 4) && !(DBG & 64)) {
                    const char *pb = (const char *)(pnext + (size_t)(i - 15) * hw); // uniform
                    uint32_t off = pbyte[g];
                    asm volatile("" : "+v"(off)); // keeps the zero-extension next to the load: `global_load v, v_off, s[base]`
                    pv[cp ^ 1][g][i - 15] = *(const float *)(pb + off);
                }
            } else if constexpr (t == 2 && i < 2 * NWU) {
                constexpr int j = i / 2;
                if constexpr (i % 2 == 0) {
                    const int u = tid + j * NT;
                    if constexpr (!(DBG & 8)) { if (u < WU) Wn[u] = wv[ws][j]; }
                } else {
                    if constexpr (!(DBG & 4) && !(DBG & 128)) {
                        uint32_t off = wbyte[j];
                        asm volatile("" : "+v"(off));
                        wv[ws][j] = *(const u32x4 *)((const char *)wnext + off);
                    }
                }
            }
        };
        auto products = [&](auto T_tag) {
            constexpr int t = decltype(T_tag)::value;
            constexpr int cur = t & 1;
#define S3P_MUL(PA, PB, CT_, PT_, M)                                                                                                    \
    if constexpr (DBG != 2) acc[CT_][PT_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][PA][CT_], bv[cur][PB][PT_], acc[CT_][PT_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                                                  \
    micro(std::integral_constant<int, t * 24 + (M)>{});                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define S3P_PRODUCT(PA, PB, QD) S3P_MUL(PA, PB, 0, 0, 4 * QD) S3P_MUL(PA, PB, 0, 1, 4 * QD + 1) S3P_MUL(PA, PB, 1, 0, 4 * QD + 2) S3P_MUL(PA, PB, 1, 1, 4 * QD + 3)
            // six piece products, small ones first (k_conv_s3's order)
            S3P_PRODUCT(2, 0, 0) S3P_PRODUCT(0, 2, 1) S3P_PRODUCT(1, 1, 2) S3P_PRODUCT(1, 0, 3) S3P_PRODUCT(0, 1, 4) S3P_PRODUCT(0, 0, 5)
#undef S3P_PRODUCT
#undef S3P_MUL
        };
        __builtin_amdgcn_sched_barrier(0);
        products(std::integral_constant<int, 0>{});
        products(std::integral_constant<int, 1>{});
        products(std::integral_constant<int, 2>{});
    };

    for (int pair = 0; pair < (nchunks >> 1); ++pair) {
        step(std::integral_constant<int, 0>{}, pair);
        step(std::integral_constant<int, 1>{}, pair);
        step(std::integral_constant<int, 2>{}, pair);
        step(std::integral_constant<int, 3>{}, pair);
        step(std::integral_constant<int, 4>{}, pair);
        step(std::integral_constant<int, 5>{}, pair);
    }

    // epilogue as k_conv_s3: register r of a lane is channel (r&3)+8(r>>2)+4*hsel of pixel l31
    uint32_t omax = 0u;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = (wave * PT + pt) * 32 + l31;
        const int oy = oy0 + m / TW, ox = ox0 + (m % TW);
        const bool pok = oy < a.h && ox < a.w_;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cobase + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (pok) {
                    float v = acc[ct][pt][r] + a.bias[ch];
                    v = v > 0.f ? v : v * 0.1f;
                    amax_fold(omax, v);
                    a.y[(((size_t)n * a.cout + ch) * a.h + oy) * a.w_ + ox] = v;
                }
            }
        }
    }
    if (a.am_out) amax_commit(a.am_out, omax);
}

template <int TW, int DBG = 0>
static int launch_conv_s3p(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    constexpr int TH = 256 / TW;
    constexpr int PP = (TH + 2) * (TW + 2);
    constexpr size_t lds = (size_t)(2 * 6 * PP + 2 * 3 * 6 * 64) * 16;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        V2E_HIP(hipFuncSetAttribute((const void *)k_conv_s3p<TW, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    a.tiles_x = (a.w_ + TW - 1) / TW;
    a.tiles_y = (a.h + TH - 1) / TH;
    a.tl_on = g_s3p_timeline_on;
    const int ntiles = a.n * a.tiles_x * a.tiles_y, ncb = a.cout / 64;
    a.ncb = ncb >= 2 ? ncb : 0;
    dim3 grid = a.ncb ? dim3((unsigned)((ntiles + 7) / 8 * 8 * ncb)) : dim3((unsigned)ntiles, (unsigned)ncb);
    k_conv_s3p<TW, DBG><<<grid, 256, lds, s>>>(a);
    return 0;
}

// 1 if the layer is not one of its shapes
static int conv_dispatch_s3p(const ConvArgs &a, int ks, hipStream_t s)
{
    if (ks != 3 || a.cout % 64 != 0 || a.cin % 32 != 0 || a.c0 % 16 != 0 || (a.x1 && a.c1 % 16 != 0)) return 1;
    if (a.w_ % 32 == 0) return launch_conv_s3p<32>(a, s);
    if (a.w_ % 16 == 0) return launch_conv_s3p<16>(a, s);
    if (a.w_ % 8 == 0) return launch_conv_s3p<8>(a, s);
    return 1;
}
