// slomo_s3.h -- the UNet convolutions on the bf16 matrix cores at f32 accuracy (included by slomo.hip).
//
// v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (157 TF/s peak, 1/16 of the bf16 MFMA rate).  Here every f32
// operand is split EXACTLY into three bf16 pieces, x = p0 + p1 + p2 (round-to-nearest at each step: p0 = bf16(x),
// p1 = bf16(x - p0), p2 = x - p0 - p1, which has at most 8 significant bits left and is therefore a bf16), and of
// the nine piece products of w*x the six with i + j <= 2 are accumulated in f32 by v_mfma_f32_32x32x16_bf16 (bf16 x
// bf16 products are exact in f32).  The dropped terms p1*q2 + p2*q1 + p2*q2 are <= 2^-23 |w x| in the worst case
// (|p1| <= 2^-8 |x|, |p2| <= 2^-16 |x|) and 2^-26 |w x| on average (tests/test_slomo_split.py): at or below one f32
// rounding of the product itself, so the result differs from the f32-MFMA kernel (an fmaf chain) by no more than
// summation order does -- measured against double-precision sums it is the closer of the two
// (scripts/conv_s3_check.hip) -- and both are checked against the same reference-generated goldens at
// 1e-5 * max(1,|y|).  Six bf16 MFMAs of K=16 (6 x 32 cycles) replace eight f32 MFMAs of K=2 (8 x 64 cycles): the
// matrix-core time of a layer drops 2.67x; the price is operand bytes (6 B per element in LDS instead of 4).
//
// Same implicit GEMM as k_conv: D[co][pixel] += W[co][k] X[k][pixel]; a K=16 slab is 16 consecutive input channels
// at one (ky,kx) tap, a lane holding 8 of them (one 16-byte LDS read per operand piece):
//   patch  in LDS  [piece][ci/8][py][px][8 bf16]     (split while staging, one thread = 8 channels of one patch pixel)
//   weights in LDS [tap][piece][ci/8][co][8 bf16]    (split ONCE at pack time: v2e_pack_conv_weight_s3)
// A and B use the same (lane>>5, element) -> channel assignment, so the contraction pairs the right channels whatever
// order the hardware walks a lane's eight elements in.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // plain vector type: stays in registers (uint4 arrays went to scratch)

// two f32 -> their three bf16 pieces, packed (first element in the low half)
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t &q0, uint32_t &q1, uint32_t &q2)
{
    const f32x2 v0 = {a, b};
    q0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2)); // v_cvt_pk_bf16_f32 (RNE)
    const float ra = a - __uint_as_float(q0 << 16), rb = b - __uint_as_float(q0 & 0xFFFF0000u); // exact
    const f32x2 v1 = {ra, rb};
    q1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2));
    const float sa = ra - __uint_as_float(q1 << 16), sb = rb - __uint_as_float(q1 & 0xFFFF0000u); // exact, <= 8 bits
    const f32x2 v2 = {sa, sb};
    q2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2));
}

// ---- the same scheme with TWO float16 pieces (NP = 2, conv_math "fp16x2"): x = h0 + h1 + r, h0 = f16(x), h1 = f16(x - h0)
// (round-to-nearest), |r| <= 2^-22 |x| while both pieces are normal float16 numbers (11 significant bits each; f16 x f16
// products are exact in f32), and the three products h0 g0, h0 g1, h1 g0 on v_mfma_f32_32x32x16_f16: HALF the multiplies and
// two thirds of the operand bytes of the three-piece scheme, for a product that is good to ~2^-21 instead of ~2^-24 -- about
// the size of the float32 rounding noise of a 2 000-term sum itself.  Pieces below the float16 normal range (|x| < 6e-5 for
// h0, < 0.06 for the h1 of it) lose bits to float16's subnormals: an ABSOLUTE error <= 6e-8 per operand, which is what the
// 1e-5 max(1, |y|) bar is about.  Measured against the reference and against float64: tests/test_slomo_gpu.py.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_pair(float a, float b, uint32_t &q0, uint32_t &q1)
{
    const f32x2 v0 = {a, b};
    const f16x2 h0 = __builtin_convertvector(v0, f16x2); // v_cvt_f16_f32 (RNE)
    q0 = __builtin_bit_cast(uint32_t, h0);
    const f32x2 v1 = {a - (float)h0[0], b - (float)h0[1]}; // exact
    q1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, f16x2));
}

// NP pieces of a pair of f32 values: q[p] = pieces p of (a, b), packed
template <int NP> __device__ __forceinline__ void split_pair(float a, float b, uint32_t (&q)[NP])
{
    if constexpr (NP == 3) split3_pair(a, b, q[0], q[1], q[2]);
    else split2_pair(a, b, q[0], q[1]);
}

// one piece product on the matrix cores: 16-byte operand registers as they come from LDS
template <int NP> __device__ __forceinline__ f32x16 mfma_pieces(u32x4 a, u32x4 b, f32x16 c)
{
    if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// weights: torch [Cout][Cin][k][k] f32 -> [Cin/16][k*k][piece][ci/8 (2)][Cout][8 x 16 bit]; one thread per 16-byte unit set
// wscale (NP = 2): an exact power of two that lifts the layer's weights out of float16's subnormal range before they are
// split (the convolution's epilogue divides it out again, exactly)
template <int NP>
__global__ void k_pack_weight_s3(const float *__restrict__ w, uint4 *__restrict__ ws, int cout, int cin, int kk, float wscale = 1.0f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nchunk = (cin + 15) / 16; // a last partial chunk is padded with zero weights
    const size_t total = (size_t)nchunk * 2 * kk * cout;
    if (i >= total) return;
    const int co = (int)(i % cout);
    size_t r = i / cout;
    const int cig = (int)(r % 2); r /= 2;
    const int tap = (int)(r % kk);
    const int chunk = (int)(r / kk);
    const int c0 = chunk * 16 + cig * 8;
    const float *src = w + ((size_t)co * cin + c0) * kk + tap;
    uint32_t q[NP][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float w0 = c0 + 2 * e < cin ? src[(size_t)(2 * e) * kk] * wscale : 0.f;
        const float w1 = c0 + 2 * e + 1 < cin ? src[(size_t)(2 * e + 1) * kk] * wscale : 0.f;
        uint32_t qe[NP];
        split_pair<NP>(w0, w1, qe);
#pragma unroll
        for (int p = 0; p < NP; ++p) q[p][e] = qe[p];
    }
    const size_t base = ((size_t)chunk * kk + tap) * 2 * NP;
#pragma unroll
    for (int p = 0; p < NP; ++p) ws[(base + p * 2 + cig) * cout + co] = make_uint4(q[p][0], q[p][1], q[p][2], q[p][3]);
}

// activations: [n][C][h][w] f32 -> [piece][n][C/8][h][w][8 bf16] (C a multiple of 8); one thread per (n, c/8, pixel)
__global__ __launch_bounds__(256) void k_split3_nchw(const float *__restrict__ x, uint4 *__restrict__ xs, long long nc8, int hw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nc8 * hw) return;
    const long long g = i / hw;
    const int p = (int)(i - g * hw);
    const float *src = x + (g * 8) * hw + p;
    uint32_t q[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3_pair(src[(size_t)(2 * e) * hw], src[(size_t)(2 * e + 1) * hw], q[0][e], q[1][e], q[2][e]);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) xs[(size_t)pc * nc8 * hw + i] = make_uint4(q[pc][0], q[pc][1], q[pc][2], q[pc][3]);
}

// registers: two 4-wave workgroups per CU (60 KB of LDS each) means two waves per SIMD, i.e. up to 256 VGPRs; left alone the
// compiler aims for more waves than the LDS allows and spills the prefetch registers to scratch.  Five-wave 20-wide tiles
// (49 KB) fit three workgroups per CU.
// MODE 1 (PADC): cin is not a multiple of 16 (the 12-channel conv1): channels beyond cin are staged as zeros (their weights
// are zero too).  MODE 2 (PRES): the input arrives already split, [piece][n][C/8][h][w][8 bf16] (k_split3_nchw or a
// producer that writes this layout): staging is three 16-byte loads and three 16-byte LDS stores per item, no conversion.
// RG (3x3 only): the weights of ONE kernel row resident at a time, as 5x5 / 7x7 do (18 KB instead of 55 KB for a 64-channel
// tile: 50 KB of LDS in all, so two workgroups of 64-channel tiles share a CU, and a 64 x 64 register tile reads half the LDS
// bytes per multiply of a 32 x 64 one).
template <int KS, int CT, int PT, int WP, int TW, int NB, int MODE = 0, int RG = 0, int NP = 3>
__global__ __launch_bounds__(WP * 64)
__attribute__((amdgpu_waves_per_eu(CT * PT * WP <= 5 ? 3 : (CT * PT * WP <= 8 || RG ? 2 : 1), CT * PT * WP <= 5 ? 4 : (CT * PT * WP <= 10 || RG ? 2 : 1))))
void k_conv_s3(ConvArgs a)
{
    constexpr bool PADC = MODE == 1, PRES = MODE == 2;
    constexpr int NT = WP * 64;
    constexpr int PAD = KS / 2;
    constexpr int NPX = WP * PT * 32;
    constexpr int TH = NPX / TW;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PP = PH * PW;
    constexpr int COT = CT * 32;
    constexpr int KK = KS * KS;
    constexpr int G = (KS == 3 && !RG) ? 9 : KS; // taps whose weights are resident in LDS at a time (3x3: all; else one kernel row)
    constexpr int NG = KK / G;
    constexpr bool ALLTAPS = G == KK;
    constexpr int NPI = (2 * PP + NT - 1) / NT; // patch items per thread; item = 8 channels of one patch pixel
    constexpr int WU = G * 2 * NP * COT;        // 16-byte weight units per group
    static_assert(NP == 3 || (NP == 2 && MODE != 2), "two-piece float16 operands: no pre-split input");
    constexpr int NWU = (WU + NT - 1) / NT;
    static_assert(NPX % TW == 0, "tile");
    extern __shared__ u32x4 s3_smem[];
    u32x4 *sp = s3_smem;               // [NP][2][PP]
    u32x4 *sw = s3_smem + 2 * NP * PP; // [G][NP][2][COT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hsel = lane >> 5, l31 = lane & 31;
    // workgroup -> (pixel tile, channel block).  Workgroups go to the 8 XCDs round-robin by linear id and each XCD has its
    // own L2, so the channel blocks of ONE pixel tile are consecutive on ONE XCD (ids L, L+8, L+16, ...): the tile's input
    // patch crosses the fabric once instead of once per channel block, and since the workgroups of an XCD walk the
    // input-channel chunks roughly in step, the weight slices they share stay L2-resident too (measured: DESIGN.md 4).
    int bx, cblk;
    if (a.ncb > 0) {
        const int L = blockIdx.x, j = L >> 3;
        cblk = j % a.ncb;
        bx = (j / a.ncb) * 8 + (L & 7);
        if (bx >= a.n * a.tiles_x * a.tiles_y) return; // padding of the last group of 8 pixel tiles
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

    int bofs[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = (wave * PT + pt) * 32 + l31;
        bofs[pt] = hsel * PP + (m / TW) * PW + (m % TW);
    }
    const int aofs = hsel * COT + l31;

    int pinfo[NPI]; // first channel of the item: offset inside one sample's chunk, or -1 (zero padding / unused)
    int pcig[NPI];  // PADC: the item's channel group (0 / 1)
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int i = tid + j * NT;
        int v = -1;
        pcig[j] = i < PP ? 0 : 1;
        if (i < 2 * PP) {
            const int cig = i / PP, r = i - cig * PP;
            const int py = r / PW, px = r - py * PW;
            const int gy = oy0 + py - PAD, gx = ox0 + px - PAD;
            if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w_) v = cig * (PRES ? 1 : 8) * hw + gy * a.w_ + gx; // PRES: 16-byte units
        }
        pinfo[j] = v;
    }
    int woff[NWU]; // unit -> offset inside one group's weight block (units)
#pragma unroll
    for (int j = 0; j < NWU; ++j) {
        const int u = tid + j * NT;
        const int uu = u < WU ? u : WU - 1; // surplus threads load a valid unit and do not store it
        const int r = uu / COT, c = uu - r * COT;
        woff[j] = r * a.cout + cobase + c;
    }

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    // global -> register prefetch, DEEP chunks ahead.  A fetch round trip is ~4 us under load and one chunk of a
    // 3x3 layer is ~2.4 us of multiplies, so the 3x3 kernels (one weight group per chunk) keep TWO chunks in flight
    // in two register sets; 5x5 / 7x7 chunks are 3-5x longer and have no registers to spare.
    constexpr int DEEP = (NG == 1 && CT * PT == 2) ? 2 : 1;
    float pv[PRES ? 1 : DEEP][PRES ? 1 : NPI][8];
    u32x4 pvs[PRES ? DEEP : 1][PRES ? NPI : 1][3]; // PRES: the item's three pieces as they are in memory
    u32x4 wv[DEEP][NWU];
    // loads are unconditional (padding reads a valid address and is zeroed by a select): no branches between them
    auto prefetch_patch = [&](int set, int cb) {
        if (PRES) {
            const u32x4 *base = (const u32x4 *)a.x0 + ((size_t)n * (a.c0 >> 3) + (cb >> 3)) * hw;
#pragma unroll
            for (int j = 0; j < NPI; ++j) {
                const int pi = pinfo[j] < 0 ? 0 : pinfo[j];
#pragma unroll
                for (int p = 0; p < 3; ++p) pvs[PRES ? set : 0][PRES ? j : 0][p] = base[(size_t)p * a.xs_plane + pi];
            }
            return;
        }
        const float *src;
        int cs, C;
        if (cb < a.c0) { src = a.x0; cs = cb; C = a.c0; }
        else { src = a.x1; cs = cb - a.c0; C = a.c1; }
        const float *base = src + ((size_t)n * C + cs) * hw;
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int pi = pinfo[j];
            const float *q = base + (pi < 0 ? 0 : pi);
            if (PADC) { // channels beyond cin: read the last valid one (zeroed when staged)
                const int nv = a.cin - cb - pcig[j] * 8; // valid channels of this item (may be <= 0: all padding)
                if (nv <= 0) q = base;
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[set][j][e] = q[(size_t)(nv <= 0 ? 0 : (e < nv ? e : nv - 1)) * hw];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[set][j][e] = q[(size_t)e * hw];
            }
        }
    };
    auto prefetch_w = [&](int set, int cb, int g) {
        const u32x4 *wb = wsrc + ((size_t)(cb >> 4) * KK + g * G) * 2 * NP * a.cout;
#pragma unroll
        for (int j = 0; j < NWU; ++j) wv[set][j] = wb[woff[j]];
    };
    // NP == 2: activations are staged times in_scale (a power of two from the producers' range slots: ConvArgs) and the
    // epilogue multiplies the sums by inv_scale again; without slots (a convolution outside v2e_unet_forward) both are 1
    float in_scale = 1.0f, inv_scale = 1.0f;
    bool in_bad = false;
    if constexpr (NP == 2) act_scale(a, in_scale, inv_scale, in_bad);
    // (the range flag is raised from the producers' slots: an inf / NaN there; the scaled operands cannot leave float16's range.
    //  A two-piece convolution launched WITHOUT slots -- outside v2e_unet_forward -- stages unscaled and is not watched.)
    auto stage_patch = [&](int set, int cbs = 0) { // split and store this thread's patch items (cbs: the chunk, PADC only)
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int i = tid + j * NT;
            if (PRES) {
                if (i < 2 * PP) {
                    const bool ok = pinfo[j] >= 0;
#pragma unroll
                    for (int p = 0; p < 3; ++p) sp[p * 2 * PP + i] = ok ? pvs[PRES ? set : 0][PRES ? j : 0][p] : u32x4{0u, 0u, 0u, 0u};
                }
            } else if (i < 2 * PP) {
                const bool ok = pinfo[j] >= 0;
                const int nv = PADC ? a.cin - cbs - pcig[j] * 8 : 8;
                uint32_t q[NP][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t qe[NP];
                    float xa = ok && 2 * e < nv ? pv[set][j][2 * e] : 0.f, xb = ok && 2 * e + 1 < nv ? pv[set][j][2 * e + 1] : 0.f;
                    if constexpr (NP == 2) { xa *= in_scale; xb *= in_scale; } // exact (a power of two); replaces round 3's running
                                                                               // maximum one for one: these kernels are power-bound
                    split_pair<NP>(xa, xb, qe);
#pragma unroll
                    for (int p = 0; p < NP; ++p) q[p][e] = qe[p];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) sp[p * 2 * PP + i] = u32x4{q[p][0], q[p][1], q[p][2], q[p][3]};
            }
        }
    };
    auto stage_w = [&](int set) {
#pragma unroll
        for (int j = 0; j < NWU; ++j) {
            const int u = tid + j * NT;
            if (u < WU) sw[u] = wv[set][j];
        }
    };

    // NB operand register sets: with 2, tap t+1 is read from LDS before the multiplies of tap t are issued
    u32x4 av[NB][NP][CT], bv[NB][NP][PT];
    auto load_ops = [&](int t, int buf, int ky0) {
        const int ky = ALLTAPS ? t / KS : 0, kx = ALLTAPS ? t % KS : t;
        const int koff_p = (ky0 + ky) * PW + kx;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[buf][p][ct] = sw[aofs + (t * 2 * NP + p * 2) * COT + ct * 32];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) bv[buf][p][pt] = sp[bofs[pt] + p * 2 * PP + koff_p];
        }
    };
    auto compute_group = [&](int ky0) { // the G taps whose weights are in LDS
        if (NB == 2) load_ops(0, 0, ky0);
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int cur = NB == 2 ? (t & 1) : 0;
            if (NB == 1) load_ops(t, 0, ky0);
            else if (t + 1 < G) load_ops(t + 1, cur ^ 1, ky0);
            if (NB == 2) __builtin_amdgcn_sched_barrier(0);
            // six piece products, small ones first; consecutive multiplies go to different accumulators
#define S3_MFMA(PA, PB)                                                                                              \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)               \
        acc[ct][pt] = mfma_pieces<NP>(av[cur][PA][ct], bv[cur][PB][pt], acc[ct][pt]);
            if constexpr (NP == 3) { S3_MFMA(2, 0) S3_MFMA(0, 2) S3_MFMA(1, 1) }
            S3_MFMA(1, 0) S3_MFMA(0, 1) S3_MFMA(0, 0)
#undef S3_MFMA
            if (NB == 2) __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (DEEP == 2) {
        prefetch_w(0, 0, 0);
        prefetch_patch(0, 0);
        int cb = 0;
        if (a.cin >= 64) {
            prefetch_w(DEEP - 1, 16, 0);
            prefetch_patch(DEEP - 1, 16);
            // steady state: both chunks of the pair have a successor two chunks on, so the prefetch is unconditional (no
            // branch between the loads of a set and its use) and the wait before staging a set covers that set only:
            // the newer set's loads stay in flight across the barrier
            for (; cb + 64 <= a.cin; cb += 32) {
#pragma unroll
                for (int set = 0; set < DEEP; ++set) {
                    const int c = cb + 16 * set;
                    __syncthreads(); // everyone is done reading what is about to be overwritten
                    stage_patch(set);
                    stage_w(set);
                    __syncthreads();
                    prefetch_w(set, c + 32, 0);
                    prefetch_patch(set, c + 32);
                    compute_group(0);
                }
            }
        } else if (a.cin > 16) {
            prefetch_w(DEEP - 1, 16, 0);
            prefetch_patch(DEEP - 1, 16);
        }
        for (; cb < a.cin; cb += 32) { // the last one to three chunks
#pragma unroll
            for (int set = 0; set < DEEP; ++set) {
                const int c = cb + 16 * set;
                if (c < a.cin) {
                    __syncthreads();
                    stage_patch(set);
                    stage_w(set);
                    __syncthreads();
                    if (c + 32 < a.cin) { prefetch_w(set, c + 32, 0); prefetch_patch(set, c + 32); }
                    compute_group(0);
                }
            }
        }
    } else {
        prefetch_w(0, 0, 0);
        prefetch_patch(0, 0);
        for (int cb = 0; cb < a.cin; cb += 16) {
            for (int g = 0; g < NG; ++g) {
                __syncthreads(); // everyone is done reading what is about to be overwritten
                if (g == 0) stage_patch(0, cb);
                stage_w(0);
                __syncthreads();
                if (g + 1 < NG) prefetch_w(0, cb, g + 1);
                else if (cb + 16 < a.cin) { prefetch_w(0, cb + 16, 0); prefetch_patch(0, cb + 16); }
                compute_group(ALLTAPS ? 0 : g); // G == KS: group g is kernel row g
            }
        }
    }
    // NP == 2: an activation beyond float16's range (it became +-inf in the split) -- or a NaN -- is reported, so that the caller
    // can redo the layer stack with the exact three-piece split (v2e_conv_set_range_flag)
    if constexpr (NP == 2) {
        if (a.ovf && in_bad && tid == 0) atomicOr(a.ovf, 1);
    }
    uint32_t omax = 0u;
    // the two scale factors as ONE multiply: act_scale keeps 2^-(k + s) a normal float32, so the product is exact
    const float os1 = NP == 2 ? a.out_scale * inv_scale : 1.0f;
    // epilogue as k_conv: register r of a lane is channel (r&3)+8(r>>2)+4*hsel of pixel l31
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
                    float s = acc[ct][pt][r];
                    if constexpr (NP == 2) s *= os1; // the powers of two the operands were staged times, divided out (exact)
                    float v = s + a.bias[ch];
                    v = v > 0.f ? v : v * 0.1f;
                    amax_fold(omax, v);
                    a.y[(((size_t)n * a.cout + ch) * a.h + oy) * a.w_ + ox] = v;
                }
            }
        }
    }
    if (a.am_out) amax_commit(a.am_out, omax);
    // avg_pool2d(2) of what was just written (model.py:96: the next block's input), in k_avgpool2's order of additions:
    // ((row0[x] + row0[x + 1]) + row1[x]) + row1[x + 1], times 0.25.  A wave of these tiles holds rows 2 wave, 2 wave + 1 of the
    // tile (pt 0 / 1), a lane one pixel of each: the window is the lane's two registers and its odd neighbour's.
    if constexpr (TW == 32 && PT == 2 && (WP * PT) % 2 == 0) {
        if (a.ypool) {
            const int oy = oy0 + wave * 2, ox = ox0 + l31; // row of pt 0 (even), this lane's column
            const int ph = a.h >> 1, pw = a.w_ >> 1;
            const bool st = (l31 & 1) == 0 && oy + 1 < a.h && ox + 1 < a.w_;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = cobase + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                    float v[2];
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) { // the stored value, recomputed from the accumulator (same operations)
                        float sacc = acc[ct][pt][r];
                        if constexpr (NP == 2) sacc *= os1;
                        float t = sacc + a.bias[ch];
                        v[pt] = t > 0.f ? t : t * 0.1f;
                    }
                    const float n0 = __shfl_xor(v[0], 1), n1 = __shfl_xor(v[1], 1);
                    const float o = (((v[0] + n0) + v[1]) + n1) * 0.25f;
                    if (st) a.ypool[(((size_t)n * a.cout + ch) * ph + (oy >> 1)) * pw + (ox >> 1)] = o;
                }
            }
        }
    }
}

template <int KS, int CT, int PT, int WP, int TW, int NB = 1, int MODE = 0, int RG = 0, int NP = 3>
static int launch_conv_s3(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    constexpr int TH = WP * PT * 32 / TW;
    constexpr int PP = (TH + KS - 1) * (TW + KS - 1);
    constexpr int G = (KS == 3 && !RG) ? 9 : KS;
    constexpr size_t lds = (size_t)(2 * NP * PP + G * 2 * NP * CT * 32) * 16;
    a.ypool = nullptr;
    if (TW == 32 && PT == 2 && g_pool_out && a.h % 2 == 0 && a.w_ % 2 == 0) { a.ypool = g_pool_out; g_pool_out = nullptr; } // taken
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false; // one per instantiation
    if (!attr_set) {
        V2E_HIP(hipFuncSetAttribute((const void *)k_conv_s3<KS, CT, PT, WP, TW, NB, MODE, RG, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    a.tiles_x = (a.w_ + TW - 1) / TW;
    a.tiles_y = (a.h + TH - 1) / TH;
    constexpr int order = 1; // channel blocks of a pixel tile consecutive (round 2; 0 = pixel tiles fastest measured slower)
    const int ntiles = a.n * a.tiles_x * a.tiles_y, ncb = a.cout / (CT * 32);
    // measured at 40 samples: +6 % where a pixel tile has >= 4 channel blocks (256->128 at 64x80, 512->256 at 32x40), within
    // noise at 2 blocks, -3 % on the five-wave 20-wide tiles (16 blocks: more workgroups than an XCD holds at once)
    a.ncb = ((order == 1 && KS == 3 && WP == 4 && ncb >= 4) || (order == 2 && ncb > 1)) ? ncb : 0;
    dim3 grid = a.ncb ? dim3((unsigned)((ntiles + 7) / 8 * 8 * ncb)) : dim3((unsigned)ntiles, (unsigned)ncb);
    k_conv_s3<KS, CT, PT, WP, TW, NB, MODE, RG, NP><<<grid, WP * 64, lds, s>>>(a);
    return 0;
}
