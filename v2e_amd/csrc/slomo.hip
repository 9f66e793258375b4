// slomo.hip -- SuperSloMo frame interpolation (two UNets + backWarp + fusion) for gfx950.
//
// Reference (SensorsINI/v2e): v2ecore/model.py:10-226 (UNet/down/up), :229-300 (backWarp),
// v2ecore/slomo.py:338-345 (flow UNet), :404-433 (per-t blend, 4 warps, interp UNet, fusion).
//
// Convolutions are implicit GEMMs on the exact-f32 matrix cores:
//   D[co][pixel] += W[co][k] * X[k][pixel],  k = (ci, ky, kx)
// with v_mfma_f32_32x32x2_f32 (A = 32 output channels x 2 k, B = 2 k x 32 pixels).  The pixel
// axis is the MFMA column axis, so every accumulator register is 32 consecutive x of one
// output channel: NCHW stores are 128-byte coalesced.  The f32 MFMA is bit-identical to an
// fmaf chain in k order, and the k order here is fixed (ci-chunk, ky, kx, ci-pair), so results
// are deterministic run to run; parity with torch-CPU (different summation order) is checked
// at 1e-5 * max(1,|y|).
//
// Per workgroup: WP waves; each wave owns CT x PT tiles of 32 channels x 32 pixels.  Per
// ci-chunk the input patch (with halo, zero padding, and optionally the preceding
// avg_pool2d / bilinear x2 upsample / channel concat applied on the fly) and the weight
// slice are staged in LDS; both operand reads are then `lane base + immediate` ds_read_b32
// over 32 consecutive dwords per half-wave (conflict-free).
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *x0, *x1; // x1: second concat source or nullptr
    int c0, c1;           // channels of x0 / x1
    const float *w;       // packed [Cin][KS][KS][Cout]
    const float *bias;
    float *y;
    int n, h, w_, cin, cout;
    int tiles_x, tiles_y;
    const void *ws3; // split-bf16 weights (slomo_s3.h) or nullptr
    int ncb;         // slomo_s3.h: channel blocks when the grid is 1-D in XCD order, else 0
    long long xs_plane; // slomo_s3.h, pre-split input: 16-byte units per piece plane (n * C/8 * h * w)
    int tl_on;          // slomo_s3p.h, dev: record workgroup 0's per-step timeline
    int np;             // pieces the split weights ws3 hold: 3 (bf16) or 2 (float16)
    float out_scale;    // np == 2: 2^-s, the inverse of the power of two the weights were packed times
    int *ovf;           // np == 2: device flag set when an activation is beyond float16's range (or NaN), or nullptr
    // Activation range tracking (v2e_unet_forward): every convolution leaves the largest |output| it wrote in *am_out (float32
    // bits: non-negative floats, inf and NaN order like their bit patterns; atomicMax per wave), and a two-float16-piece
    // convolution reads the maxima of its input's producer(s) and stages its activations times the power of two that puts that
    // maximum in [2^13, 2^14): no piece of an activation that matters falls into float16's subnormals however small the
    // layer's activations are, nothing overflows however large; the epilogue divides the power out again (exact).
    const uint32_t *am_in0, *am_in1;
    uint32_t *am_out;
    // avg_pool2d(2) of the output, written beside it by the epilogue ([n][cout][h/2][w/2]; k_conv_s3 tiles of 32-pixel rows, two
    // rows per wave: the 2 x 2 window is two registers of a lane and its neighbour lane), or nullptr
    float *ypool;
};

// |v| folded into a running maximum kept as float32 bits: one v_max_f32 with the |.| source modifier.  Non-negative floats and
// +inf order like their bit patterns, so the per-wave / per-slot reductions are integer maxima.  A NaN is ignored by the
// maximum: a NaN activation gives a NaN result in every conv math alike (as in the reference); what the range guard needs to
// see is inf, and it does.
__device__ __forceinline__ void amax_fold(uint32_t &m, float v)
{
    m = __float_as_uint(fmaxf(__uint_as_float(m), fabsf(v)));
}
// one atomicMax per wave AT MOST: a layer is tens of thousands of workgroups and an atomic to one address is serialised where
// it is performed (measured: +1.3 ms per forward pass, 8 %, with an unconditional atomic per wave), so the wave first reads
// the slot -- a plain load; a stale (smaller) value only costs an atomic that was not needed, never a wrong maximum -- and
// only a wave that would raise it goes to the atomic unit: a few hundred per layer instead of 100 000
__device__ __forceinline__ void amax_commit(uint32_t *slot, uint32_t m)
{
    // wave maximum on the VALU data-parallel-primitive path (no LDS crossbar round trips at the tail of every wave)
#define V2E_S_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, true)
    m = max(m, (uint32_t)V2E_S_DPP(m, 0xB1));  // quad_perm [1,0,3,2]   (bit patterns of non-negative floats: all below 2^31)
    m = max(m, (uint32_t)V2E_S_DPP(m, 0x4E));  // quad_perm [2,3,0,1]
    m = max(m, (uint32_t)V2E_S_DPP(m, 0x141)); // row_half_mirror
    m = max(m, (uint32_t)V2E_S_DPP(m, 0x140)); // row_mirror
#undef V2E_S_DPP
    const uint32_t w = max(max((uint32_t)__builtin_amdgcn_readlane((int)m, 0), (uint32_t)__builtin_amdgcn_readlane((int)m, 16)),
                           max((uint32_t)__builtin_amdgcn_readlane((int)m, 32), (uint32_t)__builtin_amdgcn_readlane((int)m, 48)));
    if ((threadIdx.x & 63) == 0 && w != 0u) {
        if (w > *(volatile uint32_t *)slot) atomicMax(slot, w);
    }
}
// the power of two a two-float16-piece convolution stages its activations times: in_scale = 2^k with amax * 2^k in [2^13, 2^14),
// inv = 2^-k (both normal float32 numbers: |k| <= 126); bad = the producer wrote an inf or a NaN
__device__ __forceinline__ void act_scale(const ConvArgs &a, float &in_scale, float &inv, bool &bad)
{
    in_scale = 1.0f; inv = 1.0f; bad = false;
    if (!a.am_in0) return;
    uint32_t m = *a.am_in0;
    if (a.am_in1) { const uint32_t m1 = *a.am_in1; m = m1 > m ? m1 : m; }
    const int e = (int)(m >> 23); // biased exponent of the maximum (sign bit is clear)
    if (e == 255) { bad = true; return; }
    if (m == 0u) return; // an all-zero input
    int k = 140 - e;     // amax in [2^(e-127), 2^(e-126)) -> [2^13, 2^14)
    // ... as far as 2^-k times the weights' 2^-s (a.out_scale) stays a normal float32, so that the epilogue divides both out
    // with ONE exact multiply: s <= 40, i.e. maxima down to 2^-73 are scaled all the way
    const int sl2 = 127 - (int)((__float_as_uint(a.out_scale) >> 23) & 0xFF);
    const int kmax = 126 - (sl2 > 0 ? sl2 : 0);
    k = k > kmax ? kmax : (k < -126 ? -126 : k);
    in_scale = __uint_as_float((uint32_t)(127 + k) << 23);
    inv = __uint_as_float((uint32_t)(127 - k) << 23);
}

// where v2e_unet_forward hands the next convolution its range slots (per host thread; all null outside a forward pass)
static thread_local const uint32_t *g_am_in0 = nullptr, *g_am_in1 = nullptr;
static thread_local uint32_t *g_am_out = nullptr;
// v2e_unet_forward asks the next convolution to write its avg_pool2d(2) too; the launcher that can do it clears the request
static thread_local float *g_pool_out = nullptr;

// where the two-float16-piece convolutions report an activation beyond float16's range (v2e_conv_set_range_flag), or nullptr
static thread_local int *g_conv_range_flag = nullptr;

// element fetch with the producer op fused: PRE 0 plain, 1 avg_pool2d(2) of a [2H][2W] source,
// 2 bilinear x2 upsample (align_corners=False) of a [H/2][W/2] source.
template <int PRE>
__device__ __forceinline__ float fetch(const ConvArgs &a, int n, int c, int gy, int gx)
{
    const float *src;
    int cs, C;
    if (c < a.c0) { src = a.x0; cs = c; C = a.c0; }
    else { src = a.x1; cs = c - a.c0; C = a.c1; }
    if (PRE == 0) {
        return src[(((size_t)n * C + cs) * a.h + gy) * a.w_ + gx];
    } else if (PRE == 1) { // F.avg_pool2d(x, 2): (((a+b)+c)+d) / 4
        const int sw = 2 * a.w_;
        const float *p = src + (((size_t)n * C + cs) * (2 * a.h) + 2 * gy) * sw + 2 * gx;
        return (((p[0] + p[1]) + p[sw]) + p[sw + 1]) * 0.25f;
    } else { // F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)
        const int sh = a.h >> 1, sw = a.w_ >> 1;
        float ry = ((float)gy + 0.5f) * 0.5f - 0.5f;
        float rx = ((float)gx + 0.5f) * 0.5f - 0.5f;
        ry = ry < 0.f ? 0.f : ry;
        rx = rx < 0.f ? 0.f : rx;
        const int y0 = (int)ry, x0 = (int)rx;
        const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
        const float ly = ry - (float)y0, lx = rx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float *p = src + ((size_t)n * C + cs) * sh * sw;
        return hy * (hx * p[y0 * sw + x0] + lx * p[y0 * sw + x1]) + ly * (hx * p[y1 * sw + x0] + lx * p[y1 * sw + x1]);
    }
}

// STK = 2: the tile is two whole samples stacked (levels whose sample is half a tile, 8x10 of a 320x256 input);
// in the LDS patch the two share ONE zero row between them: [0][A rows][0][B rows][0].
template <int KS, int CI_T, int CT, int PT, int WP, int TW, int PRE, int STK = 1>
__global__ __launch_bounds__(WP * 64) void k_conv(ConvArgs a)
{
    constexpr int NT = WP * 64;
    constexpr int PAD = KS / 2;
    constexpr int NPX = WP * PT * 32;
    constexpr int TH = NPX / TW;
    constexpr int THS = TH / STK; // rows per sample
    constexpr int PH = TH + KS - 1 + (STK - 1), PW = TW + KS - 1;
    static_assert(STK == 1 || (STK == 2 && KS == 3 && PRE == 0 && TH % 2 == 0), "stacked tiles: 3x3, plain fetch");
    constexpr int COT = CT * 32;
    constexpr int KK = KS * KS;
    constexpr int PATCH = CI_T * PH * PW;
    constexpr int PATCH_PAD = (PATCH + 3) / 4 * 4; // keeps the weight slice 16-byte aligned in LDS
    constexpr int WTS = CI_T * KK * COT;
    constexpr int NPE = (PATCH + NT - 1) / NT;     // patch elements staged per thread
    constexpr int NWE = (WTS / 4 + NT - 1) / NT;   // weight float4s staged per thread
    static_assert(NPX % TW == 0, "tile");
    static_assert(CI_T % 2 == 0, "ci pairs");
    __shared__ __attribute__((aligned(16))) float smem[PATCH_PAD + WTS];
    float *sp = smem, *sw = smem + PATCH_PAD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hsel = lane >> 5, l31 = lane & 31;
    int bx = blockIdx.x;
    const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty_i = bx % a.tiles_y;
    const int n = (bx / a.tiles_y) * STK; // first sample of the tile
    const int oy0 = ty_i * TH, ox0 = tx_i * TW;
    const int cobase = blockIdx.y * COT;
    const int hw = a.h * a.w_;
    const bool vecw = (a.cout & 3) == 0; // float4 weight rows (every layer but the 4/5-channel heads)

    int bofs[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = (wave * PT + pt) * 32 + l31;
        const int row = m / TW;
        bofs[pt] = (hsel * PH + row + ((STK == 2 && row >= THS) ? 1 : 0)) * PW + (m % TW);
    }
    const int aofs = hsel * KK * COT + l31;

    // ---- per-thread staging plan, computed once: which patch elements / weight rows this thread
    // moves every chunk.  Patch element idx -> (ci, py, px) is the same in every chunk.
    int pinfo[NPE]; // PRE == 0: ci*hw + gy*w + gx within one sample (or -1: zero padding / unused);
                    // PRE != 0: packed (ci << 24 | py << 12 | px) (or -1)
#pragma unroll
    for (int j = 0; j < NPE; ++j) {
        const int idx = tid + j * NT;
        int v = -1;
        if (idx < PATCH) {
            const int ci = idx / (PH * PW);
            const int r = idx - ci * (PH * PW);
            const int py = r / PW, px = r - py * PW;
            const int gx = ox0 + px - PAD;
            if (STK == 2) { // bit 30: second sample of the tile
                const int sidx = py > THS + 1 ? 1 : 0;
                const int gy = sidx ? py - (THS + 2) : py - PAD;
                if (py != THS + 1 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w_ && n + sidx < a.n) v = (sidx << 30) | (ci * hw + gy * a.w_ + gx);
            } else {
                const int gy = oy0 + py - PAD;
                if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w_) v = PRE == 0 ? ci * hw + gy * a.w_ + gx : ((ci << 24) | (gy << 12) | gx);
            }
        }
        pinfo[j] = v;
    }
    int woff[NWE]; // float4 slot q -> offset of its row start inside one chunk's weight block (or -1)
#pragma unroll
    for (int j = 0; j < NWE; ++j) {
        const int q = tid + j * NT;
        int v = -1;
        if (q < WTS / 4) {
            const int kk = (q * 4) / COT, co4 = (q * 4) - kk * COT;
            if (cobase + co4 < a.cout) v = kk * a.cout + cobase + co4;
        }
        woff[j] = v;
    }

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    float pv[NPE];
    float4 wv[NWE];
    // fetch chunk `cb` into registers (global loads stay in flight while the MFMAs of the previous chunk run)
    auto prefetch = [&](int cb) {
        if (PRE == 0) {
            // all CI_T channels of a chunk come from one concat source (c0 is a multiple of CI_T)
            const float *src;
            int cs, C;
            if (cb < a.c0) { src = a.x0; cs = cb; C = a.c0; }
            else { src = a.x1; cs = cb - a.c0; C = a.c1; }
            const float *base = src + ((size_t)n * C + cs) * hw;
#pragma unroll
            for (int j = 0; j < NPE; ++j) {
                const int pi = pinfo[j];
                if (STK == 2) pv[j] = pi >= 0 ? base[(size_t)(pi & 0x3FFFFFFF) + (size_t)(pi >> 30) * C * hw] : 0.f;
                else pv[j] = pi >= 0 ? base[pi] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPE; ++j) {
                const int pi = pinfo[j];
                pv[j] = pi >= 0 ? fetch<PRE>(a, n, cb + (pi >> 24), (pi >> 12) & 0xFFF, pi & 0xFFF) : 0.f;
            }
        }
        const float *wb = a.w + (size_t)cb * KK * a.cout;
        if (vecw) {
#pragma unroll
            for (int j = 0; j < NWE; ++j) wv[j] = woff[j] >= 0 ? *(const float4 *)(wb + woff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < NWE; ++j) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (woff[j] >= 0) {
                    const int kk = woff[j] / a.cout, co = woff[j] - kk * a.cout;
                    const float *wr = wb + woff[j];
                    t.x = wr[0];
                    if (co + 1 < a.cout) t.y = wr[1];
                    if (co + 2 < a.cout) t.z = wr[2];
                    if (co + 3 < a.cout) t.w = wr[3];
                }
                wv[j] = t;
            }
        }
    };

    prefetch(0);
    for (int cb = 0; cb < a.cin; cb += CI_T) {
        __syncthreads(); // everyone is done reading the previous chunk from LDS
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            const int idx = tid + j * NT;
            if (idx < PATCH) sp[idx] = pv[j];
        }
#pragma unroll
        for (int j = 0; j < NWE; ++j) {
            const int q = tid + j * NT;
            if (q < WTS / 4) *(float4 *)(sw + q * 4) = wv[j];
        }
        __syncthreads();
        if (cb + CI_T < a.cin) prefetch(cb + CI_T);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int cp = 0; cp < CI_T / 2; ++cp) {
                    const int koff_w = ((cp * 2) * KK + ky * KS + kx) * COT;
                    const int koff_p = (cp * 2) * PH * PW + ky * PW + kx;
                    float av[CT], bv[PT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) av[ct] = sw[aofs + koff_w + ct * 32];
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) bv[pt] = sp[bofs[pt] + koff_p];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv[pt], acc[ct][pt], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: bias + leaky_relu(0.1); register r of lane: channel (r&3)+8(r>>2)+4*hsel, pixel l31
    uint32_t omax = 0u;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = (wave * PT + pt) * 32 + l31;
        const int row = m / TW, sidx = STK == 2 ? row / THS : 0;
        const int oy = oy0 + row - sidx * THS, ox = ox0 + (m % TW);
        const bool pok = oy < a.h && ox < a.w_ && n + sidx < a.n;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cobase + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (pok && ch < a.cout) {
                    float v = acc[ct][pt][r] + a.bias[ch];
                    v = v > 0.f ? v : v * 0.1f;
                    amax_fold(omax, v);
                    a.y[(((size_t)(n + sidx) * a.cout + ch) * a.h + oy) * a.w_ + ox] = v;
                }
            }
        }
    }
    if (a.am_out) amax_commit(a.am_out, omax);
}

#include "slomo_s3.h"
#include "slomo_s3p.h"

// ... one output per thread, for widths that are not a multiple of 4
__global__ __launch_bounds__(256) void k_avgpool2_scalar(const float *__restrict__ x, float *__restrict__ y, long long nc, int h, int w)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nc * h * w) return;
    const int ox = (int)(q % w);
    const long long r = q / w;
    const int oy = (int)(r % h);
    const long long c = r / h;
    const float *p = x + ((c * (2 * h) + 2 * oy) * (long long)(2 * w)) + 2 * ox;
    y[q] = (((p[0] + p[1]) + p[2 * w]) + p[2 * w + 1]) * 0.25f;
}

// F.avg_pool2d(x, 2) as its own pass: [nc][2h][2w] -> [nc][h][w]; 4 outputs per thread
__global__ __launch_bounds__(256) void k_avgpool2(const float *__restrict__ x, float *__restrict__ y, long long nc, int h, int w)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x; // group of 4 outputs along x
    const int wq = w >> 2;
    const long long total = nc * h * wq;
    if (q >= total) return;
    const int xq = (int)(q % wq);
    const long long r = q / wq;
    const int oy = (int)(r % h);
    const long long c = r / h;
    const float *p = x + ((c * (2 * h) + 2 * oy) * (long long)(2 * w)) + 8 * xq;
    const float4 a0 = *(const float4 *)p, a1 = *(const float4 *)(p + 4);
    const float4 b0 = *(const float4 *)(p + 2 * w), b1 = *(const float4 *)(p + 2 * w + 4);
    float4 o;
    o.x = (((a0.x + a0.y) + b0.x) + b0.y) * 0.25f;
    o.y = (((a0.z + a0.w) + b0.z) + b0.w) * 0.25f;
    o.z = (((a1.x + a1.y) + b1.x) + b1.y) * 0.25f;
    o.w = (((a1.z + a1.w) + b1.z) + b1.w) * 0.25f;
    *(float4 *)(y + (c * h + oy) * (long long)w + 4 * xq) = o;
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) as its own pass:
// [nc][h/2][w/2] -> [nc][h][w].  One thread = 2 rows x 4 consecutive outputs (two float4 stores) from 2 source rows x 4
// columns: output rows 2t-1 and 2t both lie between source rows t-1 and t (weights 0.75/0.25 and 0.25/0.75; at the
// top edge the clamped coordinate gives weight 0 to the second row), and the source coordinates of outputs 4j..4j+3 are
// 2j-0.25, 2j+0.25, 2j+0.75, 2j+1.25, so the fractional weights are the exact constants that fetch<2> computes, and
// the products/sums are evaluated in the same order.
__global__ __launch_bounds__(256) void k_upsample2(const float *__restrict__ x, float *__restrict__ y, long long nc, int h, int w)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int wq = w >> 2;
    const int sh = h >> 1, sw = w >> 1;
    const long long total = nc * (sh + 1) * wq;
    if (q >= total) return;
    const int j = (int)(q % wq);
    const long long r = q / wq;
    const int t = (int)(r % (sh + 1)); // output rows 2t-1 (t >= 1) and 2t (t < sh)
    const long long c = r / (sh + 1);
    const int ya = t - 1 < 0 ? 0 : t - 1, yb = t == 0 ? (sh > 1 ? 1 : 0) : (t < sh ? t : sh - 1); // t == 0: the reference's y1 (weight 0)
    const float *p0 = x + (c * sh + ya) * (long long)sw, *p1 = x + (c * sh + yb) * (long long)sw;
    // source columns 2j-1 .. 2j+2, clamped
    const int cm = 2 * j - 1 < 0 ? 0 : 2 * j - 1, c0 = 2 * j, c1 = 2 * j + 1 < sw ? 2 * j + 1 : sw - 1,
              c2 = 2 * j + 2 < sw ? 2 * j + 2 : sw - 1;
    const float a_m = p0[cm], a_0 = p0[c0], a_1 = p0[c1], a_2 = p0[c2];
    const float b_m = p1[cm], b_0 = p1[c0], b_1 = p1[c1], b_2 = p1[c2];
    const float lx0 = j == 0 ? 0.f : 0.75f, hx0 = 1.f - lx0; // output 4j: x0 = 2j-1 (0 when clamped), x1 = x0+1
    // when j == 0 the reference has x0 = 0, x1 = min(1, sw-1); a_m == a_0 there and lx = 0
    const float t00 = j == 0 ? a_0 : a_m, t01 = j == 0 ? a_1 : a_0, u00 = j == 0 ? b_0 : b_m, u01 = j == 0 ? b_1 : b_0;
    // the four horizontal blends of each source row (same for both output rows)
    const float ax = hx0 * t00 + lx0 * t01, ay = 0.75f * a_0 + 0.25f * a_1, az = 0.25f * a_0 + 0.75f * a_1, aw = 0.75f * a_1 + 0.25f * a_2;
    const float bx = hx0 * u00 + lx0 * u01, by = 0.75f * b_0 + 0.25f * b_1, bz = 0.25f * b_0 + 0.75f * b_1, bw = 0.75f * b_1 + 0.25f * b_2;
    if (t >= 1) { // row 2t-1: ry = t - 0.75 -> y0 = t-1, ly = 0.25
        const float ly = ((float)(2 * t - 1) + 0.5f) * 0.5f - 0.5f - (float)(t - 1), hy = 1.f - ly;
        float4 o;
        o.x = hy * ax + ly * bx; o.y = hy * ay + ly * by; o.z = hy * az + ly * bz; o.w = hy * aw + ly * bw;
        *(float4 *)(y + (c * h + 2 * t - 1) * (long long)w + 4 * j) = o;
    }
    if (t < sh) { // row 2t: ry = t - 0.25 (clamped to 0 at the top) -> y0 = max(t-1, 0), ly = 0.75 (0 at the top)
        float ry = ((float)(2 * t) + 0.5f) * 0.5f - 0.5f;
        ry = ry < 0.f ? 0.f : ry;
        const float ly = ry - (float)ya, hy = 1.f - ly;
        float4 o;
        o.x = hy * ax + ly * bx; o.y = hy * ay + ly * by; o.z = hy * az + ly * bz; o.w = hy * aw + ly * bw;
        *(float4 *)(y + (c * h + 2 * t) * (long long)w + 4 * j) = o;
    }
}

// same, one output per thread, for widths that are not a multiple of 4
__global__ __launch_bounds__(256) void k_upsample2_scalar(const float *__restrict__ x, float *__restrict__ y, long long nc, int h, int w)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = nc * h * w;
    if (i >= total) return;
    const int gx = (int)(i % w);
    const long long r = i / w;
    const int gy = (int)(r % h);
    const long long c = r / h;
    const int sh = h >> 1, sw = w >> 1;
    float ry = ((float)gy + 0.5f) * 0.5f - 0.5f;
    float rx = ((float)gx + 0.5f) * 0.5f - 0.5f;
    ry = ry < 0.f ? 0.f : ry;
    rx = rx < 0.f ? 0.f : rx;
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
    const float ly = ry - (float)y0, lx = rx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float *p = x + c * sh * sw;
    y[i] = hy * (hx * p[y0 * sw + x0] + lx * p[y0 * sw + x1]) + ly * (hx * p[y1 * sw + x0] + lx * p[y1 * sw + x1]);
}

// Final UNet layer (32 -> 4/5 channels, 3x3): too few output channels for a 32-wide MFMA tile (84 % of it would be padding: at the
// ~190 TF/s the 32-channel full-resolution layers reach, a padded tile would take longer than this), so it runs on the VALU from an
// LDS-staged input patch with wave-uniform scalar weight loads.  Round 6: FOUR pixels along x per thread -- a row of the patch is read
// as 6 floats (one 16-byte and one 8-byte LDS read) for 3 taps x 4 pixels x COUT = 60 multiply-adds, where one pixel per thread read
// 9 floats for 45 (the kernel was bound by its LDS reads: 33 TF/s); same accumulation order (ci, ky, kx) per output.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv3x3_small(const float *__restrict__ x, const float *__restrict__ wp /* [cin][3][3][COUT] */,
                                                       const float *__restrict__ bias, float *__restrict__ y, int n_img, int cin,
                                                       int h, int w, int tiles_x, int tiles_y)
{
    constexpr int PXT = 4, TW = 64, TH = 16, PH = TH + 2, PW = 68 /* 66 padded: rows start 16-byte aligned */, CI_T = 8;
    __shared__ __attribute__((aligned(16))) float sp[CI_T * PH * PW];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx_i = bx % tiles_x; bx /= tiles_x;
    const int ty_i = bx % tiles_y;
    const int n = bx / tiles_y;
    const int oy0 = ty_i * TH, ox0 = tx_i * TW;
    const int ty = tid / (TW / PXT), tx = (tid % (TW / PXT)) * PXT; // this thread's row and first column inside the tile
    const int hw = h * w;
    float acc[PXT][COUT];
#pragma unroll
    for (int q = 0; q < PXT; ++q)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[q][co] = 0.f;
    for (int cb = 0; cb < cin; cb += CI_T) {
        // the chunk's patch: every load of a thread in flight before the first LDS store (a rolled loop waits for each load in turn)
        constexpr int NPATCH = CI_T * PH * (TW + 2), NLD = (NPATCH + 255) / 256;
        float stg[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 256;
            const int ci = idx / (PH * (TW + 2));
            const int r = idx - ci * (PH * (TW + 2));
            const int py = r / (TW + 2), px = r - py * (TW + 2);
            const int gy = oy0 + py - 1, gx = ox0 + px - 1;
            stg[j] = 0.f;
            if (idx < NPATCH && cb + ci < cin && gy >= 0 && gy < h && gx >= 0 && gx < w) stg[j] = x[((size_t)n * cin + cb + ci) * hw + gy * w + gx];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 256;
            const int ci = idx / (PH * (TW + 2));
            const int r = idx - ci * (PH * (TW + 2));
            const int py = r / (TW + 2), px = r - py * (TW + 2);
            if (idx < NPATCH) sp[(ci * PH + py) * PW + px] = stg[j];
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CI_T; ++ci) {
            if (cb + ci >= cin) break;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float *row = sp + (ci * PH + ty + ky) * PW + tx;
                const float4 va = *(const float4 *)row;
                const float2 vb = *(const float2 *)(row + 4);
                const float v[6] = {va.x, va.y, va.z, va.w, vb.x, vb.y};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float *wr = wp + ((size_t)(cb + ci) * 9 + ky * 3 + kx) * COUT;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float wv = wr[co];
#pragma unroll
                        for (int q = 0; q < PXT; ++q) acc[q][co] = fmaf(wv, v[q + kx], acc[q][co]);
                    }
                }
            }
        }
    }
    const int oy = oy0 + ty;
    if (oy < h) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float bv = bias[co];
            float o[PXT];
#pragma unroll
            for (int q = 0; q < PXT; ++q) {
                const float t = acc[q][co] + bv;
                o[q] = t > 0.f ? t : t * 0.1f;
            }
            float *dst = y + ((size_t)n * COUT + co) * hw + oy * w + ox0 + tx;
            if (ox0 + tx + PXT <= w && (w & 3) == 0) {
                *(float4 *)dst = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int q = 0; q < PXT; ++q)
                    if (ox0 + tx + q < w) dst[q] = o[q];
            }
        }
    }
}

__global__ void k_pack_weight(const float *__restrict__ w, float *__restrict__ wp, int cout, int cin, int kk)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)cout * cin * kk;
    if (i >= total) return;
    // wp[(ci*kk + k)*cout + co] = w[(co*cin + ci)*kk + k]
    const int co = (int)(i % cout);
    const size_t r = i / cout;
    const int k = (int)(r % kk);
    const int ci = (int)(r / kk);
    wp[i] = w[((size_t)co * cin + ci) * kk + k];
}

// torch.nn.functional.grid_sample(img, grid) as used by backWarp (model.py:289-299):
// bilinear, zeros padding, align_corners=False.  x,y already include the flow.
__device__ __forceinline__ float warp_sample(const float *__restrict__ img, int h, int w, float x, float y)
{
    // model.py:294-295: range -1..1, same float32 operation sequence
    float gx = 2.f * (x / (float)w - 0.5f);
    float gy = 2.f * (y / (float)h - 0.5f);
    // grid_sampler unnormalize, align_corners=False: ((g + 1) * size - 1) / 2
    float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = x0 + 1, y1 = y0 + 1;
    float tx = ix - fx0, ty = iy - fy0;
    // weights as in ATen grid_sampler_2d: nw = (x1-ix)*(y1-iy) ...
    float wx1 = tx, wx0 = 1.f - tx, wy1 = ty, wy0 = 1.f - ty;
    float nw = wx0 * wy0, ne = wx1 * wy0, sw_ = wx0 * wy1, se = wx1 * wy1;
    float out = 0.f;
    if (y0 >= 0 && y0 < h) {
        if (x0 >= 0 && x0 < w) out += img[y0 * w + x0] * nw;
        if (x1 >= 0 && x1 < w) out += img[y0 * w + x1] * ne;
    }
    if (y1 >= 0 && y1 < h) {
        if (x0 >= 0 && x0 < w) out += img[y1 * w + x0] * sw_;
        if (x1 >= 0 && x1 < w) out += img[y1 * w + x1] * se;
    }
    return out;
}

// slomo.py:405-419: x12[(ti*b+bi)] = [I0, I1, F01(2), F10(2), Ft1(2), Ft0(2), g(I1,Ft1), g(I0,Ft0)]
__global__ __launch_bounds__(256) void k_prep(const float *__restrict__ i0, const float *__restrict__ i1,
                                              const float *__restrict__ flow, const float *__restrict__ coef, int n_t,
                                              int b, int h, int w, float *__restrict__ x12)
{
    const int hw = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y; // ti*b + bi
    if (p >= hw) return;
    const int ti = s / b, bi = s - ti * b;
    // fCoeff of slomo.py:406-407, evaluated by the host in Python doubles and rounded to float32
    const float c00 = coef[ti * 6 + 0], c01 = coef[ti * 6 + 1], c10 = coef[ti * 6 + 2], c11 = coef[ti * 6 + 3];
    const float *fl = flow + (size_t)bi * 4 * hw;
    const float f01x = fl[p], f01y = fl[hw + p], f10x = fl[2 * hw + p], f10y = fl[3 * hw + p];
    const float ft0x = c00 * f01x + c01 * f10x, ft0y = c00 * f01y + c01 * f10y;
    const float ft1x = c10 * f01x + c11 * f10x, ft1y = c10 * f01y + c11 * f10y;
    const int py = p / w, px = p - py * w;
    const float *im0 = i0 + (size_t)bi * hw, *im1 = i1 + (size_t)bi * hw;
    const float g0 = warp_sample(im0, h, w, (float)px + ft0x, (float)py + ft0y);
    const float g1 = warp_sample(im1, h, w, (float)px + ft1x, (float)py + ft1y);
    float *o = x12 + (size_t)s * 12 * hw + p;
    o[0] = im0[p];
    o[hw] = im1[p];
    o[2 * hw] = f01x; o[3 * hw] = f01y;
    o[4 * hw] = f10x; o[5 * hw] = f10y;
    o[6 * hw] = ft1x; o[7 * hw] = ft1y;
    o[8 * hw] = ft0x; o[9 * hw] = ft0y;
    o[10 * hw] = g1;
    o[11 * hw] = g0;
}

// slomo.py:421-433
__global__ __launch_bounds__(256) void k_fuse(const float *__restrict__ i0, const float *__restrict__ i1,
                                              const float *__restrict__ x12, const float *__restrict__ intrp,
                                              const float *__restrict__ coef, int n_t, int b, int h, int w,
                                              float *__restrict__ out)
{
    const int hw = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (p >= hw) return;
    const int ti = s / b, bi = s - ti * b;
    const float *x = x12 + (size_t)s * 12 * hw + p;
    const float *q = intrp + (size_t)s * 5 * hw + p;
    const float ft0x = q[0] + x[8 * hw], ft0y = q[hw] + x[9 * hw];
    const float ft1x = q[2 * hw] + x[6 * hw], ft1y = q[3 * hw] + x[7 * hw];
    const float v0 = 1.f / (1.f + expf(-q[4 * hw])); // torch.sigmoid
    const float v1 = 1.f - v0;
    const int py = p / w, px = p - py * w;
    const float g0 = warp_sample(i0 + (size_t)bi * hw, h, w, (float)px + ft0x, (float)py + ft0y);
    const float g1 = warp_sample(i1 + (size_t)bi * hw, h, w, (float)px + ft1x, (float)py + ft1y);
    const float w0 = coef[ti * 6 + 4], w1 = coef[ti * 6 + 5]; // wCoeff = [1 - t, t]
    out[(size_t)s * hw + p] = (w0 * v0 * g0 + w1 * v1 * g1) / (w0 * v0 + w1 * v1);
}

// largest |x| of a tensor into a range slot (the network input of v2e_unet_forward)
__global__ __launch_bounds__(256) void k_amax(const float *__restrict__ x, long long n, uint32_t *__restrict__ slot)
{
    uint32_t m = 0u;
    const long long n4 = n >> 2;
    const float4 *x4 = (const float4 *)x;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x4[i];
        amax_fold(m, v.x); amax_fold(m, v.y); amax_fold(m, v.z); amax_fold(m, v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) amax_fold(m, x[(n4 << 2) + threadIdx.x]);
    amax_commit(slot, m);
}

__global__ void k_zero_u32(unsigned *__restrict__ p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// slomo.py:359-368 (auto_upsample): the largest flow magnitude of a batch.  sqrt is monotone, so the kernel reduces the
// squared speed vx*vx + vy*vy (two products and one sum, each rounded: contraction is off for this file) of both flow pairs
// and the host takes ONE float32 sqrt of the maximum -- the same number as max(sqrt(...)) over the planes.  Non-negative floats
// order like their bit patterns: one atomicMax per wave.
__global__ __launch_bounds__(256) void k_max_speed2(const float *__restrict__ flow, int b, int hw, unsigned *__restrict__ out_bits)
{
    float m = 0.f;
    const long long n = (long long)b * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int bi = (int)(i / hw);
        const int p = (int)(i - (long long)bi * hw);
        const float *fl = flow + (size_t)bi * 4 * hw + p;
        const float x0 = fl[0], y0 = fl[hw], x1 = fl[2 * (size_t)hw], y1 = fl[3 * (size_t)hw];
        const float s0 = x0 * x0 + y0 * y0;
        const float s1 = x1 * x1 + y1 * y1;
        // torch.max propagates NaN; a NaN flow makes int(np.ceil(nan)) raise in the reference: report it as NaN too
        if (s0 != s0 || s1 != s1) m = __int_as_float(0x7fc00000);
        else if (m == m) m = fmaxf(m, fmaxf(s0, s1));
    }
    unsigned bits = (m != m) ? 0xffffffffu : __float_as_uint(m);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)bits, o, 64);
        bits = other > bits ? other : bits;
    }
    if ((threadIdx.x & 63) == 0 && bits != 0u) atomicMax(out_bits, bits);
}

// ---------------------------------------------------------------- dispatch
template <int KS, int CI_T, int CT, int PT, int WP, int TW, int PRE>
static void launch_conv(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    constexpr int TH = WP * PT * 32 / TW;
    a.tiles_x = (a.w_ + TW - 1) / TW;
    a.tiles_y = (a.h + TH - 1) / TH;
    dim3 grid((unsigned)(a.n * a.tiles_x * a.tiles_y), (unsigned)((a.cout + CT * 32 - 1) / (CT * 32)));
    k_conv<KS, CI_T, CT, PT, WP, TW, PRE><<<grid, WP * 64, 0, s>>>(a);
}

// two whole samples per tile (sample height == half the tile height)
template <int KS, int CI_T, int CT, int PT, int WP, int TW>
static void launch_conv_stacked(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    a.tiles_x = (a.w_ + TW - 1) / TW;
    a.tiles_y = 1;
    dim3 grid((unsigned)(((a.n + 1) / 2) * a.tiles_x), (unsigned)((a.cout + CT * 32 - 1) / (CT * 32)));
    k_conv<KS, CI_T, CT, PT, WP, TW, 0, 2><<<grid, WP * 64, 0, s>>>(a);
}

template <int KS, int CI_T, int CT, int PRE>
static void launch_conv_tw(const ConvArgs &a, hipStream_t s)
{
    // pixel tile: prefer power-of-two widths that divide W (4 waves, one per SIMD); exact 20- / 10-wide
    // tiles for the 16x20 and 8x10 levels of a 320x256 input; 32-wide with masking otherwise
    if (a.w_ % 32 == 0) launch_conv<KS, CI_T, CT, 2, 4, 32, PRE>(a, s);
    else if (a.w_ % 16 == 0) launch_conv<KS, CI_T, CT, 2, 4, 16, PRE>(a, s);
    else if (a.w_ % 8 == 0) launch_conv<KS, CI_T, CT, 2, 4, 8, PRE>(a, s);
    else if (a.w_ % 20 == 0 && a.h % 8 == 0) {
        // 20-wide levels (16x20 of a 320x256 input): 8x20-pixel tiles, 5 waves.  What decides here is how evenly
        // the workgroups spread over 256 CUs: 40 samples x 8 blocks of 64 channels = 320 ten-wave workgroups
        // leave a quarter of the CUs with twice the work (68 TF/s); 1280 five-wave workgroups of 32 channels
        // reach 92 TF/s.  The wider channel block only pays once there are >= 8 workgroups per CU.
        const long wgs64 = (long)a.n * (a.h / 8) * (a.cout / 64);
        if (CT == 2 && wgs64 < 2048) launch_conv<KS, CI_T, 1, 1, 5, 20, PRE>(a, s);
        else launch_conv<KS, CI_T, CT, 1, 5, 20, PRE>(a, s);
    }
    else if (a.w_ % 10 == 0) {
        // 10-wide levels (8x10 of a 320x256 input): a 16x10 five-wave tile is two samples; one sample per tile
        // would leave half of every MFMA column masked
        if (KS == 3 && PRE == 0 && a.h == 8) launch_conv_stacked<3, CI_T, 1, 1, 5, 10>(a, s);
        else launch_conv<KS, CI_T, CT, 1, 5, 10, PRE>(a, s);
    }
    else launch_conv<KS, CI_T, CT, 2, 4, 32, PRE>(a, s);
}

static int conv_dispatch(const ConvArgs &a, int ks, int pre, hipStream_t s)
{
    // 64-channel tiles only when that still leaves >= 16 workgroups per CU: measured on the 40-sample interpolation
    // UNet, 32-channel tiles (twice the workgroups, ~30 fewer VGPRs) are faster at every level (forward 96.6 -> 104 TF/s)
    const long px_tiles = (long)a.n * ((a.h * a.w_ + 255) / 256);
    const bool wide = a.cout >= 64 && px_tiles * (a.cout / 64) >= 4096;
    if (ks == 7) {
        if (pre != 0) return V2E_EINVAL;
        if (a.cin % 4 == 0) launch_conv<7, 4, 1, 2, 4, 32, 0>(a, s);
        else launch_conv<7, 2, 1, 2, 4, 32, 0>(a, s);
    } else if (ks == 5) {
        if (pre == 2) return V2E_EINVAL;
        if (wide) { if (pre) launch_conv<5, 4, 2, 2, 4, 32, 1>(a, s); else launch_conv<5, 4, 2, 2, 4, 32, 0>(a, s); }
        else { if (pre) launch_conv<5, 4, 1, 2, 4, 32, 1>(a, s); else launch_conv<5, 4, 1, 2, 4, 32, 0>(a, s); }
    } else if (ks == 3) {
        if (wide) {
            if (pre == 0) launch_conv_tw<3, 8, 2, 0>(a, s);
            else if (pre == 1) launch_conv_tw<3, 8, 2, 1>(a, s);
            else launch_conv_tw<3, 8, 2, 2>(a, s);
        } else {
            if (pre == 0) launch_conv_tw<3, 8, 1, 0>(a, s);
            else if (pre == 1) launch_conv_tw<3, 8, 1, 1>(a, s);
            else launch_conv_tw<3, 8, 1, 2>(a, s);
        }
    } else {
        return V2E_EINVAL;
    }
    return 0;
}

// split-bf16 path (slomo_s3.h): returns 1 when no tile fits (the caller falls back to the f32-MFMA kernel)
// pre-split input (pre == 3): 3x3 layers only (dev: measured against in-kernel splitting by scripts/conv_s3_check)
static int conv_dispatch_s3_presplit(const ConvArgs &a, int ks, hipStream_t s)
{
    if (ks != 3) return 1;
    if (a.w_ % 32 == 0) return launch_conv_s3<3, 1, 2, 4, 32, 1, 2>(a, s);
    if (a.w_ % 16 == 0) return launch_conv_s3<3, 1, 2, 4, 16, 1, 2>(a, s);
    if (a.w_ % 8 == 0) return launch_conv_s3<3, 1, 2, 4, 8, 1, 2>(a, s);
    return 1;
}

static int conv_dispatch_s3(const ConvArgs &a, int ks, hipStream_t s)
{
    // 32-channel tiles (60 KB of LDS: two workgroups per CU, so one stages while the other multiplies) measured faster
    // than 64-channel ones at every level (16 samples: 3x3 143-174 vs 121-153 TF/s, 5x5 172-183 vs 146-160)
    static const int variant = getenv("V2E_AMD_S3_VARIANT") ? atoi(getenv("V2E_AMD_S3_VARIANT")) : 0; // dev: tile choice
    const bool c64 = a.cout % 64 == 0 && variant == 6;
    if (a.np == 2) { // two float16 pieces, three products (conv_math "fp16x2"): the 32 x 64 tiles, two workgroups per CU
        // 3x3, cout % 64 == 0: 64 x 64 register tiles with one kernel row of weights resident at a time (34 KB of LDS): a third
        // fewer LDS operand bytes per multiply than the 32 x 64 tile, bit-identical output, 4 % per forward (variant 18: off).
        // With half the multiplies of the three-piece math the matrix pipe is far from the rate at which four accumulator
        // chains x two waves per SIMD are slow (slomo_s3p.h), so the tile can keep two workgroups per CU.
        // (64 x 128 register tiles -- CT 2, PT 4: half the LDS operand bytes per multiply again -- measured slower in round 4, at two
        //  waves per SIMD (15 spilled registers) and at one with two operand sets: 128x160 803 -> 950 us, forward 17.7 -> 18.2 / 18.5 ms)
        if (ks == 3 && a.cout % 64 == 0 && variant != 18) {
            if (a.w_ % 32 == 0) return launch_conv_s3<3, 2, 2, 4, 32, 1, 0, 1, 2>(a, s);
            if (a.w_ % 16 == 0) return launch_conv_s3<3, 2, 2, 4, 16, 1, 0, 1, 2>(a, s);
            if (a.w_ % 8 == 0) return launch_conv_s3<3, 2, 2, 4, 8, 1, 0, 1, 2>(a, s);
            // (the 20-wide level's 5-wave tiles with 64 channels: measured slower, 18.39 against 17.95 ms per forward)
        }
        if (ks == 3) {
            // (dev, round 6: 32 channels x 128 pixels per wave on the 32-channel full-resolution layers -- one weight operand read feeds
            // four multiplies; variants 31 / 32)
            // 32-channel layers (up5.*: 64 -> 32 at full resolution): 32 channels x 128 pixels per wave with one kernel row of weights
            // resident at a time -- a weight operand read from LDS feeds four multiplies instead of two: 1262 / 1242 -> 1149 / 1134 us per
            // layer at 80 samples (profiles/r06_slomo_per_layer.txt; variant 33: the 32 x 64 tile of rounds 3-5; all weights resident
            // with this tile, variant 31, measured slower than either)
            if (a.w_ % 32 == 0 && a.h % 16 == 0 && variant == 31) return launch_conv_s3<3, 1, 4, 4, 32, 1, 0, 0, 2>(a, s);
            if (a.w_ % 32 == 0 && a.h % 16 == 0 && variant != 33) return launch_conv_s3<3, 1, 4, 4, 32, 1, 0, 1, 2>(a, s);
            if (a.w_ % 32 == 0) return launch_conv_s3<3, 1, 2, 4, 32, 1, 0, 0, 2>(a, s);
            if (a.w_ % 16 == 0) return launch_conv_s3<3, 1, 2, 4, 16, 1, 0, 0, 2>(a, s);
            if (a.w_ % 8 == 0) return launch_conv_s3<3, 1, 2, 4, 8, 1, 0, 0, 2>(a, s);
            // 20-wide levels of 16 rows (the 16 x 20 level of a 320 x 256 input): one whole sample per five-wave workgroup, two pixel
            // tiles per wave -- a weight operand read from LDS feeds two multiplies instead of one (forward 18.33 -> 18.09 ms, A/B x 3; variant 24:
            // the 8 x 20 tiles of round 3; 64 output channels per workgroup on top of it measured slower, 18.46)
            if (a.w_ % 20 == 0 && a.h % 16 == 0 && variant != 24) return launch_conv_s3<3, 1, 2, 5, 20, 1, 0, 0, 2>(a, s);
            if (a.w_ % 20 == 0 && a.h % 8 == 0) return launch_conv_s3<3, 1, 1, 5, 20, 1, 0, 0, 2>(a, s);
            if (a.w_ == 10 && a.h <= 16) return launch_conv_s3<3, 1, 1, 5, 10, 1, 0, 0, 2>(a, s); // the 8 x 10 level: half of a 16-row tile masked
            return 1;
        }
        if (ks == 5 && a.w_ % 32 == 0 && a.cout % 64 == 0 && variant != 18) return launch_conv_s3<5, 2, 2, 4, 32, 1, 0, 1, 2>(a, s); // 64 x 64 tiles: 1.5 % more
        if (ks == 5 && a.w_ % 32 == 0) return launch_conv_s3<5, 1, 2, 4, 32, 1, 0, 0, 2>(a, s);
        // (dev, round 6: 32 channels x 128 pixels per wave on the 7x7 layers too; variant 34)
        if (ks == 7 && a.w_ % 32 == 0 && a.h % 16 == 0 && variant == 34) return a.cin % 16 == 0 ? launch_conv_s3<7, 1, 4, 4, 32, 1, 0, 0, 2>(a, s) : launch_conv_s3<7, 1, 4, 4, 32, 1, 1, 0, 2>(a, s);
        if (ks == 7 && a.w_ % 32 == 0) return a.cin % 16 == 0 ? launch_conv_s3<7, 1, 2, 4, 32, 1, 0, 0, 2>(a, s) : launch_conv_s3<7, 1, 2, 4, 32, 1, 1, 0, 2>(a, s);
        return 1;
    }
    if (ks == 3) {
        // the software-pipelined one-wave-per-SIMD kernel where the layer has one of its shapes (bit-identical output; 3-6 % less
        // time per layer, 1.5 % per forward: the chip is at its power limit in these kernels -- the shader clock reads 1.63-1.75 GHz
        // of 2.4 -- and the 64 x 64 register tile moves fewer LDS bytes per multiply; slomo_s3p.h); variant 12: off
        if ((variant == 0 || variant == 11) && conv_dispatch_s3p(a, ks, s) == 0) return 0;
        if (a.w_ % 32 == 0) {
            if (variant == 2 && a.cout % 64 == 0) return launch_conv_s3<3, 2, 2, 8, 32>(a, s);
            if (variant == 3) return launch_conv_s3<3, 1, 4, 4, 32>(a, s);
            if (variant == 4 && a.cout % 64 == 0) return launch_conv_s3<3, 2, 4, 4, 32>(a, s);
            if (variant == 7) return launch_conv_s3<3, 1, 2, 4, 32, 2>(a, s);
            if (variant == 8 && a.cout % 64 == 0) return launch_conv_s3<3, 2, 2, 4, 32, 1, 0, 1>(a, s);
            if (variant == 9 && a.cout % 64 == 0) return launch_conv_s3<3, 2, 2, 4, 32, 2, 0, 1>(a, s);
            if (variant == 10) return launch_conv_s3<3, 1, 2, 4, 32, 1, 0, 1>(a, s);
            return c64 ? launch_conv_s3<3, 2, 2, 4, 32>(a, s) : launch_conv_s3<3, 1, 2, 4, 32>(a, s);
        }
        if (a.w_ % 16 == 0) return c64 ? launch_conv_s3<3, 2, 2, 4, 16>(a, s) : (variant == 7 ? launch_conv_s3<3, 1, 2, 4, 16, 2>(a, s) : launch_conv_s3<3, 1, 2, 4, 16>(a, s));
        if (a.w_ % 8 == 0) return c64 ? launch_conv_s3<3, 2, 2, 4, 8>(a, s) : (variant == 7 ? launch_conv_s3<3, 1, 2, 4, 8, 2>(a, s) : launch_conv_s3<3, 1, 2, 4, 8>(a, s));
        // (the 16 x 20 tile of the two-piece math measured slower here: 26.4 -> 27.0 ms per forward)
        if (a.w_ % 20 == 0 && a.h % 8 == 0) return c64 ? launch_conv_s3<3, 2, 1, 5, 20>(a, s) : launch_conv_s3<3, 1, 1, 5, 20>(a, s);
        // the 8 x 10 level: a 16 x 10 tile with its lower half masked is still 1.5x the f32 matrix-core kernel's two stacked samples
        if (a.w_ == 10 && a.h <= 16 && variant != 13) return launch_conv_s3<3, 1, 1, 5, 10>(a, s);
        return 1;
    }
    if (ks == 5 && a.w_ % 32 == 0) return c64 ? launch_conv_s3<5, 2, 2, 4, 32>(a, s) : launch_conv_s3<5, 1, 2, 4, 32>(a, s);
    if (ks == 7 && a.w_ % 32 == 0) return a.cin % 16 == 0 ? launch_conv_s3<7, 1, 2, 4, 32>(a, s) : launch_conv_s3<7, 1, 2, 4, 32, 1, 1>(a, s);
    return 1;
}

} // namespace

extern "C" {

// dev tool (not part of the public header; scripts/conv_s3_check): switch k_conv_s3p's per-step timeline of workgroup 0 on
// (out == NULL) or read it back: pairs (shader clock, 100 MHz wall clock), terminated by a zero
int v2e_slomo_debug_s3p_timeline(unsigned long long *out, int n_pairs)
{
    if (!out) {
        g_s3p_timeline_on = 1;
        return 0;
    }
    V2E_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s3p_timeline), sizeof(unsigned long long) * 2 * (size_t)(n_pairs < 512 ? n_pairs : 512)));
    return 0;
}


int v2e_pack_conv_weight_s3(const float *w_oihw, void *w_s3, int cout, int cin, int k, void *stream)
{
    V2E_REQUIRE(w_oihw && w_s3 && cout > 0 && cin > 0 && k > 0, "bad pack args");
    const size_t total = (size_t)((cin + 15) / 16) * 2 * k * k * cout; // [ceil(cin/16)][k*k][3][2][cout] 16-byte units / 3

    k_pack_weight_s3<3><<<v2e_cdiv((int64_t)total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, (uint4 *)w_s3, cout, cin, k * k);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_conv_set_range_flag(int *device_flag)
{
    g_conv_range_flag = device_flag;
    return 0;
}

int v2e_pack_conv_weight_h2(const float *w_oihw, void *w_h2, int cout, int cin, int k, int scale_log2, void *stream)
{
    V2E_REQUIRE(w_oihw && w_h2 && cout > 0 && cin > 0 && k > 0 && scale_log2 >= 0 && scale_log2 <= 60, "bad pack args");
    const size_t total = (size_t)((cin + 15) / 16) * 2 * k * k * cout; // one thread per (chunk, tap, channel group, cout): NP units
    k_pack_weight_s3<2><<<v2e_cdiv((int64_t)total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, (uint4 *)w_h2, cout, cin, k * k,
                                                                                        ldexpf(1.0f, scale_log2));
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_split3_nchw(const float *x, void *xs, int n, int c, int h, int w, void *stream)
{
    V2E_REQUIRE(x && xs && n > 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0, "bad split args (channels must be a multiple of 8)");
    const long long nc8 = (long long)n * (c / 8);
    k_split3_nchw<<<v2e_cdiv(nc8 * h * w, 256), 256, 0, (hipStream_t)stream>>>(x, (uint4 *)xs, nc8, h * w);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_pack_conv_weight(const float *w_oihw, float *w_packed, int cout, int cin, int k, void *stream)
{
    V2E_REQUIRE(w_oihw && w_packed && cout > 0 && cin > 0 && k > 0, "bad pack args");
    const size_t total = (size_t)cout * cin * k * k;
    k_pack_weight<<<v2e_cdiv((int64_t)total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, w_packed, cout, cin, k * k);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_conv2d_lrelu(const float *x0, int c0, const float *x1, int c1, int pre, const v2e_conv_desc *conv, float *y,
                     int n, int h, int w, void *stream)
{
    V2E_REQUIRE(x0 && conv && conv->weight && conv->bias && y, "null conv arg");
    V2E_REQUIRE(c0 + c1 == conv->cin && (c1 == 0 || x1), "channel split does not match the layer");
    V2E_REQUIRE(pre == 0 || c1 == 0, "fused pool/upsample takes a single source");
    if (pre == 3) { // x0 is a pre-split tensor (v2e_split3_nchw)
        V2E_REQUIRE(conv->weight_s3 && conv->cin % 16 == 0 && conv->cout % 32 == 0, "pre-split input needs split weights, cin % 16 == 0, cout % 32 == 0");
        ConvArgs a;
        a.x0 = x0; a.x1 = nullptr; a.c0 = c0; a.c1 = 0; a.w = conv->weight; a.bias = conv->bias; a.y = y;
        a.n = n; a.h = h; a.w_ = w; a.cin = conv->cin; a.cout = conv->cout; a.tiles_x = a.tiles_y = 0;
        V2E_REQUIRE((conv->split_kind & 0xFF) != 2, "pre-split input is three bf16 pieces");
        a.ws3 = conv->weight_s3; a.xs_plane = (long long)n * (c0 / 8) * h * w; a.np = 3; a.tl_on = 0; a.out_scale = 1.0f; a.ovf = nullptr;
        a.am_in0 = a.am_in1 = nullptr; a.am_out = g_am_out; a.ypool = nullptr;
        const int r3 = conv_dispatch_s3_presplit(a, conv->ksize, (hipStream_t)stream);
        V2E_REQUIRE(r3 == 0, "no pre-split tile for this layer shape");
        V2E_HIP(hipGetLastError());
        return 0;
    }
    V2E_REQUIRE(pre != 2 || (h % 2 == 0 && w % 2 == 0), "upsample target must be even");
    V2E_REQUIRE(conv->cin % 2 == 0, "cin must be even");
    V2E_REQUIRE(c1 == 0 || c0 % 8 == 0, "first concat source must be a multiple of 8 channels");
    V2E_REQUIRE(h < 4096 && w < 4096, "image too large for the staging plan");
    {
        const int k = conv->ksize, ci = conv->cin;
        V2E_REQUIRE((k == 3 && ci % 8 == 0) || (k == 5 && ci % 4 == 0) || (k == 7 && ci % 2 == 0),
                    "cin must be a multiple of the channel chunk (8 for 3x3, 4 for 5x5, 2 for 7x7)");
    }
    ConvArgs a;
    a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1;
    a.w = conv->weight; a.bias = conv->bias; a.y = y;
    a.n = n; a.h = h; a.w_ = w; a.cin = conv->cin; a.cout = conv->cout;
    a.tiles_x = a.tiles_y = 0;
    a.ws3 = conv->weight_s3;
    a.np = (conv->split_kind & 0xFF) == 2 ? 2 : 3;
    a.out_scale = ldexpf(1.0f, -(conv->split_kind >> 8));
    a.ovf = g_conv_range_flag;
    a.tl_on = 0;
    a.am_in0 = g_am_in0; a.am_in1 = g_am_in1; a.am_out = g_am_out; a.ypool = nullptr;
    if (conv->ksize == 3 && pre == 0 && c1 == 0 && (conv->cout == 4 || conv->cout == 5)) {
        const int tiles_x = (w + 63) / 64, tiles_y = (h + 15) / 16;
        dim3 grid((unsigned)(n * tiles_x * tiles_y));
        if (conv->cout == 4) k_conv3x3_small<4><<<grid, 256, 0, (hipStream_t)stream>>>(x0, conv->weight, conv->bias, y, n, conv->cin, h, w, tiles_x, tiles_y);
        else k_conv3x3_small<5><<<grid, 256, 0, (hipStream_t)stream>>>(x0, conv->weight, conv->bias, y, n, conv->cin, h, w, tiles_x, tiles_y);
        V2E_HIP(hipGetLastError());
        return 0;
    }
    constexpr bool s3_off = false;
    // split-bf16 kernel: whole 16-channel chunks -- or the 12-channel 7x7 conv1, whose last chunk is padded with zeros
    // (a third of its multiplies wasted and still 1.4x the f32 kernel; the 2-channel conv1 of the flow UNet is not worth it)
    const bool s3_cin = conv->cin % 16 == 0 || (conv->ksize == 7 && c1 == 0 && conv->cin >= 8);
    if (a.ws3 && !s3_off && pre == 0 && s3_cin && conv->cout % 32 == 0 && (c1 == 0 || c0 % 16 == 0)) {
        const int r3 = conv_dispatch_s3(a, conv->ksize, (hipStream_t)stream);
        if (r3 == 0) { V2E_HIP(hipGetLastError()); return 0; }
        if (r3 != 1) return r3;
    }
    int rc = conv_dispatch(a, conv->ksize, pre, (hipStream_t)stream);
    if (rc) { v2e_set_error("unsupported conv: k=%d pre=%d", conv->ksize, pre); return rc; }
    V2E_HIP(hipGetLastError());
    return 0;
}

// workspace layout (floats): s1 s2 s3 s4 s5 | tA tB   (model.py:198-226)
int64_t v2e_unet_workspace_bytes(int n, int h, int w, int cin)
{
    (void)cin;
    const int64_t hw = (int64_t)h * w;
    // s1 32hw, s2 16hw, s3 8hw, s4 4hw, s5 2hw, two temporaries of 32hw, one of 64hw for the
    // pooled / upsampled input of the first conv of each down / up block
    // + the range slots of the forward pass (32 words: the input's and every convolution's largest |output|)
    return (int64_t)n * hw * (32 + 16 + 8 + 4 + 2 + 32 + 32 + 64) * (int64_t)sizeof(float) + 256;
}

int v2e_unet_forward(const float *x, int cin, const v2e_conv_desc *cv, int cout, float *y, int n, int h, int w,
                     void *workspace, void *stream)
{
    V2E_REQUIRE(x && cv && y && workspace, "null unet arg");
    V2E_REQUIRE(h % 32 == 0 && w % 32 == 0, "UNet input must be a multiple of 32 (dataloader.py:122-123)");
    V2E_REQUIRE(cv[0].cin == cin && cv[22].cout == cout, "descriptor list does not match cin/cout");
    const int64_t hw = (int64_t)h * w;
    float *ws = (float *)workspace;
    float *s1 = ws; float *s2 = s1 + n * hw * 32; float *s3 = s2 + n * hw * 16; float *s4 = s3 + n * hw * 8;
    float *s5 = s4 + n * hw * 4; float *tA = s5 + n * hw * 2; float *tB = tA + n * hw * 32; float *tU = tB + n * hw * 32;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // range slots: am[0] = the input, am[1 + i] = the output of convolution i.  Tracked only when some layer runs on two
    // float16 pieces (it is what their operand scaling reads); avg_pool2d / bilinear x2 cannot exceed their input's maximum, so
    // a convolution behind one of them reads the slot of the convolution before it.
    uint32_t *am = (uint32_t *)(tU + n * hw * 64);
    bool track = false;
    for (int i = 0; i < 23; ++i) track = track || ((cv[i].split_kind & 0xFF) == 2 && cv[i].weight_s3);
    struct AmReset { ~AmReset() { g_am_in0 = g_am_in1 = nullptr; g_am_out = nullptr; g_pool_out = nullptr; } } am_reset;
    if (track) {
        k_zero_u32<<<1, 64, 0, st>>>(am, 32);
        const long long nx = (long long)n * cin * hw;
        long long nb = (nx / 4 + 255) / 256;
        k_amax<<<(unsigned)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb)), 256, 0, st>>>(x, nx, am);
    }
#define CONV(X0, C0, X1, C1, PRE, IDX, Y, HH, WW)                                              \
    do {                                                                                        \
        if (track) { g_am_in0 = am + am_src0; g_am_in1 = (X1) ? am + am_src1 : nullptr; g_am_out = am + 1 + (IDX); } \
        rc = v2e_conv2d_lrelu((X0), (C0), (X1), (C1), (PRE), &cv[(IDX)], (Y), n, (HH), (WW), stream); \
        if (rc) return rc;                                                                      \
        am_src0 = 1 + (IDX);                                                                    \
    } while (0)
    static const int fuse_pool = getenv("V2E_AMD_FUSE_POOL") ? atoi(getenv("V2E_AMD_FUSE_POOL")) : 0;
    int am_src0 = 0, am_src1 = 0; // slots of the producers of the next convolution's x0 / x1
    // producer ops as their own streaming passes: measured faster than fusing them into the conv
    // loader (the fused bilinear fetch costs 4 gathers + address math per staged element)
#define POOL(X, C, HH, WW) /* (HH,WW) = output size */                                              \
    do {                                                                                         \
        if ((WW) % 4 == 0) k_avgpool2<<<v2e_cdiv((int64_t)n * (C) * (HH) * ((WW) / 4), 256), 256, 0, st>>>((X), tU, (long long)n * (C), (HH), (WW)); \
        else k_avgpool2_scalar<<<v2e_cdiv((int64_t)n * (C) * (HH) * (WW), 256), 256, 0, st>>>((X), tU, (long long)n * (C), (HH), (WW)); \
    } while (0)
#define UPS(X, C, HH, WW)                                                                        \
    do {                                                                                         \
        if ((WW) % 4 == 0) k_upsample2<<<v2e_cdiv((int64_t)n * (C) * ((HH) / 2 + 1) * ((WW) / 4), 256), 256, 0, st>>>((X), tU, (long long)n * (C), (HH), (WW)); \
        else k_upsample2_scalar<<<v2e_cdiv((int64_t)n * (C) * (HH) * (WW), 256), 256, 0, st>>>((X), tU, (long long)n * (C), (HH), (WW)); \
    } while (0)
    // conv1, conv2
    CONV(x, cin, nullptr, 0, 0, 0, tA, h, w);
    // V2E_AMD_FUSE_POOL=1: the pooled copy of a skip tensor comes out of the producing convolution's epilogue where its tile allows
    // (the two largest of the five poolings: 0.28 of the 0.33 ms the pooling passes take at 80 samples).  Built and bit-identical
    // (tests/test_slomo_gpu.py), measured round 4, A/B x 3 at 80 samples: 17.83 ms fused against 17.80 ms with the separate
    // passes (bf16x3: 26.30 both) -- the convolutions are at the chip's power limit, and what the epilogue adds costs what the
    // streaming pass did.  Off by default.
#define CONV_POOLED(X0, C0, IDX, Y, CO, HH, WW) /* Y = conv(X0) at (HH, WW); tU = avg_pool2d(Y, 2) */ \
    do {                                                                                             \
        g_pool_out = fuse_pool ? tU : nullptr;                                                       \
        CONV((X0), (C0), nullptr, 0, 0, (IDX), (Y), (HH), (WW));                                     \
        if (g_pool_out || !fuse_pool) { g_pool_out = nullptr; POOL((Y), (CO), (HH) / 2, (WW) / 2); } /* not taken: its own pass */ \
    } while (0)
    CONV_POOLED(tA, 32, 1, s1, 32, h, w);
    // down1..down5
    CONV(tU, 32, nullptr, 0, 0, 2, tA, h / 2, w / 2);
    CONV_POOLED(tA, 64, 3, s2, 64, h / 2, w / 2);
    CONV(tU, 64, nullptr, 0, 0, 4, tA, h / 4, w / 4);
    CONV(tA, 128, nullptr, 0, 0, 5, s3, h / 4, w / 4);
    POOL(s3, 128, h / 8, w / 8);
    CONV(tU, 128, nullptr, 0, 0, 6, tA, h / 8, w / 8);
    CONV(tA, 256, nullptr, 0, 0, 7, s4, h / 8, w / 8);
    POOL(s4, 256, h / 16, w / 16);
    CONV(tU, 256, nullptr, 0, 0, 8, tA, h / 16, w / 16);
    CONV(tA, 512, nullptr, 0, 0, 9, s5, h / 16, w / 16);
    POOL(s5, 512, h / 32, w / 32);
    CONV(tU, 512, nullptr, 0, 0, 10, tA, h / 32, w / 32);
    CONV(tA, 512, nullptr, 0, 0, 11, tB, h / 32, w / 32);
    // up1..up5: skip concat fused into conv2 (x first, skip second)
    static const int fuse_up = getenv("V2E_AMD_FUSE_UP") ? atoi(getenv("V2E_AMD_FUSE_UP")) : 0; // dev: bit u-1 = fuse the bilinear x2 of up<u> into its conv loader
#define UPCONV(U, X, C, IDX, Y, HH, WW)                                                    \
    do {                                                                                    \
        if (fuse_up & (1 << ((U) - 1))) CONV((X), (C), nullptr, 0, 2, (IDX), (Y), (HH), (WW)); \
        else { UPS((X), (C), (HH), (WW)); CONV(tU, (C), nullptr, 0, 0, (IDX), (Y), (HH), (WW)); } \
    } while (0)
    UPCONV(1, tB, 512, 12, tA, h / 16, w / 16);
    am_src1 = 1 + 9;  // s5 = the output of convolution 9 (down4.conv2)
    CONV(tA, 512, s5, 512, 0, 13, tB, h / 16, w / 16);
    UPCONV(2, tB, 512, 14, tA, h / 8, w / 8);
    am_src1 = 1 + 7;  // s4
    CONV(tA, 256, s4, 256, 0, 15, tB, h / 8, w / 8);
    UPCONV(3, tB, 256, 16, tA, h / 4, w / 4);
    am_src1 = 1 + 5;  // s3
    CONV(tA, 128, s3, 128, 0, 17, tB, h / 4, w / 4);
    UPCONV(4, tB, 128, 18, tA, h / 2, w / 2);
    am_src1 = 1 + 3;  // s2
    CONV(tA, 64, s2, 64, 0, 19, tB, h / 2, w / 2);
    UPCONV(5, tB, 64, 20, tA, h, w);
    am_src1 = 1 + 1;  // s1
    CONV(tA, 32, s1, 32, 0, 21, tB, h, w);
    // conv3 (+ leaky relu, model.py:225)
    CONV(tB, 32, nullptr, 0, 0, 22, y, h, w);
#undef UPCONV
#undef CONV_POOLED
#undef POOL
#undef UPS
#undef CONV
    return 0;
}

int v2e_slomo_prep(const float *i0, const float *i1, const float *flow, const float *tcoef, int n_t, int b, int h, int w,
                   float *x12, void *stream)
{
    const float *t = tcoef;
    V2E_REQUIRE(i0 && i1 && flow && t && x12 && n_t > 0 && b > 0, "bad prep args");
    dim3 grid(v2e_cdiv((int64_t)h * w, 256), n_t * b);
    k_prep<<<grid, 256, 0, (hipStream_t)stream>>>(i0, i1, flow, t, n_t, b, h, w, x12);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_slomo_fuse(const float *i0, const float *i1, const float *x12, const float *intrp, const float *tcoef, int n_t,
                   int b, int h, int w, float *out, void *stream)
{
    const float *t = tcoef;
    V2E_REQUIRE(i0 && i1 && x12 && intrp && t && out && n_t > 0 && b > 0, "bad fuse args");
    dim3 grid(v2e_cdiv((int64_t)h * w, 256), n_t * b);
    k_fuse<<<grid, 256, 0, (hipStream_t)stream>>>(i0, i1, x12, intrp, t, n_t, b, h, w, out);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_slomo_max_speed2(const float *flow, int b, int h, int w, uint32_t *out_bits, void *stream)
{
    V2E_REQUIRE(flow && out_bits && b > 0 && h > 0 && w > 0, "bad max_speed args");
    hipStream_t s = (hipStream_t)stream;
    k_zero_u32<<<1, 64, 0, s>>>(out_bits, 1);
    const long long n = (long long)b * h * w;
    int blocks = v2e_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    k_max_speed2<<<blocks, 256, 0, s>>>(flow, b, h * w, out_bits);
    V2E_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
