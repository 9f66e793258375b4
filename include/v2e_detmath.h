/*
 * v2e_detmath.h -- counter-based RNG (Philox4x32-10) and *deterministic* f32 math
 * shared verbatim by the gfx950 HIP kernels (v2e_amd/csrc) and the CPU oracle
 * (oracle/emu_oracle.c).
 *
 * Why this exists: the reference draws its per-pixel noise from torch's global
 * generators (v2ecore/emulator.py:459-471, 501-505; emulator_utils.py:122-124,
 * 338-341).  A GPU kernel cannot replay MT19937, so "philox mode" defines its own
 * counter-based streams.  To make device and host produce the SAME bits, every
 * function below uses only IEEE-exact operations (+, -, *, /, fmaf, sqrtf, rintf,
 * floorf and integer ops) -- no libm/ocml transcendental whose rounding differs
 * between glibc and the device library.  Compile with -ffp-contract=off on both
 * sides; all fusion is spelled fmaf().
 *
 * Plain C99; under hipcc the functions become __host__ __device__.
 */
#ifndef V2E_DETMATH_H
#define V2E_DETMATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define V2E_HD __host__ __device__ static inline
#else
#define V2E_HD static inline
#endif

/* ------------------------------------------------------------------ Philox */
#define V2E_PHILOX_M0 0xD2511F53u
#define V2E_PHILOX_M1 0xCD9E8D57u
#define V2E_PHILOX_W0 0x9E3779B9u
#define V2E_PHILOX_W1 0xBB67AE85u

/* The ten round keys of a Philox4x32-10 call (key bumped by the Weyl constants per round).  They depend on the key alone, so a
 * kernel that makes many calls under one key computes them once (k_ahead: they are scalar registers there). */
typedef struct { uint32_t k0[10], k1[10]; } v2e_philox_keys;
V2E_HD void v2e_philox_key_schedule(uint32_t k0, uint32_t k1, v2e_philox_keys *ks)
{
    for (int r = 0; r < 10; ++r) {
        ks->k0[r] = k0 + (uint32_t)r * V2E_PHILOX_W0;
        ks->k1[r] = k1 + (uint32_t)r * V2E_PHILOX_W1;
    }
}

/* Philox4x32-10 (Salmon et al., SC'11) under a precomputed key schedule.  ctr[4] in, out[4] out. */
V2E_HD void v2e_philox4x32_ks(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const v2e_philox_keys *ks, uint32_t out[4])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)V2E_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)V2E_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ ks->k0[r];
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ ks->k1[r];
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Philox4x32-10.  ctr[4] in, out[4] out. */
V2E_HD void v2e_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)V2E_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)V2E_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += V2E_PHILOX_W0; k1 += V2E_PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Stream ids (counter word 2). */
#define V2E_STREAM_FRAME 0u /* per (pixel, frame pair): [0],[1] two leak normals, [2],[3] two shot uniforms */
#define V2E_STREAM_THRES 1u /* per pixel: [0],[1] pos-threshold normal, [2],[3] neg-threshold normal */
#define V2E_STREAM_RATE  2u /* per pixel: [0],[1] noise-rate normal */
#define V2E_STREAM_PERM  3u /* per (frame, iteration): shuffle key */
#define V2E_STREAM_PNOISE 4u /* per (pixel, frame): [0],[1] photoreceptor-noise normal */

/* 24-bit uniform in [0,1): same granularity as torch's f32 uniform. */
V2E_HD float v2e_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
/* 24-bit uniform in (0,1]. */
V2E_HD float v2e_u01_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }

V2E_HD uint32_t v2e_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
V2E_HD float v2e_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

/* ------------------------------------------------------- deterministic math */

/* ln(x) for normal positive x.  ~1.5 ulp; deterministic. */
V2E_HD float v2e_det_logf(float x)
{
    uint32_t ux = v2e_f2u(x);
    /* normalise mantissa to [sqrt(1/2), sqrt(2)) */
    int32_t e = (int32_t)(ux >> 23) - 127;
    uint32_t mbits = (ux & 0x007FFFFFu) | 0x3F800000u;
    float m = v2e_u2f(mbits);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    /* 2*atanh(s) = 2s(1 + z/3 + z^2/5 + z^3/7 + z^4/9) , |s| <= 0.1716 */
    float p = 0.2222222222f;
    p = fmaf(p, z, 0.2857142857f);
    p = fmaf(p, z, 0.4f);
    p = fmaf(p, z, 0.6666666667f);
    p = fmaf(p, z, 2.0f);
    float lm = p * s;
    float fe = (float)e;
    /* ln2 split: hi has 12 trailing zero bits so fe*hi is exact for |e|<=127+ */
    float r = fmaf(fe, 1.42860677e-06f, lm);
    return fmaf(fe, 0.693145752f, r);
}

/* sin/cos of 2*pi*u for u in [0,1).  deterministic, ~1e-7 abs. */
V2E_HD void v2e_det_sincos2pi(float u, float *sn, float *cs)
{
    float a = u * 4.0f;                 /* exact */
    float kf = floorf(a + 0.5f);
    float r = a - kf;                   /* exact, in [-0.5, 0.5] */
    float t = r * 1.57079632679f;       /* quadrant angle in [-pi/4, pi/4] */
    float t2 = t * t;
    float sp = -1.9515295891e-4f;
    sp = fmaf(sp, t2, 8.3321608736e-3f);
    sp = fmaf(sp, t2, -1.6666654611e-1f);
    float s0 = fmaf(sp * t2, t, t);
    float cp = 2.443315711809948e-5f;
    cp = fmaf(cp, t2, -1.388731625493765e-3f);
    cp = fmaf(cp, t2, 4.166664568298827e-2f);
    float c0 = fmaf(cp, t2 * t2, fmaf(-0.5f, t2, 1.0f));
    /* quadrant k: (s, c) = (s0, c0), (c0, -s0), (-s0, -c0), (-c0, s0) -- as selects and sign flips (exact, -0 included): no branch */
    int k = ((int)kf) & 3;
    float s = (k & 1) ? c0 : s0;
    float c = (k & 1) ? s0 : c0;
    s = (k & 2) ? -s : s;
    c = ((k + 1) & 2) ? -c : c;
    *sn = s; *cs = c;
}

/* exp(x), |x| < 80.  deterministic, ~1 ulp. */
V2E_HD float v2e_det_expf(float x)
{
    float n = rintf(x * 1.44269504089f);
    float r = fmaf(-n, 0.693145752f, x);
    r = fmaf(-n, 1.42860677e-06f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p * r, r, r) + 1.0f;
    int32_t ni = (int32_t)n;
    /* scale by 2^n in two steps to stay in range */
    int32_t h = ni / 2;
    float s1 = v2e_u2f((uint32_t)(h + 127) << 23);
    float s2 = v2e_u2f((uint32_t)(ni - h + 127) << 23);
    return y * s1 * s2;
}

/* Standard normal from two Philox words (Box-Muller, cosine branch). */
V2E_HD float v2e_normal(uint32_t a, uint32_t b)
{
    float u1 = v2e_u01_open0(a);
    float u2 = v2e_u01(b);
    float rad = sqrtf(-2.0f * v2e_det_logf(u1));
    float s, c;
    v2e_det_sincos2pi(u2, &s, &c);
    return rad * c;
}

/* Two independent standard normals from two Philox words: both branches of one Box-Muller transform. */
V2E_HD void v2e_normal2(uint32_t a, uint32_t b, float *n_cos, float *n_sin)
{
    float u1 = v2e_u01_open0(a);
    float u2 = v2e_u01(b);
    float rad = sqrtf(-2.0f * v2e_det_logf(u1));
    float s, c;
    v2e_det_sincos2pi(u2, &s, &c);
    *n_cos = rad * c;
    *n_sin = rad * s;
}

/*
 * Per-(pixel, frame) draws: leak-jitter normal and shot-noise uniform.  ONE Philox4x32-10 call serves the two
 * consecutive frames of a pair q = (frame + 1) >> 1 (frames 2q-1 and 2q; frame 0 is the first frame, which draws
 * from the THRES / RATE streams only): words [0],[1] -> both Box-Muller branches (cosine: the odd frame, sine: the
 * even frame), [2] -> the odd frame's uniform, [3] -> the even frame's.  A kernel that advances both frames of a
 * pair in one launch (k_step2) pays for one call; v2e_draw_frame is the same stream read one frame at a time.
 */
V2E_HD uint32_t v2e_frame_pair(uint32_t frame) { return (frame + 1u) >> 1; }
V2E_HD uint32_t v2e_frame_half(uint32_t frame) { return (frame + 1u) & 1u; } /* 0: odd frame (first of its pair) */

V2E_HD void v2e_draw_pair(uint64_t seed, uint32_t clip, uint32_t pair, uint32_t pixel, int want_normal,
                          float *randn_odd, float *u_odd, float *randn_even, float *u_even)
{
    uint32_t o[4];
    v2e_philox4x32(pixel, pair, V2E_STREAM_FRAME, clip, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    *randn_odd = 0.0f;
    *randn_even = 0.0f;
    if (want_normal) v2e_normal2(o[0], o[1], randn_odd, randn_even);
    *u_odd = v2e_u01(o[2]);
    *u_even = v2e_u01(o[3]);
}

/* The same draws under a precomputed key schedule of the seed (v2e_philox_key_schedule((uint32_t)seed, (uint32_t)(seed >> 32), ..)). */
V2E_HD void v2e_draw_pair_ks(const v2e_philox_keys *ks, uint32_t clip, uint32_t pair, uint32_t pixel, int want_normal,
                             float *randn_odd, float *u_odd, float *randn_even, float *u_even)
{
    uint32_t o[4];
    v2e_philox4x32_ks(pixel, pair, V2E_STREAM_FRAME, clip, ks, o);
    *randn_odd = 0.0f;
    *randn_even = 0.0f;
    if (want_normal) v2e_normal2(o[0], o[1], randn_odd, randn_even);
    *u_odd = v2e_u01(o[2]);
    *u_even = v2e_u01(o[3]);
}

V2E_HD void v2e_draw_frame(uint64_t seed, uint32_t clip, uint32_t frame, uint32_t pixel,
                           float *leak_randn, float *shot_u)
{
    float r0, u0, r1, u1;
    v2e_draw_pair(seed, clip, v2e_frame_pair(frame), pixel, 1, &r0, &u0, &r1, &u1);
    const int even = (int)v2e_frame_half(frame);
    *leak_randn = even ? r1 : r0;
    *shot_u = even ? u1 : u0;
}

/* Per-(pixel,frame) photoreceptor-noise draw (emulator.py:698 in philox mode). */
V2E_HD float v2e_draw_pnoise(uint64_t seed, uint32_t clip, uint32_t frame, uint32_t pixel)
{
    uint32_t o[4];
    v2e_philox4x32(pixel, frame, V2E_STREAM_PNOISE, clip, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return v2e_normal(o[0], o[1]);
}

/* Per-pixel first-frame draws (emulator.py:459-471, 501-505 in philox mode). */
V2E_HD void v2e_draw_init(uint64_t seed, uint32_t clip, uint32_t pixel,
                          float *n_pos, float *n_neg, float *n_rate)
{
    uint32_t o[4];
    v2e_philox4x32(pixel, 0u, V2E_STREAM_THRES, clip, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    *n_pos = v2e_normal(o[0], o[1]);
    *n_neg = v2e_normal(o[2], o[3]);
    v2e_philox4x32(pixel, 0u, V2E_STREAM_RATE, clip, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    *n_rate = v2e_normal(o[0], o[1]);
}

/* SCIDVS per-pixel time-constant normal (emulator.py:480-483 in philox mode): the second pair of the noise-rate call. */
V2E_HD float v2e_draw_scidvs(uint64_t seed, uint32_t clip, uint32_t pixel)
{
    uint32_t o[4];
    v2e_philox4x32(pixel, 0u, V2E_STREAM_RATE, clip, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return v2e_normal(o[2], o[3]);
}

/* ---- torch.sinh on float32 CPU tensors, bit for bit (SCIDVS with float32 state: emulator.py:78 `torch.sinh(v / efold)`).
 * ATen's vectorised float32 sinh is Sleef's `sinhf_u10` (Sleef_sinhf8_u10 / Sleef_sinhf16_u10: the AVX2 and the AVX-512 build of
 * torch give the same bits, tests/golden/make_golden_scidvs.py checks 17 M arguments), i.e. the published algorithm of
 * sleef/src/libm/sleefsimdsp.c (xsinhf, expk2f) on double-float numbers with fused multiply-adds, restated here operation for
 * operation: every fusion is an fmaf, every other operation is rounded on its own (compile with -ffp-contract=off). */
typedef struct { float x, y; } v2e_f2;
V2E_HD float v2e_fmapn(float x, float y, float z) { return fmaf(x, y, -z); } /* x y - z */
V2E_HD float v2e_fmanp(float x, float y, float z) { return fmaf(-x, y, z); } /* -x y + z */
V2E_HD v2e_f2 v2e_dfmul_f2_f(v2e_f2 x, float y) { float s = x.x * y; v2e_f2 r; r.x = s; r.y = fmaf(x.y, y, v2e_fmapn(x.x, y, s)); return r; }
V2E_HD v2e_f2 v2e_dfmul_f2_f2(v2e_f2 x, v2e_f2 y)
{
    float s = x.x * y.x;
    v2e_f2 r; r.x = s; r.y = fmaf(x.x, y.y, fmaf(x.y, y.x, v2e_fmapn(x.x, y.x, s)));
    return r;
}
V2E_HD v2e_f2 v2e_dfsqu(v2e_f2 x) { float s = x.x * x.x; v2e_f2 r; r.x = s; r.y = fmaf(x.x + x.x, x.y, v2e_fmapn(x.x, x.x, s)); return r; }
V2E_HD v2e_f2 v2e_dfrec(v2e_f2 d) { float s = 1.0f / d.x; v2e_f2 r; r.x = s; r.y = s * v2e_fmanp(d.y, s, v2e_fmanp(d.x, s, 1.0f)); return r; }
V2E_HD v2e_f2 v2e_dfadd_f_f2(float x, v2e_f2 y) { float s = x + y.x; v2e_f2 r; r.x = s; r.y = ((x - s) + y.x) + y.y; return r; }
V2E_HD v2e_f2 v2e_dfadd2_f2_f(v2e_f2 x, float y)
{
    float s = x.x + y, v = s - x.x, t = (x.x - (s - v)) + (y - v);
    v2e_f2 r; r.x = s; r.y = t + x.y;
    return r;
}
V2E_HD v2e_f2 v2e_dfadd2_f2_f2(v2e_f2 x, v2e_f2 y)
{
    float s = x.x + y.x, v = s - x.x, t = (x.x - (s - v)) + (y.x - v);
    v2e_f2 r; r.x = s; r.y = t + (x.y + y.y);
    return r;
}
V2E_HD v2e_f2 v2e_dfsub_f2_f2(v2e_f2 x, v2e_f2 y)
{
    float s = x.x - y.x, t = x.x - s;
    t = t - y.x; t = t + x.y;
    v2e_f2 r; r.x = s; r.y = t - y.y;
    return r;
}
V2E_HD float v2e_pow2if(int q) { return v2e_u2f((uint32_t)(q + 0x7f) << 23); }
V2E_HD float v2e_ldexp2f(float d, int e) { return d * v2e_pow2if(e >> 1) * v2e_pow2if(e - (e >> 1)); }
V2E_HD v2e_f2 v2e_sleef_expk2f(v2e_f2 d)
{
    float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    int q = (int)rintf(u);
    v2e_f2 s = v2e_dfadd2_f2_f(d, (float)q * -0.693145751953125f);
    s = v2e_dfadd2_f2_f(s, (float)q * -1.428606765330187045e-06f);
    u = +0.1980960224e-3f;
    u = fmaf(u, s.x, +0.1394256484e-2f);
    u = fmaf(u, s.x, +0.8333456703e-2f);
    u = fmaf(u, s.x, +0.4166637361e-1f);
    v2e_f2 t = v2e_dfadd2_f2_f(v2e_dfmul_f2_f(s, u), +0.166666659414234244790680580464e+0f);
    t = v2e_dfadd2_f2_f(v2e_dfmul_f2_f2(s, t), 0.5f);
    t = v2e_dfadd2_f2_f2(s, v2e_dfmul_f2_f2(v2e_dfsqu(s), t));
    t = v2e_dfadd_f_f2(1.0f, t);
    t.x = v2e_ldexp2f(t.x, q);
    t.y = v2e_ldexp2f(t.y, q);
    if (d.x < -104.0f) { t.x = 0.0f; t.y = 0.0f; }
    return t;
}
V2E_HD float v2e_sleef_sinhf(float x)
{
    float y = fabsf(x);
    v2e_f2 in; in.x = y; in.y = 0.0f;
    v2e_f2 d = v2e_sleef_expk2f(in);
    d = v2e_dfsub_f2_f2(d, v2e_dfrec(d));
    y = (d.x + d.y) * 0.5f;
    if (fabsf(x) > 89.0f || y != y) y = v2e_u2f(0x7F800000u);
    y = v2e_u2f((v2e_f2u(y) & 0x7FFFFFFFu) | (v2e_f2u(x) & 0x80000000u)); /* mulsign */
    if (x != x) y = v2e_u2f(0x7FC00000u);
    return y;
}


/* ------------------------------------------------ keyed bijection (shuffle) */
/*
 * Philox-mode replacement for `idx = torch.randperm(n_i)` (emulator.py:868): a keyed
 * bijection on [0,n).  sigma(c) is the OUTPUT position of the event whose canonical
 * (ON-row-major-then-OFF-row-major) index is c, i.e. reference idx[sigma(c)] = c.
 *
 * Construction: mixed-radix Feistel (Black & Rogaway, "Ciphers with arbitrary finite
 * domains") on [0,a) x [0,2^k), k ~ half the bits of n, a = ceil(n / 2^k), 4 rounds, plus
 * cycle walking for the < 2^k ~ sqrt(n) values of [n, a*2^k).  The domain is within
 * 1/sqrt(n) of n, so a walk is rare (a wave waits for its slowest lane), and splitting by a
 * power of two makes init and apply shifts, masks, adds, a multiply-high and a conditional subtraction: no
 * division, no loops.  Integer arithmetic only, so host and device agree bit for bit.
 */
V2E_HD uint32_t v2e_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

V2E_HD uint32_t v2e_bits32(uint32_t x) /* number of bits needed to represent x (0 -> 0) */
{
    return x ? 32u - (uint32_t)__builtin_clz(x) : 0u;
}

typedef struct { uint32_t k[4]; uint32_t a, amask, sh, rmask, n; } v2e_perm_t;

/* the four round keys of (seed, clip, frame, iteration) */
V2E_HD void v2e_perm_keys(uint64_t seed, uint32_t clip, uint32_t frame, uint32_t iter, uint32_t k[4])
{
    uint32_t h = v2e_mix32((uint32_t)seed ^ 0x9E3779B9u);
    h = v2e_mix32(h + (uint32_t)(seed >> 32));
    h = v2e_mix32(h + frame * 0x85EBCA6Bu + iter * 0xC2B2AE35u + clip * 0x27D4EB2Fu + V2E_STREAM_PERM);
    k[0] = v2e_mix32(h + 0x165667B1u);
    k[1] = v2e_mix32(h + 0x2CACCF62u);
    k[2] = v2e_mix32(h + 0x43033713u);
    k[3] = v2e_mix32(h + 0x59599EC4u);
}

/* the domain split of n: low part 2^sh, high part a = ceil(n / 2^sh), amask = 2^ceil(log2 a) - 1 */
V2E_HD void v2e_perm_shape(uint32_t n, uint32_t *sh_o, uint32_t *a_o, uint32_t *amask_o)
{
    const uint32_t nb = v2e_bits32(n > 1u ? n - 1u : 1u); /* bits of the largest index */
    const uint32_t sh = nb >> 1;
    const uint32_t a = ((n > 0u ? n - 1u : 0u) >> sh) + 1u;
    *sh_o = sh;
    *a_o = a;
    *amask_o = a > 1u ? (1u << v2e_bits32(a - 1u)) - 1u : 0u; /* < 2a */
}

V2E_HD void v2e_perm_init(v2e_perm_t *p, uint64_t seed, uint32_t clip, uint32_t frame,
                          uint32_t iter, uint32_t n)
{
    v2e_perm_keys(seed, clip, frame, iter, p->k);
    v2e_perm_shape(n, &p->sh, &p->a, &p->amask);
    p->rmask = (1u << p->sh) - 1u;
    p->n = n;
}

V2E_HD uint32_t v2e_perm_apply(const v2e_perm_t *p, uint32_t c)
{
    uint32_t x = c;
    do {
        uint32_t l = x >> p->sh, r = x & p->rmask; /* l in [0,a), r in [0,2^sh) */
        for (int round = 0; round < 4; ++round) {
            if ((round & 1) == 0) { /* add a value uniform on [0,a): high word of hash * a (a mask-and-subtract
                                       reduction is measurably biased: chi-square of 6 000 shuffles of 24 elements) */
                l += (uint32_t)(((uint64_t)v2e_mix32(r + p->k[round]) * p->a) >> 32);
                if (l >= p->a) l -= p->a;
            } else {
                r = (r + v2e_mix32(l + p->k[round])) & p->rmask;
            }
        }
        x = (l << p->sh) | r;
    } while (x >= p->n);
    return x;
}

/* the inverse map: v2e_perm_invert(p, v2e_perm_apply(p, c)) == c for c < n.  The rounds backwards with the additions undone; the
 * forward map's cycle walk (re-apply while the image lies outside [0, n)) is undone by walking the same cycle backwards.  The event
 * writer pulls with it: output row j of an iteration holds the event of canonical index v2e_perm_invert(j) (emulator.py:861-870). */
V2E_HD uint32_t v2e_perm_invert(const v2e_perm_t *p, uint32_t y)
{
    uint32_t x = y;
    do {
        uint32_t l = x >> p->sh, r = x & p->rmask;
        for (int round = 3; round >= 0; --round) {
            if ((round & 1) == 0) {
                const uint32_t f = (uint32_t)(((uint64_t)v2e_mix32(r + p->k[round]) * p->a) >> 32); /* in [0, a) */
                l = l >= f ? l - f : l + (p->a - f);
            } else {
                r = (r - v2e_mix32(l + p->k[round])) & p->rmask;
            }
        }
        x = (l << p->sh) | r;
    } while (x >= p->n);
    return x;
}

/* position (0 .. 255) of the r-th set bit (r counted from 0, r < the number of set bits) in the 256-bit word m[0] | m[1] << 64 | ...:
 * which pixel of a 256-pixel emission group is the r-th of its (iteration, polarity) block in pixel order */
V2E_HD uint32_t v2e_nth_set_bit_256(uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, uint32_t r)
{
    const uint32_t c0 = (uint32_t)__builtin_popcountll(m0), c1 = c0 + (uint32_t)__builtin_popcountll(m1),
                   c2 = c1 + (uint32_t)__builtin_popcountll(m2);
    uint64_t w = m0;
    uint32_t pos = 0;
    if (r >= c0) { w = m1; pos = 64; }
    if (r >= c1) { w = m2; pos = 128; }
    if (r >= c2) { w = m3; pos = 192; }
    r -= r >= c2 ? c2 : (r >= c1 ? c1 : (r >= c0 ? c0 : 0u));
    uint32_t v = (uint32_t)w;
    const uint32_t cl = (uint32_t)__builtin_popcount(v);
    if (r >= cl) { r -= cl; v = (uint32_t)(w >> 32); pos += 32; }
    for (uint32_t s = 16; s >= 1; s >>= 1) {
        const uint32_t cs = (uint32_t)__builtin_popcount(v & ((1u << s) - 1u));
        if (r >= cs) { r -= cs; v >>= s; pos += s; }
    }
    return pos;
}

/* ------------------------------------------------------- timestamp formula */
/*
 * In-kernel stand-in for torch.linspace(start, end, n, dtype=float32)
 * (emulator.py:793-796) used when no host timestamp table is supplied.  It is the
 * scalar form of ATen's symmetric fill (RangeFactoriesKernel.cpp) with the
 * multiply-add fused, which is what the AVX2/AVX-512 builds of that kernel
 * compute for n below one unrolled vector step.  start/end/step are f32.
 */
V2E_HD float v2e_ts_formula(uint32_t i, uint32_t n, float start, float end, float step)
{
    if (n <= 1u) return start;
    if (i < n / 2u) return fmaf(step, (float)i, start);
    return fmaf(-step, (float)(n - 1u - i), end);
}

#endif /* V2E_DETMATH_H */
