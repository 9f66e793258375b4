"""The exact three-way bf16 split behind the default SuperSloMo convolution math (v2e_amd/csrc/slomo_s3.h).

CPU: the split restated in numpy -- x == p0 + p1 + p2 exactly, every piece a bf16, the dropped piece products bounded by
2^-24 |w x|, and the six kept products summed in double equal the f32 product to that bound.
GPU: v2e_pack_conv_weight_s3 writes exactly those pieces in the layout the kernel reads.
"""
import ctypes as C

import numpy as np
import pytest
import torch


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    p0 = bf16_rne(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_rne(r1)
    r2 = (r1 - p1).astype(np.float32)
    p2 = bf16_rne(r2)
    return p0, p1, p2


def _samples(seed, n):
    rng = np.random.Generator(np.random.PCG64(seed))
    mant = rng.integers(0, 1 << 23, size=n, dtype=np.uint32)
    exp = rng.integers(127 - 20, 127 + 8, size=n, dtype=np.uint32)  # 1e-6 .. 256
    sign = rng.integers(0, 2, size=n, dtype=np.uint32)
    x = ((sign << 31) | (exp << 23) | mant).view(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 + 2.0 ** -23, 255.0 / 256, 0.428, 3.0e-5], np.float32)
    return np.concatenate((x, edge))


def test_split_is_exact_and_pieces_are_bf16():
    x = _samples(1, 200_000)
    p0, p1, p2 = split3(x)
    for p in (p0, p1, p2):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0), "piece is not a bf16"
    # exact reconstruction (evaluated in double: the three pieces do not overlap)
    assert np.array_equal(p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64), x.astype(np.float64))
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(p1) <= ax * 2.0 ** -8) and np.all(np.abs(p2) <= ax * 2.0 ** -16)


def test_six_products_reproduce_the_f32_product_to_2_pow_minus_24():
    w = _samples(2, 100_000)
    x = _samples(3, 100_000)
    pw, px = [p.astype(np.float64) for p in split3(w)], [p.astype(np.float64) for p in split3(x)]
    kept = sum(pw[i] * px[j] for i in range(3) for j in range(3) if i + j <= 2)
    exact = w.astype(np.float64) * x.astype(np.float64)
    err = np.abs(kept - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -23)          # worst case
    nz = exact != 0
    assert np.mean(err[nz] / np.abs(exact[nz])) < 2.0 ** -26  # typical: far below one f32 rounding (2^-24)


def test_split_dot_products_are_as_accurate_as_f32_ones():
    """Whole dot products of conv size (K = 4608 = 512 channels x 9 taps): the six piece products accumulated in float32
    against the plain float32 dot product, both measured against float64 -- the split adds no error beyond what float32
    accumulation has anyway."""
    rng = np.random.Generator(np.random.PCG64(11))
    K, n = 4608, 400
    w = (rng.standard_normal((n, K)) * (1.5 / np.sqrt(K))).astype(np.float32)
    x = (rng.standard_normal((n, K)) * 2.0).astype(np.float32)
    exact = np.sum(w.astype(np.float64) * x.astype(np.float64), axis=1)
    pw, px = split3(w), split3(x)
    acc_split = np.zeros(n, np.float32)
    acc_f32 = np.zeros(n, np.float32)
    order = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]  # the kernel's: small products first
    for k0 in range(0, K, 16):  # one MFMA slab = 16 k values; products are exact in f32, the adds round
        for i, j in order:
            acc_split = (acc_split + np.sum((pw[i][:, k0:k0 + 16].astype(np.float64) * px[j][:, k0:k0 + 16].astype(np.float64)), axis=1)
                         .astype(np.float32)).astype(np.float32)
        acc_f32 = (acc_f32 + np.sum(w[:, k0:k0 + 16].astype(np.float64) * x[:, k0:k0 + 16].astype(np.float64), axis=1)
                   .astype(np.float32)).astype(np.float32)
    scale = np.maximum(1.0, np.abs(exact))
    e_split = np.max(np.abs(acc_split - exact) / scale)
    e_f32 = np.max(np.abs(acc_f32 - exact) / scale)
    assert e_split < 1e-5 and e_split <= 2.0 * e_f32, (e_split, e_f32)  # both ~3e-6 here: float32 accumulation of 4608 terms


@pytest.mark.gpu
def test_pack_kernel_writes_the_numpy_pieces():
    from v2e_amd import _capi
    from v2e_amd._capi import check
    lib = _capi.lib()
    rng = np.random.Generator(np.random.PCG64(5))
    cout, cin, k = 64, 48, 3
    w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
    dev = torch.device("cuda")
    tw = torch.from_numpy(w).to(dev)
    w3 = torch.empty(cin * k * k * cout * 6, dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.v2e_pack_conv_weight_s3(C.c_void_p(tw.data_ptr()), C.c_void_p(w3.data_ptr()), cout, cin, k, s), "pack_s3")
    torch.cuda.synchronize()
    got = w3.cpu().numpy().view(np.uint16).reshape(cin // 16, k * k, 3, 2, cout, 8)  # [chunk][tap][piece][ci/8][co][8]
    pieces = split3(w)
    for p in range(3):
        hi = (pieces[p].view(np.uint32) >> 16).astype(np.uint16)          # [co][ci][ky][kx]
        want = hi.reshape(cout, cin // 16, 2, 8, k * k).transpose(1, 4, 2, 0, 3)  # [chunk][tap][ci/8][co][8]
        assert np.array_equal(got[:, :, p], want), "piece %d" % p


# ---- the two-piece float16 split of conv_math "fp16x2" (slomo_s3.h: split2_pair, k_pack_weight_s3<2>)
def split2(x):
    x = np.asarray(x, np.float32)
    h0 = x.astype(np.float16).astype(np.float32)          # round to nearest even, like v_cvt_f16_f32
    h1 = (x - h0).astype(np.float32).astype(np.float16).astype(np.float32)
    return h0, h1


def test_two_float16_pieces_leave_2_pow_minus_22_in_the_normal_range():
    rng = np.random.Generator(np.random.PCG64(4))
    mant = rng.integers(0, 1 << 23, size=200_000, dtype=np.uint32)
    exp = rng.integers(127 - 3, 127 + 15, size=200_000, dtype=np.uint32)  # 0.125 .. 32768: h1 is a normal float16 too
    x = ((rng.integers(0, 2, size=200_000, dtype=np.uint32) << 31) | (exp << 23) | mant).view(np.float32)
    h0, h1 = split2(x)
    r = x.astype(np.float64) - h0.astype(np.float64) - h1.astype(np.float64)
    assert np.all(np.abs(r) <= np.abs(x.astype(np.float64)) * 2.0 ** -22)
    # the three kept products against the f32 product: the dropped h1 g1 and the two residual terms
    w = x[::-1].copy()
    g0, g1 = split2(w)
    kept = h0.astype(np.float64) * g0 + h0.astype(np.float64) * g1 + h1.astype(np.float64) * g0
    exact = x.astype(np.float64) * w.astype(np.float64)
    assert np.all(np.abs(kept - exact) <= np.abs(exact) * 2.0 ** -20.4)
    assert np.sqrt(np.mean(((kept - exact) / exact) ** 2)) < 2.0 ** -22.5


def test_two_float16_pieces_below_the_normal_range_are_off_by_an_absolute_6e_8():
    """Where h1 (or h0) is a float16 subnormal the split loses bits: what is left is an ABSOLUTE error of half a subnormal
    step, 2^-25 -- which is why the weights are packed times a power of two (their pieces then stay normal) and why the
    activations' share is bounded by 3e-8 |w| per term."""
    rng = np.random.Generator(np.random.PCG64(5))
    x = (rng.standard_normal(200_000) * np.exp(rng.uniform(np.log(1e-7), np.log(0.125), 200_000))).astype(np.float32)
    h0, h1 = split2(x)
    r = x.astype(np.float64) - h0.astype(np.float64) - h1.astype(np.float64)
    assert np.all(np.abs(r) <= 2.0 ** -25 * (1 + 1e-9) + np.abs(x.astype(np.float64)) * 2.0 ** -22)


def test_weight_scale_exponent_keeps_pieces_normal_and_finite():
    """The exponent SloMoEngine passes to v2e_pack_conv_weight_h2: the layer's largest weight lands in [2^12, 2^13)."""
    import math
    for wmax in (1e-4, 0.0123, 0.5, 1.0, 3.7, 100.0):
        sl2 = max(0, min(40, 12 - math.frexp(wmax)[1] + 1))
        assert 2.0 ** 12 <= wmax * 2.0 ** sl2 < 2.0 ** 13 or sl2 == 0
        assert np.isfinite(np.float16(wmax * 2.0 ** sl2))
