"""PIL-exact 8-bit image resampling tables for the on-device SloMo <-> emulator hand-off.

The reference moves frames between its two hot kernels through PIL and PNG files:
  * input side   dataloader.py:136-147  Image.resize(dim, Image.LANCZOS) of each uint8 source frame
  * output side  slomo.py:437-444       ToPILImage (x*255 -> byte) + Image.resize(ori_dim, Image.BILINEAR)
To keep the emulator's input bit-identical while staying in HBM, the kernels in
v2e_amd/csrc/resample.hip apply Pillow's 8-bit-per-channel algorithm (libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc): separable
windowed filter, coefficients normalised in double then rounded to 22-bit fixed point, a uint8
(clip8) intermediate between the horizontal and the vertical pass.  The coefficient tables are
small and computed here on the host exactly as Pillow does; the per-pixel integer arithmetic runs
on the GPU.  tests/test_resample.py checks the whole thing against PIL itself.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "lanczos": (_lanczos, 3.0)}


def coeffs_8bpc(in_size, out_size, filt):
    """(bounds int32 [out,2] = (xmin, count), coefs int32 [out, ksize]) of one axis (Resample.c)."""
    fn, fsupport = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return bounds, np.trunc(ik).astype(np.int32)


def resample_u8_numpy(img, out_hw, filt):
    """Pure-numpy statement of the two passes (test helper; the product path is the HIP kernel)."""
    ih, iw = img.shape
    oh, ow = out_hw
    src = img.astype(np.int64)
    if ow != iw:
        b, k = coeffs_8bpc(iw, ow, filt)
        tmp = np.zeros((ih, ow), np.int64)
        for xx in range(ow):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n] * k[xx, :n].astype(np.int64)).sum(axis=1)
            tmp[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
        src = tmp
    if oh != ih:
        b, k = coeffs_8bpc(ih, oh, filt)
        out = np.zeros((oh, src.shape[1]), np.int64)
        for yy in range(oh):
            y0, n = b[yy]
            acc = (1 << (PRECISION_BITS - 1)) + (src[y0:y0 + n, :] * k[yy, :n].astype(np.int64)[:, None]).sum(axis=0)
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        src = out
    return src.astype(np.uint8)
