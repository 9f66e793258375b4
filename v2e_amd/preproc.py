"""Stage 1 of v2e.py (:687-738) on the device: crop, `cv2.resize(..., interpolation=cv2.INTER_AREA)` to the DVS size,
`cv2.cvtColor(..., cv2.COLOR_BGR2GRAY)` -- the frames then go to SuperSloMo / the emulator without leaving HBM.

PARITY UNPINNED: OpenCV is an un-vendored, un-pinned dependency of the reference
(setup.py: 'opencv-python') that is absent from this image, and the reference has no vectors for this stage.  What is restated
is OpenCV 4.x's published 8-bit algorithm (modules/imgproc/src/resize.cpp, color_yuv.simd.hpp) as read in
oracle/preproc_oracle.py; the kernels (v2e_amd/csrc/preproc.hip) equal that restatement bit for bit.  Shrinking only (the
INTER_AREA branch proper); a target larger than the source in either direction raises NotImplementedError.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi
from ._capi import check


def area_tab(ssize, dsize):
    """computeResizeAreaTab (resize.cpp) for one axis, grouped per destination index: (ofs int32 [dsize + 1], si int32 [K],
    alpha float32 [K]); entries ofs[d] .. ofs[d + 1] - 1 feed destination d, in the order OpenCV walks them."""
    scale = 1.0 / (float(dsize) / float(ssize))  # scale_x = 1. / inv_scale_x, inv_scale_x = (double)dsize.width / ssize.width
    ofs, si, al = [0], [], []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(math.ceil(fsx1)), int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            si.append(sx1 - 1)
            al.append(np.float32((sx1 - fsx1) / cell))
        for sx in range(sx1, sx2):
            si.append(sx)
            al.append(np.float32(1.0 / cell))
        if fsx2 - sx2 > 1e-3:
            si.append(sx2)
            al.append(np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell))
        ofs.append(len(si))
    return np.asarray(ofs, np.int32), np.asarray(si, np.int32), np.asarray(al, np.float32)


def is_area_fast(ssize_wh, dsize_wh):
    """resize.cpp: both scale factors are integers (|scale - int(scale)| < DBL_EPSILON)."""
    out = []
    for s, d in zip(ssize_wh, dsize_wh):
        scale = 1.0 / (float(d) / float(s))
        out.append(abs(scale - int(scale)) < np.finfo(np.float64).eps)
    return all(out)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Stage1:
    """v2e.py:687-738 for a batch of source frames already in HBM: uint8 [N, H, W] (grey) or [N, H, W, 3] (BGR, as cv2 reads
    a video) -> uint8 [N, out_h, out_w] grey.  crop = (left, right, top, bottom) pixels as --crop gives them."""

    def __init__(self, in_hw, out_wh, channels=1, crop=None, device="cuda", gray_shift=15):
        self.device = torch.device(device)
        self.lib = _capi.lib()
        self.cn = int(channels)
        # BGR2GRAY fixed-point form: 15 = OpenCV 4.x (what `pip install opencv-python` gives v2e today), 14 = OpenCV 3.x (csrc/preproc.hip)
        if gray_shift not in (14, 15):
            raise ValueError("gray_shift is 15 (OpenCV 4.x) or 14 (OpenCV 3.x)")
        self.gray_shift = int(gray_shift)
        if self.cn not in (1, 3):
            raise ValueError("frames have 1 (grey) or 3 (BGR) channels")
        self.crop = crop
        ih, iw = int(in_hw[0]), int(in_hw[1])
        self.in_hw = (ih, iw)
        if crop is not None:  # v2e.py:650-660, 702-712
            if len(crop) != 4:
                raise ValueError("--crop must have 4 elements (left, right, top, bottom)")
            l, r, t, b = (max(int(v), 0) for v in crop)
            if l + r >= iw:
                raise ValueError(f'left {l}+ right crop {r} is larger than image width {iw}')
            if t + b >= ih:
                raise ValueError(f'top {t}+ bottom crop {b} is larger than image height {ih}')
            self._sl = (slice(t, ih - b if b > 0 else None), slice(l, iw - r if r > 0 else None))
            ih, iw = ih - t - b, iw - l - r
        self.src_hw = (ih, iw)
        self.ow, self.oh = int(out_wh[0]), int(out_wh[1])
        # v2e.py:714-722 compares the UNCROPPED input size with the output size to decide whether to resize
        self.resize = (self.in_hw[0] != self.oh or self.in_hw[1] != self.ow)
        self.tabs = None
        if self.resize:
            if self.oh > ih or self.ow > iw:
                raise NotImplementedError("INTER_AREA is restated for shrinking only (source %dx%d -> %dx%d): OpenCV switches to its "
                                          "bilinear code there" % (iw, ih, self.ow, self.oh))
            if not is_area_fast((iw, ih), (self.ow, self.oh)):
                xo, xs, xa = area_tab(iw, self.ow)
                yo, ys, ya = area_tab(ih, self.oh)
                self.tabs = [torch.from_numpy(a).to(self.device) for a in (xo, xs, xa, yo, ys, ya)]

    def __call__(self, frames):
        assert frames.dtype == torch.uint8 and frames.is_cuda
        n = int(frames.shape[0])
        x = frames
        if self.crop is not None:
            x = x[:, self._sl[0], self._sl[1]]
        x = x.contiguous()
        ih, iw = self.src_hw
        assert tuple(x.shape[1:3]) == (ih, iw) and (x.ndim == 3) == (self.cn == 1)
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.resize:
            shape = (n, self.oh, self.ow) + ((3,) if self.cn == 3 else ())
            y = torch.empty(shape, dtype=torch.uint8, device=self.device)
            t = self.tabs or [None] * 6
            check(self.lib.v2e_resize_area_u8(_ptr(x), _ptr(y), n, ih, iw, self.oh, self.ow, self.cn, *[_ptr(a) for a in t], s),
                  "v2e_resize_area_u8")
            x = y
        if self.cn == 3:
            g = torch.empty(tuple(x.shape[:3]), dtype=torch.uint8, device=self.device)
            check(self.lib.v2e_bgr2gray_u8(_ptr(x), _ptr(g), n * int(x.shape[1]) * int(x.shape[2]), self.gray_shift, s), "v2e_bgr2gray_u8")
            x = g
        return x
