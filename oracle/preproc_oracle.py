"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of stage 1 of v2e.py (:687-738): cv2.resize(INTER_AREA) + cv2.cvtColor
(COLOR_BGR2GRAY) on uint8 frames.  Only tests/ import this.

PARITY UNPINNED.  The algorithm lives in OpenCV (reference dependency 'opencv-python', setup.py:42 / requirements.txt:39, no
version pinned; not vendored, not installable here: no network), and the reference's tests hold no vectors for it.  This file
restates OpenCV 4.x's published source as follows; it has NOT been compared with cv2 output:

  modules/imgproc/src/resize.cpp, cv::resize: inv_scale_x = (double)dsize.width / ssize.width, scale_x = 1. / inv_scale_x.
  INTER_AREA with scale_x >= 1 and scale_y >= 1:
    * iscale = saturate_cast<int>(scale); is_area_fast = |scale - iscale| < DBL_EPSILON on both axes.
      resizeAreaFast_<uchar, int, ResizeAreaFastVec<...>>: sum of the iscale_x * iscale_y source samples as int;
      2 x 2 (the vector op's own formula, also its scalar tail): D = (s00 + s01 + s10 + s11 + 2) >> 2;
      otherwise D = saturate_cast<uchar>(sum * scale) with float scale = 1.f / area (cvRound: half to even).
    * else computeResizeAreaTab per axis (restated in v2e_amd.preproc.area_tab) and resizeArea_<uchar, float>: for the table's
      source rows in order: buf[dx] = 0; buf[di] += S[si] * alpha for the x table in order (float32); rows of one
      destination row: sum = beta * buf for the first, sum += beta * buf after; D = saturate_cast<uchar>(sum).
      Each product and each sum is rounded on its own (no fused multiply-add: resizeArea_ has no FMA dispatch variant).
  modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar> (BGR order: blueIdx 0), the DEFAULT here since round 5: gray_shift 15,
      BY15 3735, GY15 19235, RY15 9798 (color.simd_helpers.hpp), D = (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15.
      Rounds 3-4 restated the 14-bit set (B2Y 1868, G2Y 9617, R2Y 4899, (1 << 13) >> 14): that is OpenCV 3.x's table form and 4.x's
      YUV path; the round-4 advisor and the builder, both from memory and independently, place the 15-bit set in 4.x's RGB2Gray<uchar>,
      and an unpinned `pip install opencv-python` has given v2e a 4.x for years -- so 15 is the default and 14 stays selectable
      (`Stage1(gray_shift=14)`).  The two differ by one grey level on ~1 % of random pixels.  Neither of us can open the file here:
      scripts/check_stage1_against_cv2.py prints which of the two equals cv2.cvtColor on a host that has OpenCV -- until someone
      runs it this stage stays "parity unpinned".
  What CAN be checked without OpenCV is checked in tests/test_preproc.py: the integer-factor path equals the exact rational box mean
  (ties as OpenCV rounds them) and stays within one grey level of Pillow's independent Image.reduce.
"""
import numpy as np


def resize_area(img, out_wh, area_tab, is_area_fast):
    """img uint8 [H, W] or [H, W, 3] -> uint8 [oh, ow(, 3)]."""
    ow, oh = int(out_wh[0]), int(out_wh[1])
    a = img if img.ndim == 3 else img[:, :, None]
    sh, sw, cn = a.shape
    assert oh <= sh and ow <= sw, "shrinking only"
    if is_area_fast((sw, sh), (ow, oh)):
        isx, isy = sw // ow, sh // oh
        box = a[:oh * isy, :ow * isx].astype(np.int64).reshape(oh, isy, ow, isx, cn).sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            out = ((box + 2) >> 2).astype(np.uint8)
        else:
            f = box.astype(np.float32) * (np.float32(1.0) / np.float32(isx * isy))
            out = np.clip(np.rint(f), 0, 255).astype(np.uint8)  # np.rint: half to even, as cvRound
    else:
        xo, xs, xa = area_tab(sw, ow)
        yo, ys, ya = area_tab(sh, oh)
        out = np.empty((oh, ow, cn), np.uint8)
        S = a.astype(np.float32)
        nx = np.diff(xo)
        for dy in range(oh):
            tot = None
            for j in range(yo[dy], yo[dy + 1]):
                row = S[ys[j]]                                   # [sw, cn]
                buf = np.zeros((ow, cn), np.float32)
                for kk in range(int(nx.max())):                  # the kk-th table entry of every destination column that has one
                    dxs = np.nonzero(nx > kk)[0]
                    k = xo[dxs] + kk
                    buf[dxs] = buf[dxs] + row[xs[k]] * xa[k][:, None]
                t = ya[j] * buf
                tot = t if tot is None else tot + t
            out[dy] = np.clip(np.rint(tot), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def bgr2gray(img, gray_shift=15):
    b, g, r = (img[..., c].astype(np.int64) for c in range(3))
    if gray_shift == 15:
        return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)
    assert gray_shift == 14
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def stage1(frame, out_wh, area_tab, is_area_fast, crop=None, gray_shift=15):
    """One frame through v2e.py:702-731: crop, resize if the uncropped size differs from the output size, grey."""
    ih, iw = frame.shape[:2]
    x = frame
    if crop is not None:
        l, r, t, b = (max(int(v), 0) for v in crop)
        x = x[t:(ih - b if b > 0 else None), l:(iw - r if r > 0 else None)]
    if ih != out_wh[1] or iw != out_wh[0]:
        x = resize_area(np.ascontiguousarray(x), out_wh, area_tab, is_area_fast)
    if x.ndim == 3:
        x = bgr2gray(x, gray_shift)
    return x
