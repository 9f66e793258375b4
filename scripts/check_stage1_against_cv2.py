#!/usr/bin/env python
"""For a host that HAS OpenCV (this image does not): compare stage 1 of the drop-in (crop, INTER_AREA resize, BGR2GRAY;
v2e.py:687-738) with cv2 itself and say which restatement matches.  The stage is built from OpenCV's published algorithm and is
PARITY UNPINNED until somebody runs this.

  python scripts/check_stage1_against_cv2.py            # CPU restatement (oracle/preproc_oracle.py) against cv2
  python scripts/check_stage1_against_cv2.py --hip      # the HIP kernels too (needs a GPU)

Prints, per case, the fraction of pixels that differ and the largest difference, and for BGR2GRAY both candidate fixed-point
forms: the 15-bit one that is the default since round 5 ((3735 B + 19235 G + 9798 R + 16384) >> 15: OpenCV 4.x's `RGB2Gray<uchar>`
as the oracle's header restates it) and the 14-bit one of OpenCV 3.x ((1868 B + 9617 G + 4899 R + 8192) >> 14; `Stage1(gray_shift=14)`).
Exactly one of the two BGR2GRAY lines is expected to say 0 differing pixels: that is the gray_shift to use with this cv2."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import cv2
except ImportError:
    sys.exit("cv2 is not importable here: nothing to compare with (this is the situation DESIGN.md section 7 describes)")
from oracle import preproc_oracle as po  # noqa: E402
from v2e_amd.preproc import area_tab, is_area_fast  # noqa: E402

rng = np.random.default_rng(1)
bad = 0


def report(tag, a, b, count=True):
    global bad
    d = np.abs(a.astype(int) - b.astype(int))
    if count:
        bad += int(d.max() > 0)
    print("%-58s differing pixels %.6f  max |d| %d" % (tag, float((d > 0).mean()), int(d.max())))
    return int(d.max()) == 0


img = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
g = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
ok15 = report("BGR2GRAY gray_shift = 15 (the default; OpenCV 4.x)", po.bgr2gray(img, 15), g, count=False)
ok14 = report("BGR2GRAY gray_shift = 14 (OpenCV 3.x)", po.bgr2gray(img, 14), g, count=False)
shift = 15 if ok15 else (14 if ok14 else None)
bad += shift is None
print("  -> this cv2 computes the %s form%s" % (("%d-bit" % shift) if shift else "NEITHER", "" if shift == 15 else ": construct Stage1(gray_shift=14)" if shift == 14 else ""))
for (sw, sh), (ow, oh) in (((1280, 720), (346, 260)), ((1280, 720), (640, 360)), ((1038, 780), (346, 260)), ((1280, 720), (320, 240)),
                           ((700, 530), (346, 260)), ((347, 261), (346, 260))):
    for cn in (1, 3):
        x = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, 3), dtype=np.uint8)
        report("INTER_AREA %dx%d -> %dx%d, %d channel(s)" % (sw, sh, ow, oh, cn), po.resize_area(x, (ow, oh), area_tab, is_area_fast),
               cv2.resize(x, (ow, oh), interpolation=cv2.INTER_AREA))
if "--hip" in sys.argv:
    import torch
    from v2e_amd import Stage1
    x = rng.integers(0, 256, (4, 720, 1280, 3), dtype=np.uint8)
    st = Stage1((720, 1280), (346, 260), channels=3, device="cuda", gray_shift=shift or 15)
    out = st(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = np.stack([cv2.cvtColor(cv2.resize(f, (346, 260), interpolation=cv2.INTER_AREA), cv2.COLOR_BGR2GRAY) for f in x])
    report("HIP Stage1 1280x720 BGR -> 346x260 grey", out, ref)
print("cv2 %s: %s" % (cv2.__version__, "every case identical" if bad == 0 else "%d case(s) differ -- please report which" % bad))
