#!/usr/bin/env python
"""Golden frames of the reference's EventRenderer (v2ecore/renderer.py:161-400, imported from /root/reference; numba's
@jit is an identity stub, so hist2d_numba_seq and the nested helpers run as plain Python): the seeded event list of
sinks.npz fed in three packets through render_events_to_frames(return_frames=True) in every exposure mode, at sensor
resolution and down-scaled.

  renderer.npz   per case `<mode>_<h>x<w>`: frames of packet k as `..._p<k>` ([n,h,w] float64 in 0..1, or absent when the
                 call returned None) and the accumulator left behind (`..._cur`, or absent when None)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

CASES = [("duration", 0.5, None), ("duration", 2.0, None), ("count", 700, None), ("source", None, None), ("area_count", 40, 32)]
SIZES = [(260, 346), (130, 173)]
PACKETS = [(0, 1800), (1800, 1801), (1801, 5000)]


def main():
    rh.install_stubs()
    from v2ecore.renderer import EventRenderer, ExposureMode
    modes = {"duration": ExposureMode.DURATION, "count": ExposureMode.COUNT, "source": ExposureMode.SOURCE,
             "area_count": ExposureMode.AREA_COUNT}
    ev = np.load(os.path.join(HERE, "sinks.npz"))["ev_346x260"]
    out = {}
    for mode, val, area in CASES:
        for (h, w) in SIZES:
            if mode == "area_count" and (h, w) != SIZES[0]:
                continue  # the reference indexes its area counters with sensor coordinates: out of bounds when down-scaled
            r = EventRenderer(full_scale_count=3, exposure_mode=modes[mode], exposure_value=val if val is not None else 1 / 300.0,
                              area_dimension=area)
            key = "%s_%s_%dx%d" % (mode, str(val).replace(".", "p"), h, w)
            nfr = 0
            for k, (a, b) in enumerate(PACKETS):
                fr = r.render_events_to_frames(ev[a:b].copy(), height=h, width=w, return_frames=True)
                if fr is not None:
                    out[key + "_p%d" % k] = fr
                    nfr += len(fr)
            if r.currentFrame is not None:
                out[key + "_cur"] = r.currentFrame
            print("%-28s frames %d" % (key, nfr))
    np.savez_compressed(os.path.join(HERE, "renderer.npz"), **out)
    print("renderer.npz", os.path.getsize(os.path.join(HERE, "renderer.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
