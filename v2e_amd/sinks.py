"""Device-side packing of events into the reference writers' integer formats."""
import ctypes as C

import torch

from . import _capi
from ._capi import check

# aedat2_output.py:41-77
_AEDAT2_LAYOUT = {(346, 260): (12, 22, 11, 1, 1), (240, 180): (12, 22, 11, 1, 1), (640, 480): (1, 11, 0, 1, 1)}


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def pack_aedat2(events_dev, width, height, noise_from=-1):
    """events_dev: device float32 [N,4] -> device uint8 [N*8], the bytes AEDat2Output.appendEvents would write."""
    if (width, height) not in _AEDAT2_LAYOUT:
        raise ValueError(f'AEDAT-2.0 output width={width} height={height} not supported')  # aedat2_output.py:78-80
    xs, ys, ps, fx, fy = _AEDAT2_LAYOUT[(width, height)]
    n = int(events_dev.shape[0])
    out = torch.empty((n * 8,), dtype=torch.uint8, device=events_dev.device)
    s = C.c_void_p(torch.cuda.current_stream(events_dev.device).cuda_stream)
    check(_capi.lib().v2e_events_pack_aedat2(_ptr(events_dev.contiguous()), _ptr(out), n, width, height, xs, ys, ps, fx, fy,
                                             int(noise_from), s), "v2e_events_pack_aedat2")
    return out


def pack_h5(events_dev):
    """events_dev: device float32 [N,4] -> device int64-free uint32 rows (as int32 tensor bit pattern) [N,4]."""
    n = int(events_dev.shape[0])
    out = torch.empty((n, 4), dtype=torch.int32, device=events_dev.device)  # torch has no uint32 arithmetic; same bits
    s = C.c_void_p(torch.cuda.current_stream(events_dev.device).cuda_stream)
    check(_capi.lib().v2e_events_pack_h5(_ptr(events_dev.contiguous()), _ptr(out), n, s), "v2e_events_pack_h5")
    return out


class EventFrameAccumulator:
    """Device version of EventRenderer.accumulate_event_frame (renderer.py:368-400): ON/OFF 2-D histogram of a
    slice of events added to a running, clipped frame.  The exposure-mode slicing (renderer.py:161-366) stays
    with the caller."""

    def __init__(self, height, width, full_scale_count, device="cuda", histrange=None):
        self.height, self.width, self.full_scale_count = int(height), int(width), float(full_scale_count)
        self.device = torch.device(device)
        self.histrange = histrange
        self.currentFrame = torch.zeros((self.height, self.width), dtype=torch.float64, device=self.device)
        self._diff = torch.zeros((self.height, self.width), dtype=torch.int32, device=self.device)

    def accumulate(self, events_dev, sensor_height=None, sensor_width=None):
        hr = self.histrange or [(0, sensor_height or self.height), (0, sensor_width or self.width)]  # renderer.py:232
        n = int(events_dev.shape[0])
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(_capi.lib().v2e_events_accumulate_frame(_ptr(events_dev.contiguous()) if n else None, n, _ptr(self.currentFrame),
                                                      _ptr(self._diff), self.height, self.width, float(hr[0][0]), float(hr[0][1]),
                                                      float(hr[1][0]), float(hr[1][1]), self.full_scale_count, s),
              "v2e_events_accumulate_frame")
        return self.currentFrame
