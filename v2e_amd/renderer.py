"""Drop-in `EventRenderer`: DVS frames from events, rendered on the GPU a packet at a time.

Mirrors the surface of v2ecore/renderer.py: `ExposureMode`, the constructor signature (:37-99),
`render_events_to_frames(event_arr, height, width, return_frames) -> frames or None` (:161-366) with the frame the packet
ends in left behind in `currentFrame`, and `accumulate_event_frame` (:368-400).

How a packet is rendered here (v2e_amd/csrc/render.hip): the reference walks the packet window by window; this class
never walks it.  The host only works out the packet's window boundaries -- DURATION: the frame start times, the same
scalar additions the reference makes, until they pass the packet's second-to-last time stamp (the only two values read
back from the device); COUNT: plain index arithmetic; AREA_COUNT: the segment ends found by one single-wave scan kernel
(per-cell counters in LDS, the trigger event re-counted in the next window as renderer.py:253-266 does) -- and ONE
launch then drops every event into the frame(s) its window rule assigns it to (binary search over the boundaries; an
event exactly on a DURATION boundary belongs to both neighbours, as the reference's searchsorted left / right pair gives
it), one more clips and normalises all frames of the packet.  The reference's edge behaviour is part of the contract and
kept: the running frame is reset at every packet (:273), the last event of a packet is never accumulated (:303-306), the
frame a packet ends in is not emitted (except SOURCE).  Pinned by frames of the reference's own class in every mode
(tests/golden/make_golden_renderer.py).  AVI output and the preview window (cv2) are outside the hot path: `dvs_vid` /
`preview` raise NotImplementedError.
"""
import ctypes as C
import logging
from enum import Enum

import numpy as np
import torch

from . import _capi
from ._capi import check
from .sinks import EventFrameAccumulator

logger = logging.getLogger(__name__)


class ExposureMode(Enum):  # renderer.py:19-23
    DURATION = 1
    COUNT = 2
    AREA_COUNT = 3
    SOURCE = 4


class EventRenderer(object):
    """Class for DVS rendering from events (MI355X implementation of v2ecore.renderer.EventRenderer)."""

    def __init__(self, full_scale_count=3, output_path=None, dvs_vid=None, preview=False,
                 exposure_mode=ExposureMode.DURATION, exposure_value=1 / 300.0, area_dimension=None,
                 frame_times_suffix='-frame_times.txt', avi_frame_rate=30, device="cuda"):
        if dvs_vid is not None or preview:
            raise NotImplementedError("v2e_amd.EventRenderer: AVI output / preview (cv2) are outside the hot path; "
                                      "use return_frames=True")
        self.exposure_mode = exposure_mode
        self.exposure_value = exposure_value
        self.output_path = output_path
        self.width = None
        self.height = None
        self.full_scale_count = full_scale_count
        self.accum_mode = 'duration'
        self.dvs_frame_times_suffix = frame_times_suffix
        self.frame_rate_hz = None
        self.event_count = None
        self.frameIntevalS = None
        self.avi_frame_rate = avi_frame_rate
        self.area_counts = None
        self.area_count = None
        self.area_dimension = area_dimension
        if self.exposure_mode == ExposureMode.DURATION:
            self.frame_rate_hz = 1 / self.exposure_value
            self.frameIntevalS = 1 / self.frame_rate_hz
        elif self.exposure_mode == ExposureMode.COUNT:
            self.event_count = int(self.exposure_value)
        elif self.exposure_mode == ExposureMode.AREA_COUNT:
            self.area_count = int(self.exposure_value)
        elif self.exposure_mode == ExposureMode.SOURCE:
            pass
        else:
            raise ValueError(f'exposure mode {self.exposure_mode} is unknown; must be duration, count, or area-count')
        self.video_output_file_name = None
        self.video_output_file = None
        self.frame_times_output_file = None
        self.emulator = None
        self.preview = False
        self.numFramesWritten = 0
        self.currentFrameStartTime = None
        self.currentFrame = None  # device float64 [h,w] while a frame is being filled, else None
        self.printed_empty_packet_warning = False
        if not torch.cuda.is_available():
            raise _capi.V2EAmdError("v2e_amd.EventRenderer needs a ROCm GPU; there is no CPU fallback")
        self.device = torch.device(device)
        self._acc = None

    def cleanup(self):
        pass

    # renderer.py:368-400
    def accumulate_event_frame(self, events_dev, histrange):
        """ON/OFF histograms of `events_dev` (device float32 [n,4]) added to the running frame, clipped to +-full scale."""
        if self._acc is None or (self._acc.height, self._acc.width) != (self.height, self.width):
            self._acc = EventFrameAccumulator(self.height, self.width, self.full_scale_count, device=self.device)
        acc = self._acc
        acc.histrange = [(float(histrange[0][0]), float(histrange[0][1])), (float(histrange[1][0]), float(histrange[1][1]))]
        if self.currentFrame is None:
            acc.currentFrame.zero_()
        acc.accumulate(events_dev)
        self.currentFrame = acc.currentFrame

    def _duration_bounds(self, ts_first, ts_second_last, n):
        """Start times T_0 .. T_m of the frames this packet touches (frame k = [T_k, T_k+1], k < m complete; T_m opens the
        frame the packet ends in), advancing `currentFrameStartTime` as the reference does (renderer.py:209-213, 315-318):
        a frame is complete while events beyond its end remain besides the packet's last one, i.e. while the
        second-to-last time stamp lies past the next frame's start."""
        if self.currentFrameStartTime is None:
            self.currentFrameStartTime = ts_first
        bounds = [self.currentFrameStartTime]
        if n >= 2:
            while ts_second_last > self.currentFrameStartTime + self.frameIntevalS:
                nxt = self.currentFrameStartTime + self.frameIntevalS
                if not nxt > self.currentFrameStartTime:
                    raise ValueError("exposure duration %g s does not advance the frame time %r (float32 time stamps)"
                                     % (self.frameIntevalS, self.currentFrameStartTime))
                self.currentFrameStartTime += self.frameIntevalS
                bounds.append(self.currentFrameStartTime)
        return bounds

    def render_events_to_frames(self, event_arr, height: int, width: int, return_frames=False):
        """Render the frames this packet completes (renderer.py:161-366).  `event_arr`: [n,4] events (ts, x, y, pol) as a
        host array or a device tensor.  Returns the completed frames as a float64 [m,h,w] numpy array in 0..1 (only if
        return_frames), or None; the frame the packet ends in stays in `currentFrame` (device float64 [h,w])."""
        self.width = width
        self.height = height
        if event_arr is None or event_arr.shape[0] == 0:
            if not self.printed_empty_packet_warning:
                logger.info('event_arr is None or there are no events, doing nothing, supressing further warnings')
                self.printed_empty_packet_warning = True
            return None
        dev = self.device
        if torch.is_tensor(event_arr):
            ev = event_arr.to(dev, torch.float32).contiguous()
            n = int(ev.shape[0])
            edge = ev[[0, max(n - 2, 0)], 0].cpu().numpy() if self.exposure_mode == ExposureMode.DURATION else None
        else:
            host = np.ascontiguousarray(event_arr, dtype=np.float32)
            n = int(host.shape[0])
            ev = torch.from_numpy(host).to(dev)
            edge = host[[0, max(n - 2, 0)], 0] if self.exposure_mode == ExposureMode.DURATION else None
        lib = _capi.lib()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        mode = self.exposure_mode
        bounds_dev = seg_dev = None
        n_bounds = n_seg = 0
        count_per_frame = 1
        has_open = 1
        if mode == ExposureMode.DURATION:
            bounds = self._duration_bounds(edge[0], edge[1], n)
            n_complete = len(bounds) - 1
            bounds_dev = torch.tensor([float(b) for b in bounds], dtype=torch.float64).to(dev)
            n_bounds = len(bounds)
        elif mode == ExposureMode.COUNT:
            count_per_frame = int(self.event_count)
            n_complete = (n - 2) // count_per_frame if n >= 2 else 0  # frames [k c, (k + 1) c) with (k + 1) c < n - 1
        elif mode == ExposureMode.AREA_COUNT:
            nw = 1 + self.width // self.area_dimension
            nh = 1 + self.height // self.area_dimension
            if self.area_counts is None or tuple(self.area_counts.shape) != (nw, nh):
                self.area_counts = torch.zeros((nw, nh), dtype=torch.int32, device=dev)
            seg_dev = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
            out2 = torch.zeros((2,), dtype=torch.int32, device=dev)
            check(lib.v2e_render_area_segments(C.c_void_p(ev.data_ptr()), n, C.c_void_p(self.area_counts.data_ptr()), nw, nh,
                                               float(self.area_dimension), int(self.area_count), C.c_void_p(seg_dev.data_ptr()),
                                               int(seg_dev.numel()), C.c_void_p(out2.data_ptr()), stream), "v2e_render_area_segments")
            n_seg = n_complete = int(out2[0].item())
        else:  # SOURCE: the whole packet is one frame, and it is emitted
            n_complete, has_open = 1, 0
        npx = height * width
        diff = torch.empty((n_complete + has_open, npx), dtype=torch.int32, device=dev)
        frames = torch.empty((n_complete, height, width), dtype=torch.float64, device=dev)
        cur = torch.empty((height, width), dtype=torch.float64, device=dev) if has_open else None
        check(lib.v2e_render_packet(C.c_void_p(ev.data_ptr()), n - 1, mode.value,
                                    C.c_void_p(bounds_dev.data_ptr()) if bounds_dev is not None else None, n_bounds,
                                    count_per_frame, C.c_void_p(seg_dev.data_ptr()) if seg_dev is not None else None, n_seg,
                                    n_complete, has_open, C.c_void_p(diff.data_ptr()),
                                    C.c_void_p(frames.data_ptr()) if n_complete else None,
                                    C.c_void_p(cur.data_ptr()) if cur is not None else None, height, width,
                                    0.0, float(self.height), 0.0, float(self.width), float(self.full_scale_count), stream),
              "v2e_render_packet")
        self.currentFrame = cur
        if n_complete == 0 or not return_frames:
            return None
        return frames.cpu().numpy()
