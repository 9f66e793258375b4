"""Drop-in `EventRenderer`: DVS frames from events, the histograms on the GPU.

Mirrors v2ecore/renderer.py: `ExposureMode`, constructor signature (:37-99), `render_events_to_frames(event_arr,
height, width, return_frames) -> frames or None` (:161-366) with the frame being filled held between packets, and
`accumulate_event_frame` (:368-400).  What is re-built: the ON/OFF 2-D histograms and the clipped running frame
(`hist2d_numba_seq`, v2e_utils.py:474-486) as HIP kernels on the device-resident event packet, and the 0..1
normalisation.  What is kept on the host, statement for statement, is the exposure bookkeeping -- which events belong to
which frame (duration / count / area-count / source), including the reference's own edge behaviour (the last event of a
packet is never accumulated, renderer.py:303-306; `searchsorted` on the remaining time stamps) -- so that the frames
are the reference's frames.  AVI output and the preview window (cv2) are outside the hot path: `dvs_vid` / `preview`
raise NotImplementedError.
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

    def render_events_to_frames(self, event_arr, height: int, width: int, return_frames=False):
        """Incrementally render event frames (renderer.py:161-366).  `event_arr`: [n,4] events (ts, x, y, pol) as a host
        array or a device tensor.  Returns the frames filled by this packet as a float64 [n,h,w] numpy array in 0..1 (only
        if return_frames), or None."""
        self.width = width
        self.height = height
        if event_arr is None or event_arr.shape[0] == 0:
            if not self.printed_empty_packet_warning:
                logger.info('event_arr is None or there are no events, doing nothing, supressing further warnings')
                self.printed_empty_packet_warning = True
            return None
        if torch.is_tensor(event_arr):
            ev_dev = event_arr.to(self.device, torch.float32).contiguous()
            ev_host = None
            ts = ev_dev[:, 0].cpu().numpy()
        else:
            ev_host = np.ascontiguousarray(event_arr, dtype=np.float32)
            ev_dev = torch.from_numpy(ev_host).to(self.device)
            ts = ev_host[:, 0]
        if self.exposure_mode == ExposureMode.DURATION:
            if self.currentFrameStartTime is None:
                self.currentFrameStartTime = ts[0]
            nextFrameStartTs = self.currentFrameStartTime + self.frameIntevalS
        if self.exposure_mode == ExposureMode.AREA_COUNT and self.area_counts is None:
            nw = 1 + self.width // self.area_dimension
            nh = 1 + self.height // self.area_dimension
            self.area_counts = np.zeros(shape=(nw, nh), dtype=int)
        returned = []
        thisFrameIdx = 0
        numEvents = len(ts)
        histrange = np.asarray([(0, v) for v in (self.height, self.width)], dtype=np.int64)
        doneWithTheseEvents = False
        start, end = 0, numEvents
        self.currentFrame = None  # (renderer.py:273: reset at every packet, as the reference does)
        while not doneWithTheseEvents:
            if self.exposure_mode == ExposureMode.DURATION:
                rest = ts[thisFrameIdx:]
                start = int(np.searchsorted(rest, self.currentFrameStartTime, side="left"))
                end = int(np.searchsorted(rest, nextFrameStartTs, side="right"))
            elif self.exposure_mode == ExposureMode.COUNT:
                start = thisFrameIdx
                end = start + self.event_count
            elif self.exposure_mode == ExposureMode.AREA_COUNT:
                start = thisFrameIdx
                if ev_host is None:
                    ev_host = ev_dev.cpu().numpy()
                self.area_counts, end = self._compute_area_counts(ev_host, self.area_counts, self.area_count,
                                                                  self.area_dimension, start)
            elif self.exposure_mode == ExposureMode.SOURCE:
                start = 0
                end = numEvents
            if end >= numEvents - 1:
                doneWithTheseEvents = True
                end = numEvents - 1
            self.accumulate_event_frame(ev_dev[start:end], histrange)
            if not doneWithTheseEvents or self.exposure_mode == ExposureMode.SOURCE:
                if self.exposure_mode == ExposureMode.DURATION:
                    self.currentFrameStartTime += self.frameIntevalS
                    nextFrameStartTs = self.currentFrameStartTime + self.frameIntevalS
                elif self.exposure_mode == ExposureMode.COUNT or self.exposure_mode == ExposureMode.AREA_COUNT:
                    thisFrameIdx = end
                # img output is 0-1 range (renderer.py:245-247)
                img = torch.empty_like(self.currentFrame)
                check(_capi.lib().v2e_frame_normalize(C.c_void_p(self.currentFrame.data_ptr()), C.c_void_p(img.data_ptr()),
                                                      img.numel(), float(self.full_scale_count),
                                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                      "v2e_frame_normalize")
                self.currentFrame = None
                if return_frames:
                    returned.append(img)
                self.numFramesWritten += 0  # only a video file advances the reference's counter
        if self.currentFrame is not None:
            self.currentFrame = self.currentFrame.clone()  # detach from the accumulator, which the next packet zeroes
        if not returned:
            return None
        return torch.stack(returned).cpu().numpy()

    @staticmethod
    def _compute_area_counts(events, area_counts, area_count, area_dimension, start):
        """renderer.py:253-266 (a sequential scan by definition: host loop)."""
        ev_idx = start
        for ev_idx in range(start, events.shape[0]):
            x = int(events[ev_idx, 1] // area_dimension)
            y = int(events[ev_idx, 2] // area_dimension)
            count = 1 + area_counts[x, y]
            area_counts[x, y] = count
            if count >= area_count:
                area_counts = np.zeros_like(area_counts)
                break
        return area_counts, ev_idx
