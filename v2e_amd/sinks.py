"""Device-side packing of events into the reference writers' integer formats."""
import logging
import ctypes as C

import torch

from . import _capi
from ._capi import check

# aedat2_output.py:41-77
_AEDAT2_LAYOUT = {(346, 260): (12, 22, 11, 1, 1), (240, 180): (12, 22, 11, 1, 1), (640, 480): (1, 11, 0, 1, 1)}

logger = logging.getLogger(__name__)


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


class DeviceAEDat2Output:
    """AEDAT-2.0 file sink fed from the device-resident event buffer (mirrors v2ecore/output/aedat2_output.py: same
    header lines, same address layout per sensor size, big-endian (address, timestamp) int32 pairs, noise events marked
    as special events with label_signal_noise).  The integer conversion and the byte order are done on the GPU
    (v2e_events_pack_aedat2, byte-identical to the reference writer: tests/test_sinks.py); the host only moves bytes."""

    def __init__(self, filepath, output_width=346, output_height=260, label_signal_noise=False):
        import atexit
        if (output_width, output_height) not in _AEDAT2_LAYOUT:
            raise ValueError(f'AEDAT-2.0 output width={output_width} height={output_height} not supported')  # aedat2_output.py:78-80
        self.filepath = filepath
        self.width, self.height = output_width, output_height
        self.label_signal_noise = label_signal_noise
        self.numEventsWritten = self.numOnEvents = self.numOffEvents = 0
        self.file = open(filepath, 'wb')
        self._writeHeader()
        atexit.register(self.cleanup)

    def cleanup(self):
        self.close()

    def close(self):
        if self.file:
            self.file.close()
            self.file = None

    def _writeHeader(self):  # aedat2_output.py:111-133, CRLF line ends (jAER)
        import datetime
        import getpass
        import time
        date = datetime.datetime.now().strftime('# Creation time: %I:%M%p %B %d %Y\r\n')
        tms = '# Creation time: System.currentTimeMillis() {}\r\n'.format(int(time.time() * 1000.))
        user = '# User name: {}\r\n'.format(getpass.getuser())
        sn_comment = ('# noise events are labeled as addressed external input events when the --label_signal_noise option is '
                      'selected for output\r\n') if self.label_signal_noise else ''
        header = ('#!AER-DAT2.0\r\n',
                  '# This is a raw AE data file created by AEDat2Output in v2e (see https://github.com/SensorsINI/v2e) as '
                  'specified at https://inivation.com/support/software/fileformat/#aedat-20\r\n',
                  '# Data format is int32 address, int32 timestamp (8 bytes total), repeated for each event\r\n',
                  '# Timestamps tick is 1 us\r\n', sn_comment, date, tms, user)
        for line in header:
            self.file.write(line.encode('UTF-8'))

    def appendEvents(self, events, signnoise_label=None, n_signal=None):
        """events: [N,4] float32 rows (t, x, y, p) as a device tensor or a host array; noise labelling either by the
        reference's boolean array (True = signal) or by `n_signal` (rows from n_signal on are noise, how the emulator
        orders a frame's events)."""
        if self.file is None or events is None or len(events) == 0:
            return
        ev = events if torch.is_tensor(events) else torch.from_numpy(events.astype('float32', copy=False))
        if not ev.is_cuda:
            ev = ev.cuda()
        n = int(ev.shape[0])
        noise_from = -1
        if self.label_signal_noise:
            if n_signal is not None:
                noise_from = int(n_signal)
            elif signnoise_label is not None:  # general labels: sort is not needed, mark per run of equal labels
                lab = torch.as_tensor(signnoise_label).bool().cpu().numpy()
                out = bytearray()
                i = 0
                while i < n:
                    j = i
                    while j < n and lab[j] == lab[i]:
                        j += 1
                    out += bytes(pack_aedat2(ev[i:j], self.width, self.height, noise_from=-1 if lab[i] else 0).cpu().numpy())
                    i = j
                data = bytes(out)
                self._write(data, ev)
                return
        data = pack_aedat2(ev, self.width, self.height, noise_from=noise_from).cpu().numpy().tobytes()
        self._write(data, ev)

    def _write(self, data, ev):
        n = len(data) // 8
        if self.numEventsWritten == 0:  # aedat2_output.py:173-179: a first byte '#' would read as a comment line
            while data[0:1] == b'#':
                data = data[8:]
        self.file.write(data)
        self.numEventsWritten += n
        on = int((ev[:, 3] > 0).sum().item())
        self.numOnEvents += on
        self.numOffEvents += n - on
        self.file.flush()


class DeviceTextOutput:
    """Text event sink (v2ecore/output/ae_text_output.py:51-101: `t x y p[ label]` per line, p in {0, 1}, t the Python
    repr of the float64 value of the float32 time stamp).  The integer columns are converted on the GPU (same kernel as
    the HDF5 rows); producing a shortest round-trip decimal of t is Python's float repr, on the host."""

    def __init__(self, filepath, label_signal_noise=False):
        import atexit
        self.filepath = filepath
        self.label_signal_noise = label_signal_noise
        self.numEventsWritten = 0
        self.file = open(filepath, 'w')
        self._writeHeader()
        atexit.register(self.cleanup)

    def cleanup(self):
        self.close()

    def close(self):
        if self.file:
            self.file.close()
            self.file = None

    def _writeHeader(self):  # ae_text_output.py:51-66
        import datetime
        import getpass
        import time
        if not self.label_signal_noise:
            fmt = '# Format is time (float s), x, y, polarity (0=off, 1=on) as specified at http://rpg.ifi.uzh.ch/davis_data.html\n'
        else:
            fmt = ('# Format is time (float s), x, y, polarity (0=off, 1=on), signal/noise (1/0)\n#  as specified at '
                   'http://rpg.ifi.uzh.ch/davis_data.html\n')
        date = datetime.datetime.now().strftime('# Creation time: %I:%M%p %B %d %Y\n')
        tms = '# Creation time: System.currentTimeMillis() {}\n'.format(int(time.time() * 1000.))
        user = '# User name: {}\n'.format(getpass.getuser())
        for line in ('#!events.txt\n', '# This is a text DVS created by v2e (see https://github.com/SensorsINI/v2e)\n', fmt, date,
                     tms, user):
            self.file.write(line)

    def appendEvents(self, events, signnoise_label=None, n_signal=None):
        if self.file is None:
            raise Exception('output file closed already')
        if events is None or len(events) == 0:
            return
        ev = events if torch.is_tensor(events) else torch.from_numpy(events.astype('float32', copy=False))
        if not ev.is_cuda:
            ev = ev.cuda()
        n = int(ev.shape[0])
        cols = pack_h5(ev).cpu().numpy()             # x, y as int32, p: -1 -> 0, +1 -> 1 (same as ((p + 1) / 2).astype(int32))
        t = ev[:, 0].double().cpu().numpy()          # events[:, 0].astype(float)
        lab = None
        if self.label_signal_noise:
            if n_signal is not None:
                lab = [1 if i < int(n_signal) else 0 for i in range(n)]
            elif signnoise_label is not None:
                lab = [int(v) for v in signnoise_label]
        if lab is None:
            self.file.writelines('{} {} {} {}\n'.format(t[i], cols[i, 1], cols[i, 2], cols[i, 3]) for i in range(n))
        else:
            self.file.writelines('{} {} {} {} {}\n'.format(t[i], cols[i, 1], cols[i, 2], cols[i, 3], lab[i]) for i in range(n))
        self.numEventsWritten += n


class HostAEDat4Output:
    """AEDAT-4.0 sink (v2ecore/output/aedat4_output.py:17-99) over the third-party `dv_processing` package the reference
    requires for this format (requirements.txt: dv-processing >= 1.7.8; not part of either tree).  Same conversions as the
    reference's appendEvents -- t = int(t * 1e6) in the precision this host's numpy gives a float32 scalar times a Python float
    (float64 before numpy 2, float32 from numpy 2 on), p = int((p + 1) / 2), no flips, one EventStore written at close() -- vectorised on the host; the file
    container (flatbuffers + compression) is dv_processing's.  Raises NotImplementedError where dv_processing is not importable:
    pass any object with appendEvents(events, signnoise_label=None) / close() as `dvs_aedat4` instead."""

    def __init__(self, filepath, output_width=640, output_height=480):
        try:
            import dv_processing as dv
        except ImportError as e:
            raise NotImplementedError("dvs_aedat4 needs the dv_processing package (%s); alternatively pass a writer object with "
                                      "appendEvents(events, signnoise_label=None) and close() as dvs_aedat4" % e)
        self.filepath = filepath
        self.numEventsWritten = self.numOnEvents = self.numOffEvents = 0
        self.sizex, self.sizey = output_width, output_height
        self.store = dv.EventStore()
        config = dv.io.MonoCameraWriter.EventOnlyConfig("DVXplorer_sample", (640, 480))  # aedat4_output.py:36-38
        self.writer = dv.io.MonoCameraWriter(filepath, config)

    def cleanup(self):
        self.close()

    def close(self):
        if self.writer:
            self.writer.writeEvents(self.store)
            self.writer = None

    def appendEvents(self, events, signnoise_label=None):
        import numpy as np
        if self.writer is None or events is None or len(events) == 0:
            return
        ev = events.detach().cpu().numpy() if torch.is_tensor(events) else np.asarray(events)
        # aedat4_output.py:82 `int(event[0] * 1e6)`: a float32 SCALAR times a Python float -- float64 under numpy < 2 (value-based
        # promotion), float32 under numpy >= 2 (NEP 50: the Python float is weak); this host's numpy decides, as it would for the reference
        if int(np.__version__.split(".")[0]) >= 2:
            t = (ev[:, 0].astype(np.float32) * np.float32(1e6)).astype(np.int64)
        else:
            t = (ev[:, 0].astype(np.float64) * 1e6).astype(np.int64)
        x = ev[:, 1].astype(np.int64)
        y = ev[:, 2].astype(np.int64)
        p = ((ev[:, 3].astype(np.float64) + 1) / 2).astype(np.int64)
        for i in range(ev.shape[0]):
            try:  # aedat4_output.py:84-87: an event the store refuses (a time stamp that goes backwards) is logged and skipped
                self.store.push_back(int(t[i]), int(x[i]), int(y[i]), int(p[i]))
            except RuntimeError as e:
                logger.warning('caught exception event {} to store'.format(e))
        on = int((p == 1).sum())
        self.numOnEvents += on
        self.numOffEvents += int(ev.shape[0]) - on
        self.numEventsWritten += int(ev.shape[0])


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
