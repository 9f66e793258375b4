"""Thin torch-facing wrapper of the emulator C ABI (include/v2e_amd.h).

PyTorch is plumbing here: it owns the HBM allocations (state planes, frame and event
buffers) and the HIP stream; all arithmetic happens in libv2e_amd.so.  `EmuEngine`
advances `n_clips` independent pixel arrays in lock-step; the drop-in
`EventEmulator` (v2e_amd/emulator.py) uses n_clips = 1.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _capi
from ._capi import EmuParams, FrameRec, check

_DT = {torch.uint8: _capi.DT_U8, torch.float32: _capi.DT_F32, torch.float64: _capi.DT_F64}


class _RowsOwner:
    """What a result array of the frame API hangs on: exposes a pinned buffer's first n rows through the array interface
    and keeps the buffer out of the pool until the array AND every view derived from it are gone (numpy makes derived views
    hold the first array, which holds this object)."""
    __slots__ = ("__array_interface__", "buf", "__weakref__")

    def __init__(self, buf, n):
        self.buf = buf
        self.__array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (buf.ptr, False), "version": 3}


class _RowsBuf:
    __slots__ = ("tensor", "ptr", "cap", "free")

    def __init__(self, cap):
        self.tensor = torch.empty((cap, 4), dtype=torch.float32, pin_memory=True)
        self.ptr, self.cap, self.free = self.tensor.data_ptr(), cap, True


class _RowsPool:
    """Pinned host buffers the frame API's last kernel writes the event rows into, handed out AS the result arrays
    (`generate_events` returns a fresh array the caller owns: here one whose memory returns to the pool when the caller drops
    it).  A caller that keeps its arrays makes the pool grow to MAX_BUFS buffers; beyond that, and for frames larger than a
    buffer, the rows are copied into an ordinary array as before."""
    MAX_BUFS = 64
    MAX_BYTES = 1 << 30  # page-locked memory the pool may hold in all (a caller that keeps every array gets copies beyond it)
    MIN_ROWS = 1 << 16   # 1 MB

    def __init__(self):
        self.bufs = []

    def acquire(self, est_rows):
        want = max(self.MIN_ROWS, 1 << int(2 * est_rows).bit_length())
        for b in self.bufs:
            if b.free and b.cap >= want // 2:
                b.free = False
                return b
        held = sum(b.cap for b in self.bufs) * 16
        if len(self.bufs) >= self.MAX_BUFS or held + want * 16 > self.MAX_BYTES:
            # make room by dropping a free buffer that is too small (frames grew), else hand out nothing: the frame's rows
            # are then copied into an ordinary array
            small = [b for b in self.bufs if b.free]
            if not small:
                return None
            self.bufs.remove(small[0])
            if held - small[0].cap * 16 + want * 16 > self.MAX_BYTES:
                return None
        b = _RowsBuf(want)
        b.free = False
        self.bufs.append(b)
        return b

    @staticmethod
    def release(buf):
        buf.free = True

    def hand_out(self, buf, n):
        owner = _RowsOwner(buf, n)
        weakref.finalize(owner, _RowsPool.release, buf)
        return np.asarray(owner)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:  # older / other torch builds
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dbl_array(vals):
    arr = (C.c_double * len(vals))(*[float(v) for v in vals])
    return arr


_REC_DT = np.dtype([("max_events", "<i4"), ("flags", "<u4"), ("n_signal", "<u4"), ("n_events", "<u4"),
                    ("n_on", "<u4"), ("n_off", "<u4"), ("ev_offset", "<u8")])


class EmuEngine:
    """Device state + kernels of `n_clips` DVS pixel arrays of size H x W."""

    def __init__(self, H, W, n_clips=1, device="cuda", max_iters=64):
        self.H, self.W, self.n_clips = int(H), int(W), int(n_clips)
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _capi.V2EAmdError(
                "v2e_amd needs a ROCm GPU (device=%r, torch.cuda.is_available()=%s); "
                "there is no CPU fallback" % (device, torch.cuda.is_available()))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _capi.lib()
        self.max_iters = int(max_iters)
        self.npx = self.H * self.W
        self.npx_pad = int(self.lib.v2e_emu_npx_pad(self.H, self.W))
        h = C.c_void_p()
        check(self.lib.v2e_emu_create(self.H, self.W, self.n_clips, int(max_iters), self.device.index,
                                      C.byref(h)), "v2e_emu_create")
        self._h = h
        self.f64_state = None
        self.lp = self.base = self.ts_mem = self.pos_thres = self.neg_thres = self.noise_rate = None
        self._events = None
        self._events_tmp = None
        self._dev_index = self.device.index
        self._rows_pool = _RowsPool()  # pinned result arrays of the frame API
        self._rows_est = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.v2e_emu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        # (the raw-handle call: torch.cuda.current_stream() builds a Stream object, 2 us a call on the frame API's path)
        return C.c_void_p(_raw_stream(self._dev_index))

    # ------------------------------------------------------------ state
    def alloc_state(self, f64_state):
        """(Re)allocate the per-pixel planes [n_clips][npx_pad] and bind them."""
        self.f64_state = bool(f64_state)
        sd = torch.float64 if self.f64_state else torch.float32
        shp = (self.n_clips, self.npx_pad)
        dev = self.device
        self.lp = torch.zeros(shp, dtype=sd, device=dev)
        self.base = torch.zeros(shp, dtype=sd, device=dev)
        self.ts_mem = torch.zeros(shp, dtype=torch.float32, device=dev)
        self.pos_thres = torch.zeros(shp, dtype=torch.float32, device=dev)
        self.neg_thres = torch.zeros(shp, dtype=torch.float32, device=dev)
        self.noise_rate = torch.zeros(shp, dtype=torch.float32, device=dev)
        check(self.lib.v2e_emu_bind_state(self._h, _ptr(self.lp), _ptr(self.base), _ptr(self.ts_mem),
                                          _ptr(self.pos_thres), _ptr(self.neg_thres),
                                          _ptr(self.noise_rate)), "v2e_emu_bind_state")

    def plane(self, t, clip=0):
        """[H,W] view of a state plane."""
        return t[clip, :self.npx].view(self.H, self.W)

    # ----------------------------------------------------------- frames
    def to_device_frame(self, frame):
        """np.ndarray / torch tensor [n_clips?,H,W] -> contiguous device tensor u8/f32/f64.

        Anything that is not uint8/float32/float64 is converted to float64 on the way,
        which is what `torch.tensor(new_frame, dtype=torch.float64)` (emulator.py:663) does.
        """
        if isinstance(frame, np.ndarray):
            if frame.dtype not in (np.uint8, np.float32, np.float64):
                frame = frame.astype(np.float64)
            t = torch.from_numpy(np.ascontiguousarray(frame))
        elif torch.is_tensor(frame):
            t = frame
            if t.dtype not in _DT:
                t = t.to(torch.float64)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(frame, dtype=np.float64)))
        t = t.to(self.device, non_blocking=True).contiguous()
        return t

    def _check_frame(self, t, lead):
        if t.numel() != lead * self.npx:
            raise ValueError("frame has %d elements, expected %d x %d x %d" % (t.numel(), lead, self.H, self.W))

    # ------------------------------------------------------------ calls
    def init_state(self, P, frame_dev, t_frame, thres_pos=None, thres_neg=None, noise_rate=None):
        self._check_frame(frame_dev, self.n_clips)
        check(self.lib.v2e_emu_init_state(self._h, C.byref(P), _ptr(frame_dev), _DT[frame_dev.dtype],
                                          float(t_frame), _ptr(thres_pos), _ptr(thres_neg), _ptr(noise_rate),
                                          self.stream), "v2e_emu_init_state")

    def count(self, P, frame_dev, t_prev, t_frame, frame_idx, leak_randn=None, shot_rand=None):
        self._check_frame(frame_dev, self.n_clips)
        check(self.lib.v2e_emu_count(self._h, C.byref(P), _ptr(frame_dev), _DT[frame_dev.dtype],
                                     _dbl_array(t_prev), _dbl_array(t_frame), int(frame_idx),
                                     _ptr(leak_randn), _ptr(shot_rand), self.stream), "v2e_emu_count")

    def set_csdvs(self, surround_plane):
        """cs_surround_frame plane [n_clips][npx_pad] (state dtype) that count() subtracts; None switches CSDVS off."""
        check(self.lib.v2e_emu_set_csdvs(self._h, _ptr(surround_plane)), "v2e_emu_set_csdvs")

    def lp_preview(self, P, frame_dev, t_prev, t_frame, frame_idx, lp_out):
        """The coming frame's lp_log_frame into lp_out, the state untouched (v2e_emu_lp_preview)."""
        self._check_frame(frame_dev, self.n_clips)
        check(self.lib.v2e_emu_lp_preview(self._h, C.byref(P), _ptr(frame_dev), _DT[frame_dev.dtype], _dbl_array(t_prev),
                                          _dbl_array(t_frame), int(frame_idx), _ptr(lp_out), self.stream), "v2e_emu_lp_preview")

    def csdvs_update(self, p_plane, h_plane, h_scratch, alpha_p, alpha_h, num_steps, stop):
        """EventEmulator._update_csdvs's stepping loop on device planes (v2e_csdvs_update); returns the steps taken."""
        steps = C.c_int(0)
        check(self.lib.v2e_csdvs_update(_ptr(p_plane), _ptr(h_plane), _ptr(h_scratch), self.H, self.W, 1 if self.f64_state else 0,
                                        float(alpha_p), float(alpha_h), int(num_steps), float(stop), C.byref(steps), None,
                                        self.stream), "v2e_csdvs_update")
        return int(steps.value)

    def set_pnoise(self, pn_plane, randn_tape=None):
        """photoreceptor_noise_arr plane [n_clips][npx_pad] float64 (+ the frame's torch.randn draws in tape mode) for
        the next count(); None switches the feature off."""
        check(self.lib.v2e_emu_set_pnoise(self._h, _ptr(pn_plane), _ptr(randn_tape)), "v2e_emu_set_pnoise")

    def set_model_state_planes(self, planes):
        """Three float64 planes [n_clips][npx_pad] (log_new_frame, c_minus_s_frame, diff_frame) every count() fills from
        now on; None switches it off (v2e_emu_set_model_state_planes)."""
        a, b, c = planes if planes is not None else (None, None, None)
        check(self.lib.v2e_emu_set_model_state_planes(self._h, _ptr(a), _ptr(b), _ptr(c)), "v2e_emu_set_model_state_planes")

    def shot(self, P, frame_dev, frame_idx, shot_rand):
        check(self.lib.v2e_emu_shot(self._h, C.byref(P), _ptr(frame_dev), _DT[frame_dev.dtype],
                                    int(frame_idx), _ptr(shot_rand), self.stream), "v2e_emu_shot")

    def read_rec(self, frame_idx):
        recs = (FrameRec * self.n_clips)()
        check(self.lib.v2e_emu_read_rec(self._h, int(frame_idx), recs, self.stream), "v2e_emu_read_rec")
        return recs

    def reserve_iters(self, max_events):
        check(self.lib.v2e_emu_reserve_iters(self._h, int(max_events), self.stream), "v2e_emu_reserve_iters")

    def rank(self, P, frame_idx, ts_table=None):
        n_ts = 0 if ts_table is None else int(ts_table.shape[-1])
        check(self.lib.v2e_emu_rank(self._h, C.byref(P), int(frame_idx), _ptr(ts_table), n_ts, self.stream),
              "v2e_emu_rank")

    def read_iter_counts(self, frame_idx, n_iters):
        nk = 2 * (n_iters + 1)
        out = (C.c_uint32 * (nk * self.n_clips))()
        check(self.lib.v2e_emu_read_iter_counts(self._h, int(frame_idx), int(n_iters), out, self.stream),
              "v2e_emu_read_iter_counts")
        return np.frombuffer(out, dtype=np.uint32).reshape(self.n_clips, n_iters + 1, 2).copy()

    def event_buffer(self, cap, which=0):
        """Device [n_clips][cap][4] float32 buffer, grown geometrically (`which`: one of several sets, so that the events of
        a run can still be read while the next run writes another set; set 0 is also the frame-at-a-time API's)."""
        if which:
            sets = self.__dict__.setdefault("_events_sets", {})
            cur = sets.get(which)
            if cur is None or cur.shape[1] < cap:
                ncap = max(int(cap), 1024) if cur is None else max(int(cap), 2 * cur.shape[1])
                cur = sets[which] = torch.empty((self.n_clips, ncap, 4), dtype=torch.float32, device=self.device)
            return cur
        if self._events is None or self._events.shape[1] < cap:
            ncap = max(int(cap), 1024)
            if self._events is not None:
                ncap = max(ncap, 2 * self._events.shape[1])
            self._events = torch.empty((self.n_clips, ncap, 4), dtype=torch.float32, device=self.device)
            self._events_tmp = None
        return self._events

    def emit(self, P, frame_idx, events, ts_table=None, ev_offset0=None):
        n_ts = 0 if ts_table is None else int(ts_table.shape[-1])
        off = None
        if ev_offset0 is not None:
            off = (C.c_uint64 * self.n_clips)(*[int(v) for v in ev_offset0])
        check(self.lib.v2e_emu_emit(self._h, C.byref(P), int(frame_idx), _ptr(ts_table), n_ts, _ptr(events),
                                    int(events.shape[1]), off, self.stream), "v2e_emu_emit")

    def permute(self, events_in, events_out, idx_dev, row0, n):
        check(self.lib.v2e_emu_permute(self._h, _ptr(events_in), _ptr(events_out), _ptr(idx_dev), int(row0),
                                       int(n), self.stream), "v2e_emu_permute")

    def frame(self, P, frame, t_prev, t_frame, frame_idx, events):
        """One frame in one C call (v2e_emu_frame; Philox mode, one clip): `frame` a C-contiguous uint8/float32/float64
        numpy array (copied through the handle's pinned staging) or a device tensor.  Returns (rc, out8, rows) with rows a
        fresh host [n,4] float32 array (or None): rc 0 done; 1 more iteration scratch needed (M = out8[4]); 2 the event
        buffer is too small for out8[0] rows -- in both cases the frame is counted and nothing was emitted."""
        if isinstance(frame, np.ndarray):
            fp, on_host, dt = C.c_void_p(frame.ctypes.data), 1, {np.dtype(np.uint8): _capi.DT_U8, np.dtype(np.float32): _capi.DT_F32,
                                                                 np.dtype(np.float64): _capi.DT_F64}[frame.dtype]
            n_el = frame.size
        else:
            fp, on_host, dt, n_el = _ptr(frame), 0, _DT[frame.dtype], frame.numel()
        if n_el != self.npx:
            raise ValueError("frame has %d elements, expected %d x %d" % (n_el, self.H, self.W))
        out8 = (C.c_uint32 * 8)()
        rows = C.POINTER(C.c_float)()
        # the rows land in a pinned buffer of ours that becomes the result array itself (no host copy); see _RowsPool
        pool = self._rows_pool
        buf = pool.acquire(self._rows_est)
        try:  # whatever raises below, the buffer goes back to the pool unless it has become the result array
            if buf is not None:
                check(self.lib.v2e_emu_frame_host_rows(self._h, C.c_void_p(buf.ptr), buf.cap), "v2e_emu_frame_host_rows")
            rc = self.lib.v2e_emu_frame(self._h, C.byref(P), fp, on_host, dt, float(t_prev), float(t_frame), int(frame_idx),
                                        _ptr(events), int(events.shape[1]), out8, C.byref(rows), self.stream)
            if rc < 0:
                check(rc, "v2e_emu_frame")
            ev = None
            n = int(out8[0])
            if rc == 0 and n > 0:
                self._rows_est = n + n // 4 + 256
                if buf is not None and C.cast(rows, C.c_void_p).value == buf.ptr:
                    ev = pool.hand_out(buf, n)
                    buf = None
                else:
                    # the handle's own pinned rows are reused by the next call (memmove, not ctypeslib.as_array: that builds a
                    # ctypes array type per distinct row count, ~100 us a frame)
                    ev = np.empty((n, 4), dtype=np.float32)
                    C.memmove(ev.ctypes.data, rows, 16 * n)
            return rc, out8, ev
        finally:
            if buf is not None:
                pool.release(buf)

    def run(self, P, frames_dev, t_prev, t_frame, frame_idx0, events, recs_dev, use_graph=True):
        """Device-resident Philox run over frames_dev [F][n_clips][H*W]; no host sync."""
        F = int(frames_dev.shape[0])
        self._check_frame(frames_dev, F * self.n_clips)
        tp = np.ascontiguousarray(np.asarray(t_prev, dtype=np.float64).reshape(F * self.n_clips))
        tf = np.ascontiguousarray(np.asarray(t_frame, dtype=np.float64).reshape(F * self.n_clips))
        check(self.lib.v2e_emu_run(self._h, C.byref(P), _ptr(frames_dev), _DT[frames_dev.dtype], F,
                                   tp.ctypes.data_as(C.POINTER(C.c_double)),
                                   tf.ctypes.data_as(C.POINTER(C.c_double)), int(frame_idx0), _ptr(events),
                                   int(events.shape[1]), _ptr(recs_dev), int(use_graph), self.stream),
              "v2e_emu_run")

    def run_done_event(self):
        """A torch event behind every piece of every overlapped run (use_graph | 1024) enqueued so far: recorded on a stream of the
        engine that waits for them (v2e_emu_run_join), never on the caller's stream -- which carries the next run's chain."""
        js = self.__dict__.get("_join_stream")
        if js is None:
            js = self._join_stream = torch.cuda.Stream(self.device)
        e0 = torch.cuda.Event()  # what the caller's stream holds (the run's chain; the whole run where the library did not overlap it)
        e0.record(torch.cuda.current_stream(self.device))
        js.wait_event(e0)
        check(self.lib.v2e_emu_run_join(self._h, js.cuda_stream), "v2e_emu_run_join")
        ev = torch.cuda.Event()
        ev.record(js)
        return ev

    def launch_stamps(self, runs, read=0):
        """Device time stamps of the chain kernel's launches (v2e_emu_launch_stamps): keep `runs` runs from now on (0: off); with
        read > 0 first return the last `read` runs as an array [run][launch][2] of nanoseconds {first workgroup's start, last
        workgroup's end} (zeros: no such launch)."""
        out = None
        if read > 0:
            n, lpr = C.c_int(), C.c_int()
            buf = np.zeros((read, 128, 2), dtype=np.uint64)
            check(self.lib.v2e_emu_launch_stamps(self._h, int(runs), buf.ctypes.data_as(C.c_void_p), int(read), C.byref(n), C.byref(lpr)),
                  "v2e_emu_launch_stamps")
            assert lpr.value == 128
            out = buf[:n.value]
        else:
            check(self.lib.v2e_emu_launch_stamps(self._h, int(runs), None, 0, None, None), "v2e_emu_launch_stamps")
        return out

    def last_profile(self):
        """Kernel-class times (ms) of the last run(use_graph=2); see v2e_emu_last_profile[_pipe]."""
        v = [C.c_double() for _ in range(4)]
        n = C.c_int()
        check(self.lib.v2e_emu_last_profile(self._h, *[C.byref(x) for x in v], C.byref(n)), "v2e_emu_last_profile")
        nb, fpb, nsl = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.v2e_emu_last_profile_pipe(self._h, C.byref(nb), C.byref(fpb), C.byref(nsl)), "v2e_emu_last_profile_pipe")
        nl = C.c_int()
        us = (C.c_float * 4096)()
        check(self.lib.v2e_emu_last_profile_launches(self._h, us, 4096, C.byref(nl)), "v2e_emu_last_profile_launches")
        return dict(count=v[0].value, rank=v[1].value, scan=v[2].value, emit=v[3].value, launches=n.value,
                    emit_batches=nb.value, frames_per_batch=fpb.value, step_launches=nsl.value,
                    chain_launch_us=[float(us[i]) for i in range(min(nl.value, 4096))])

    def event_writer(self):
        """'k_cpull' / 'k_cemit' / None: the event writer the handle's k_chain pipeline was given (v2e_emu_event_writer)."""
        return {1: "k_cpull", 0: "k_cemit"}.get(int(self.lib.v2e_emu_event_writer(self._h)))

    def last_pipeline(self):
        """(kind, frames per chain launch, frames per emission batch) of the last run(); see v2e_emu_last_pipeline."""
        k, a, b = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.v2e_emu_last_pipeline(self._h, C.byref(k), C.byref(a), C.byref(b)), "v2e_emu_last_pipeline")
        names = {0: "k_count/k_rank/k_scan/k_emit", 3: "k_chain", 4: "k_chain(fused records)"}
        return names.get(k.value, "?"), a.value, b.value

    def alloc_recs(self, n_frames, which=0):
        """Device record array [F][n_clips] (struct v2e_frame_rec = 32 bytes), cached per (F, set) so that
        the hipGraph of run() (which bakes the pointer in) stays valid across calls."""
        cache = self.__dict__.setdefault("_recs_cache", {})
        key = (n_frames, which)
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = torch.zeros((n_frames, self.n_clips, C.sizeof(FrameRec)), dtype=torch.uint8,
                                     device=self.device)
        return cache[key]

    def run_ticket(self):
        return int(self.lib.v2e_emu_run_ticket(self._h))

    def run_wait(self, ticket):
        check(self.lib.v2e_emu_run_wait(self._h, int(ticket)), "v2e_emu_run_wait")

    def run_recs(self, ticket):
        """Records of the pipelined run with that ticket (behind run_wait) as a structured array [n_frames][n_clips], copied out of the
        handle's pinned buffer."""
        n = C.c_uint64()
        ptr = self.lib.v2e_emu_run_recs(self._h, int(ticket), C.byref(n))
        if not ptr:
            raise _capi.V2EAmdError("no records for that ticket")
        raw = np.frombuffer(C.string_at(ptr, int(n.value) * C.sizeof(FrameRec)), dtype=_REC_DT)
        return raw.reshape(-1, self.n_clips)

    def read_recs_after(self, recs_dev, done_event):
        """Records of a run as a structured numpy array, copied on a side stream once `done_event` (recorded behind the run)
        has completed: later runs already enqueued on the main stream are not waited for."""
        cs = self.__dict__.get("_copy_stream")
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(self.device)
        host = self.__dict__.setdefault("_recs_host", {})
        key = tuple(recs_dev.shape)
        if key not in host:
            host[key] = torch.empty(recs_dev.shape, dtype=torch.uint8).pin_memory()
        with torch.cuda.stream(cs):
            if done_event is not None:
                cs.wait_event(done_event)
            host[key].copy_(recs_dev, non_blocking=True)
            cs.synchronize()
        a = host[key].numpy().copy()
        return a.view(_REC_DT).reshape(a.shape[0], a.shape[1])

    @staticmethod
    def recs_to_numpy(recs_dev):
        dt = np.dtype([("max_events", "<i4"), ("flags", "<u4"), ("n_signal", "<u4"), ("n_events", "<u4"),
                       ("n_on", "<u4"), ("n_off", "<u4"), ("ev_offset", "<u8")])
        a = recs_dev.cpu().numpy()
        return a.view(dt).reshape(a.shape[0], a.shape[1])
