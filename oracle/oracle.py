"""Python driver of the CPU oracle (oracle/emu_oracle.c, oracle/slomo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package v2e_amd/.

`OracleEmulator` restates EventEmulator.generate_events (v2ecore/emulator.py:619-1022)
around the C passes, including the reference's RNG call order:
frame 0: normal, normal, randn; frame k: randn, randperm per non-empty iteration,
rand.  Random numbers come from a *tape source*: `TorchTape` draws from torch's
global CPU generator exactly like the reference; `RecordedTape` replays stored
draws; `PhiloxTape` is None (in-"kernel" Philox, include/v2e_detmath.h).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libv2e_oracle.so")

import sys
sys.path.insert(0, os.path.dirname(_HERE))
from v2e_amd._capi import EmuParams, FrameRec, RNG_TAPE, RNG_PHILOX  # noqa: E402  (struct layouts only)

_lib = None


def build(force=False):
    """make is a no-op when the library is newer than its sources; a stale library (a symbol added since) is rebuilt."""
    try:
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        if not os.path.isfile(_LIB):
            raise


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.v2e_oracle_det_logf.restype = C.c_float
        _lib.v2e_oracle_det_logf.argtypes = [C.c_float]
        _lib.v2e_oracle_det_expf.restype = C.c_float
        _lib.v2e_oracle_det_expf.argtypes = [C.c_float]
        _lib.v2e_oracle_normal.restype = C.c_float
        _lib.v2e_oracle_normal.argtypes = [C.c_uint32, C.c_uint32]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ tapes
class TorchTape:
    """Live draws from torch's global CPU generator, same calls as the reference."""

    def normal(self, mean, std, shape):
        import torch
        return torch.normal(mean, std, size=shape, dtype=torch.float32).numpy()

    def randn(self, shape):
        import torch
        return torch.randn(shape, dtype=torch.float32).numpy()

    def rand(self, shape):
        import torch
        return torch.rand(size=shape, dtype=torch.float32).numpy()

    def randperm(self, n, frame=None, it=None):
        import torch
        return torch.randperm(n).numpy()

    def linspace(self, start, end, n):
        import torch
        return torch.linspace(start=start, end=end, steps=n, dtype=torch.float32).numpy()

    def exp_noise_rate(self, cov, randn):
        import torch
        return torch.exp(math.log(10) * cov * torch.from_numpy(randn)).numpy()

    def exp_scidvs(self, draw):  # emulator.py:480-483: torch.exp of the float32 normal draw
        import torch
        return torch.exp(torch.from_numpy(draw)).numpy()


class RecordedTape:
    """Replays a list of (kind, array) draws recorded from a reference run."""

    def __init__(self, items):
        self.items = list(items)
        self.pos = 0

    def _next(self, kind):
        k, a = self.items[self.pos]
        assert k == kind, "tape out of order: want %s got %s at %d" % (kind, k, self.pos)
        self.pos += 1
        return a

    def normal(self, mean, std, shape):
        return self._next("normal").reshape(shape)

    def randn(self, shape):
        return self._next("randn").reshape(shape)

    def rand(self, shape):
        return self._next("rand").reshape(shape)

    def randperm(self, n, frame=None, it=None):
        a = self._next("randperm")
        assert a.shape[0] == n
        return a

    def linspace(self, start, end, n):
        a = self._next("linspace")
        assert a.shape[0] == n
        return a

    def exp_noise_rate(self, cov, randn):
        return self._next("noise_rate").reshape(randn.shape)

    def exp_scidvs(self, draw):  # torch.exp of the SCIDVS time-constant draw (recorded like every torch.exp on the path)
        return self._next("noise_rate").reshape(draw.shape)


def philox_frame(seed, clip, frame, npx):
    a = np.empty(npx, np.float32)
    b = np.empty(npx, np.float32)
    lib().v2e_oracle_philox_frame(C.c_uint64(seed), C.c_uint32(clip), C.c_uint32(frame),
                                  C.c_int64(npx), _p(a), _p(b))
    return a, b


def philox_pnoise(seed, clip, frame, npx):
    a = np.empty(npx, np.float32)
    lib().v2e_oracle_philox_pnoise(C.c_uint64(seed), C.c_uint32(clip), C.c_uint32(frame), C.c_int64(npx), _p(a))
    return a


def philox_init(seed, clip, npx):
    a = np.empty(npx, np.float32)
    b = np.empty(npx, np.float32)
    c = np.empty(npx, np.float32)
    lib().v2e_oracle_philox_init(C.c_uint64(seed), C.c_uint32(clip), C.c_int64(npx), _p(a), _p(b), _p(c))
    return a, b, c


def sleef_sinhf(x):
    """torch.sinh of a float32 CPU tensor, bit for bit (v2e_sleef_sinhf, include/v2e_detmath.h)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().v2e_oracle_sleef_sinhf(_p(x), _p(y), C.c_int64(x.size))
    return y


def philox_scidvs_tau(seed, clip, npx):
    a = np.empty(npx, np.float32)
    lib().v2e_oracle_philox_scidvs_tau(C.c_uint64(seed), C.c_uint32(clip), C.c_int64(npx), _p(a))
    return a


def csdvs_update(p_plane, h_plane, alpha_p, alpha_h, num_steps, stop=1e-5):
    """EventEmulator._update_csdvs's stepping loop (emulator.py:1102-1124) on host planes; h_plane is updated in place.
    Returns (steps, last max_change)."""
    assert p_plane.dtype == h_plane.dtype and p_plane.shape == h_plane.shape and h_plane.flags.c_contiguous
    H, W = h_plane.shape
    last = C.c_double(0)
    L = lib()
    L.v2e_oracle_csdvs_update.restype = C.c_int
    steps = L.v2e_oracle_csdvs_update(_p(np.ascontiguousarray(p_plane)), _p(h_plane), C.c_int(H), C.c_int(W),
                                      C.c_int(1 if h_plane.dtype == np.float64 else 0), C.c_double(alpha_p), C.c_double(alpha_h),
                                      C.c_int(num_steps), C.c_double(stop), C.byref(last))
    return int(steps), float(last.value)


def perm_idx(seed, clip, frame, it, n):
    idx = np.empty(n, np.int64)
    lib().v2e_oracle_perm_idx(C.c_uint64(seed), C.c_uint32(clip), C.c_uint32(frame), C.c_uint32(it),
                              C.c_uint32(n), _p(idx))
    return idx


def perm_inv_idx(seed, clip, frame, it, n):
    """the same list as perm_idx, computed through v2e_perm_invert (the event writer's pull)"""
    idx = np.empty(n, np.int64)
    lib().v2e_oracle_perm_inv_idx(C.c_uint64(seed), C.c_uint32(clip), C.c_uint32(frame), C.c_uint32(it),
                                  C.c_uint32(n), _p(idx))
    return idx


def nth_set_bit_256(words, r):
    m = np.ascontiguousarray(words, dtype=np.uint64)
    L = lib()
    L.v2e_oracle_nth_set_bit_256.restype = C.c_uint32
    return int(L.v2e_oracle_nth_set_bit_256(_p(m), C.c_uint32(r)))


def ts_formula(t_prev, t_frame, n):
    ts = np.empty(n, np.float32)
    lib().v2e_oracle_ts(C.c_double(t_prev), C.c_double(t_frame), C.c_int32(n), _p(ts))
    return ts


# --------------------------------------------------------------- emulator
class OracleEmulator:
    """CPU restatement of v2ecore.emulator.EventEmulator (hot path only)."""

    def __init__(self, pos_thres=0.2, neg_thres=0.2, sigma_thres=0.03, cutoff_hz=0.0,
                 leak_rate_hz=0.1, refractory_period_s=0.0, shot_noise_rate_hz=0.0,
                 leak_jitter_fraction=0.1, noise_rate_cov_decades=0.1, seed=0,
                 rng_mode="tape", tape=None, shuffle=True, clip=0, hdr=False, photoreceptor_noise=False,
                 photoreceptor_noise_vrms=None, cs_lambda_pixels=None, cs_tau_p_ms=None, scidvs=False):
        # CSDVS (emulator.py:245-272, 707-716, 753-754, 1061-1124)
        self.cs_lambda_pixels, self.cs_tau_p_ms = cs_lambda_pixels, cs_tau_p_ms
        self.csdvs_enabled = cs_lambda_pixels is not None
        self.cs_tau_h_ms = 0 if (not self.csdvs_enabled or cs_tau_p_ms is None or cs_tau_p_ms == 0) else cs_tau_p_ms / (cs_lambda_pixels ** 2)
        self.cs_surround_frame = None
        self.cs_steps_taken = []
        # SCIDVS (emulator.py:56-80, 307-309, 480-483, 719-725, 747); float64 state only, like the HIP path
        self.scidvs = bool(scidvs)
        self.scidvs_highpass = self.scidvs_previous_photo = self.scidvs_tau_arr = None
        self.log_input = bool(hdr)  # emulator.py:304
        # emulator.py:192-205, 694-703; the vrms itself (emulator_utils.py:177-290, unseeded numpy draws in the
        # reference) is an input here
        self.photoreceptor_noise = bool(photoreceptor_noise)
        self.photoreceptor_noise_vrms = photoreceptor_noise_vrms
        self.photoreceptor_noise_arr = None
        self.pos_thres = pos_thres
        self.neg_thres = neg_thres
        self.pos_thres_nominal = pos_thres
        self.neg_thres_nominal = neg_thres
        self.sigma_thres = sigma_thres
        self.cutoff_hz = cutoff_hz
        self.leak_rate_hz = leak_rate_hz
        self.refractory_period_s = refractory_period_s
        self.shot_noise_rate_hz = shot_noise_rate_hz
        self.leak_jitter_fraction = leak_jitter_fraction
        self.noise_rate_cov_decades = noise_rate_cov_decades
        self.SHOT_NOISE_INTEN_FACTOR = 0.25
        self.seed = seed
        self.rng_mode = rng_mode
        self.shuffle = shuffle
        self.clip = clip
        if rng_mode == "tape":
            if tape is None:
                tape = TorchTape()
                if seed != 0:  # emulator.py:221-224
                    import torch, random
                    torch.manual_seed(seed)
                    np.random.seed(seed)
                    random.seed(seed)
        self.tape = tape
        self.t_previous = 0
        self.frame_counter = 0
        self.num_events_total = self.num_events_on = self.num_events_off = 0
        self.base_log_frame = None
        self.lp_log_frame = None
        self.timestamp_mem = None
        self.noise_rate_array = None
        self.last = {}

    def set_dvs_params(self, model):  # emulator.py:513-535
        if model == "clean":
            self.pos_thres = self.neg_thres = 0.2
            self.sigma_thres = 0.02
            self.cutoff_hz = 0
            self.leak_rate_hz = 0
            self.leak_jitter_fraction = 0
            self.noise_rate_cov_decades = 0
            self.shot_noise_rate_hz = 0
            self.refractory_period_s = 0
        elif model == "noisy":
            self.pos_thres = self.neg_thres = 0.2
            self.sigma_thres = 0.05
            self.cutoff_hz = 30
            self.leak_rate_hz = 0.1
            self.shot_noise_rate_hz = 5.0
            self.refractory_period_s = 0
            self.leak_jitter_fraction = 0.1
            self.noise_rate_cov_decades = 0.1

    def _params(self):
        P = EmuParams()
        P.f64_state = 1 if (self.cutoff_hz > 0 or self.log_input) else 0
        P.log_input = 1 if self.log_input else 0
        P.photoreceptor_noise = 1 if self.photoreceptor_noise else 0
        P.photoreceptor_noise_vrms = float(self.photoreceptor_noise_vrms or 0.0)
        P.scalar_thres = 0 if self.sigma_thres > 0 else 1
        P.rng_mode = RNG_PHILOX if self.rng_mode == "philox" else RNG_TAPE
        P.shuffle = 1 if self.shuffle else 0
        P.pos_thres_nominal = self.pos_thres_nominal
        P.neg_thres_nominal = self.neg_thres_nominal
        P.pos_thres_scalar = self._pos_scalar
        P.neg_thres_scalar = self._neg_scalar
        P.sigma_thres = self.sigma_thres
        P.cutoff_hz = self.cutoff_hz
        P.leak_rate_hz = self.leak_rate_hz
        P.leak_jitter_fraction = self.leak_jitter_fraction
        P.noise_rate_cov_decades = self.noise_rate_cov_decades
        P.refractory_period_s = self.refractory_period_s
        # with photoreceptor noise the shot-event generator and its base reset are skipped (emulator.py:893, 940)
        P.shot_noise_rate_hz = 0.0 if self.photoreceptor_noise else self.shot_noise_rate_hz
        P.shot_noise_inten_factor = self.SHOT_NOISE_INTEN_FACTOR
        if P.scalar_thres:
            import torch
            P.pos_pre_scalar = float(torch.div(self.pos_thres_nominal, self._pos_scalar))
            P.neg_pre_scalar = float(torch.div(self.neg_thres_nominal, self._neg_scalar))
        P.seed = self.seed
        return P

    def _update_csdvs(self, P, H, W, frame, t_prev, t_frame):
        """emulator.py:1061-1124: the host arithmetic as the reference writes it, the stepping loop in C (csdvs_update)."""
        delta_time = t_frame - t_prev
        abs_min_tau_p = 1e-9
        tau_p = abs_min_tau_p if (self.cs_tau_p_ms is None or self.cs_tau_p_ms == 0) else self.cs_tau_p_ms * 1e-3
        tau_h = abs_min_tau_p / (self.cs_lambda_pixels ** 2) if (self.cs_tau_h_ms is None or self.cs_tau_h_ms == 0) \
            else self.cs_tau_h_ms * 1e-3
        num_steps = int(np.ceil((delta_time / min(tau_p, tau_h)) * 5))
        actual_delta_time = delta_time / num_steps
        alpha_p, alpha_h = actual_delta_time / tau_p, actual_delta_time / tau_h
        if alpha_p >= 1 or alpha_h >= 1:
            raise SystemExit(1)
        lp_new = np.empty_like(self.lp_log_frame)
        rc = lib().v2e_oracle_lp_preview(C.byref(P), H, W, _p(frame), C.c_double(t_prev), C.c_double(t_frame),
                                         _p(self.lp_log_frame), _p(lp_new))
        assert rc == 0
        steps, _ = csdvs_update(lp_new, self.cs_surround_frame, alpha_p, alpha_h, num_steps)
        self.cs_steps_taken.append(steps)

    def generate_events(self, new_frame, t_frame):
        L = lib()
        frame = np.ascontiguousarray(np.asarray(new_frame), dtype=np.float64)
        H, W = frame.shape
        npx = H * W
        self.frame_counter += 1
        if t_frame < self.t_previous:
            raise ValueError("this frame time={} must be later than previous frame time={}".format(
                t_frame, self.t_previous))
        t_prev = float(self.t_previous)
        t_frame = float(t_frame)
        philox = self.rng_mode == "philox"
        if self.base_log_frame is None:
            self._pos_scalar = float(self.pos_thres)
            self._neg_scalar = float(self.neg_thres)
            P = self._params()
            sdt = np.float64 if P.f64_state else np.float32
            self.lp_log_frame = np.zeros((H, W), sdt)
            self.base_log_frame = np.zeros((H, W), sdt)
            self.timestamp_mem = np.zeros((H, W), np.float32)
            self.pos_thres_arr = np.zeros((H, W), np.float32)
            self.neg_thres_arr = np.zeros((H, W), np.float32)
            self.noise_rate_array = np.zeros((H, W), np.float32)
            tp = tn = nr = None
            if self.scidvs:
                self.scidvs_highpass = np.zeros((H, W), sdt)       # emulator.py:720: zeros_like(lp_log_frame)
                self.scidvs_previous_photo = np.zeros((H, W), sdt)  # taken from the first counted frame (:721)
                self._sc_first = True
                if philox:
                    self.scidvs_tau_arr = philox_scidvs_tau(self.seed, self.clip, npx).reshape(H, W)
            if not philox:
                if self.sigma_thres > 0:
                    tp = np.ascontiguousarray(self.tape.normal(self.pos_thres, self.sigma_thres, (H, W)))
                    tn = np.ascontiguousarray(self.tape.normal(self.neg_thres, self.sigma_thres, (H, W)))
                if self.scidvs:  # emulator.py:480-483: drawn between the thresholds and the noise rates
                    d = np.ascontiguousarray(self.tape.normal(0, 0.5, (H, W)))
                    self.scidvs_tau_arr = np.ascontiguousarray((np.float32(0.01) * self.tape.exp_scidvs(d)).astype(np.float32))
                if self.leak_rate_hz > 0:
                    r = np.ascontiguousarray(self.tape.randn((H, W)))
                    nr = np.ascontiguousarray(self.tape.exp_noise_rate(self.noise_rate_cov_decades, r))
            rc = L.v2e_oracle_init_state(C.byref(P), H, W, _p(frame), C.c_double(t_frame),
                                         C.c_uint32(self.clip), _p(tp), _p(tn), _p(nr),
                                         _p(self.lp_log_frame), _p(self.base_log_frame),
                                         _p(self.timestamp_mem), _p(self.pos_thres_arr),
                                         _p(self.neg_thres_arr), _p(self.noise_rate_array))
            assert rc == 0
            if self.csdvs_enabled:  # emulator.py:1062-1063, 715: surround = lp_log_frame, base = lp_log_frame - surround = 0
                self.cs_surround_frame = self.lp_log_frame.copy()
                self.base_log_frame[...] = 0
            return None  # t_previous intentionally NOT advanced (emulator.py:717)

        P = self._params()
        fidx = self.frame_counter - 1
        pos_cnt = np.zeros((H, W), np.int32)
        neg_cnt = np.zeros((H, W), np.int32)
        shot_on = np.zeros((H, W), np.uint8)
        shot_off = np.zeros((H, W), np.uint8)
        M = C.c_int32(0)
        pn_rand = None
        if self.photoreceptor_noise:
            if self.photoreceptor_noise_arr is None:
                self.photoreceptor_noise_arr = np.zeros((H, W), np.float64)  # emulator.py:684
            if not philox:
                pn_rand = np.ascontiguousarray(self.tape.randn((H, W)))  # emulator.py:698, before the leak draw
        leak = None
        if not philox and self.leak_rate_hz > 0:
            leak = np.ascontiguousarray(self.tape.randn((H, W)))
        if self.csdvs_enabled:
            self._update_csdvs(P, H, W, frame, t_prev, t_frame)
        rc = L.v2e_oracle_count(C.byref(P), H, W, _p(frame), C.c_double(t_prev), C.c_double(t_frame),
                                C.c_uint32(fidx), C.c_uint32(self.clip), _p(leak), None,
                                _p(self.lp_log_frame), _p(self.base_log_frame), _p(self.pos_thres_arr),
                                _p(self.neg_thres_arr), _p(self.noise_rate_array), _p(pos_cnt),
                                _p(neg_cnt), _p(shot_on), _p(shot_off), C.byref(M),
                                _p(self.photoreceptor_noise_arr) if self.photoreceptor_noise else None, _p(pn_rand),
                                _p(self.cs_surround_frame) if self.csdvs_enabled else None,
                                _p(self.scidvs_highpass), _p(self.scidvs_previous_photo), _p(self.scidvs_tau_arr),
                                C.c_int(1 if (self.scidvs and self._sc_first) else 0))
        if self.scidvs:
            self._sc_first = False
        assert rc == 0
        M = M.value
        self.last_M = M
        n = M if M > 0 else 1
        ts = None
        if not philox:
            ts_step = (t_frame - t_prev) / n
            ts = np.ascontiguousarray(self.tape.linspace(t_prev + ts_step, t_frame, n))
        cap = int(pos_cnt.sum() + neg_cnt.sum()) + 2 * npx + 1
        events = np.zeros((cap, 4), np.float32)
        itc = np.zeros(2 * (M + 1), np.uint32)
        rec = FrameRec()
        perms = []

        def emit(dry):
            rc = L.v2e_oracle_emit(C.byref(P), H, W, C.c_double(t_prev), C.c_double(t_frame),
                                   C.c_uint32(fidx), C.c_uint32(self.clip), _p(ts), n, _p(pos_cnt),
                                   _p(neg_cnt), _p(shot_on), _p(shot_off), C.c_int32(M),
                                   _p(self.lp_log_frame), _p(self.base_log_frame),
                                   _p(self.timestamp_mem), _p(self.pos_thres_arr),
                                   _p(self.neg_thres_arr), _p(events), C.c_uint64(cap), _p(itc),
                                   C.byref(rec), C.c_int(1 if dry else 0))
            assert rc == 0

        if not philox:
            emit(True)  # per-iteration totals -> randperm sizes, then rand for shot noise
            for i in range(M):
                n_i = int(itc[2 * i]) + int(itc[2 * i + 1])
                perms.append(self.tape.randperm(n_i, fidx, i) if n_i > 0 else None)
            if self.shot_noise_rate_hz > 0 and not self.photoreceptor_noise:
                u = np.ascontiguousarray(self.tape.rand((H, W)))
                rc = L.v2e_oracle_shot(C.byref(P), H, W, _p(frame), C.c_double(t_prev),
                                       C.c_double(t_frame), _p(u), _p(self.pos_thres_arr),
                                       _p(self.neg_thres_arr), _p(shot_on), _p(shot_off))
                assert rc == 0
        emit(False)
        ne = rec.n_events
        events = events[:ne]
        if not philox:
            row = 0
            for i in range(M):
                n_i = int(itc[2 * i]) + int(itc[2 * i + 1])
                if n_i > 0:
                    events[row:row + n_i] = events[row:row + n_i][perms[i]]  # emulator.py:869
                row += n_i
        self.num_events_total += ne
        self.num_events_on += rec.n_on
        self.num_events_off += rec.n_off
        self.last = dict(pos_cnt=pos_cnt, neg_cnt=neg_cnt, M=M, ts=ts, iter_counts=itc.copy(),
                         shot_on=shot_on, shot_off=shot_off)
        self.t_previous = t_frame
        return events if ne > 0 else None


# ----------------------------------------------------------------- SuperSloMo
UNET_LAYERS = (["conv1", "conv2"] + ["down%d.conv%d" % (d, c) for d in range(1, 6) for c in (1, 2)] +
               ["up%d.conv%d" % (u, c) for u in range(1, 6) for c in (1, 2)] + ["conv3"])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d_lrelu(x, w, b):
    """leaky_relu(conv2d(x, w, b, padding=(k-1)//2), 0.1); x [n,cin,h,w], w [cout,cin,k,k]."""
    x, w, b = _f32(x), _f32(w), _f32(b)
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    y = np.empty((n, cout, h, wd), np.float32)
    lib().v2e_oracle_conv2d_lrelu(_p(x), n, cin, h, wd, _p(w), _p(b), cout, k, _p(y))
    return y


def avgpool2(x):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, h // 2, w // 2), np.float32)
    lib().v2e_oracle_avgpool2(_p(x), n * c, h, w, _p(y))
    return y


def upsample2(x):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, 2 * h, 2 * w), np.float32)
    lib().v2e_oracle_upsample2(_p(x), n * c, h, w, _p(y))
    return y


def unet_forward(x, sd):
    """model.UNet.forward with state dict `sd` (numpy arrays, torch key names)."""
    x = _f32(x)
    n, cin, h, w = x.shape
    ws = [_f32(sd[k + ".weight"]) for k in UNET_LAYERS]
    bs = [_f32(sd[k + ".bias"]) for k in UNET_LAYERS]
    meta = np.array([[wt.shape[1], wt.shape[0], wt.shape[2]] for wt in ws], np.int32)
    cout = ws[-1].shape[0]
    wp = (C.c_void_p * 23)(*[wt.ctypes.data for wt in ws])
    bp = (C.c_void_p * 23)(*[bt.ctypes.data for bt in bs])
    y = np.empty((n, cout, h, w), np.float32)
    rc = lib().v2e_oracle_unet_forward(_p(x), n, cin, h, w, wp, bp, _p(meta), _p(y))
    assert rc == 0
    return y


def time_coefficients(ts):
    rows = []
    for t in ts:
        t = float(t)
        temp = -t * (1 - t)
        rows.append([temp, t * t, (1 - t) * (1 - t), temp, 1 - t, t])
    return np.asarray(rows, dtype=np.float64).astype(np.float32)


def slomo_prep(I0, I1, flow, ts):
    I0, I1, flow = _f32(I0), _f32(I1), _f32(flow)
    b, _, h, w = I0.shape
    coef = time_coefficients(ts)
    x12 = np.empty((len(ts) * b, 12, h, w), np.float32)
    lib().v2e_oracle_slomo_prep(_p(I0), _p(I1), _p(flow), _p(coef), len(ts), b, h, w, _p(x12))
    return x12


def slomo_fuse(I0, I1, x12, intrp, ts):
    I0, I1, x12, intrp = _f32(I0), _f32(I1), _f32(x12), _f32(intrp)
    b, _, h, w = I0.shape
    coef = time_coefficients(ts)
    out = np.empty((len(ts) * b, 1, h, w), np.float32)
    lib().v2e_oracle_slomo_fuse(_p(I0), _p(I1), _p(x12), _p(intrp), _p(coef), len(ts), b, h, w, _p(out))
    return out.reshape(len(ts), b, 1, h, w)


def slomo_interpolate(I0, I1, ts, flow_sd, interp_sd):
    """slomo.py:343, 404-433 on numpy arrays; returns dict(flow, x12, intrp, Ft)."""
    flow = unet_forward(np.concatenate((I0, I1), axis=1), flow_sd)
    x12 = slomo_prep(I0, I1, flow, ts)
    intrp = unet_forward(x12, interp_sd)
    Ft = slomo_fuse(I0, I1, x12, intrp, ts)
    return dict(flow=flow, x12=x12, intrp=intrp, Ft=Ft)
