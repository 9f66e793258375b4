"""Drop-in `EventEmulator` running the DVS pixel model on MI355X HIP kernels.

Mirrors the public surface of the reference class (v2ecore/emulator.py:35-1059):
same constructor signature (:86-117), `generate_events(new_frame, t_frame)` (:619),
`set_dvs_params` (:513), `reset` (:558), `prepare_storage` (:374), `cleanup` (:402),
the counters `num_events_total/on/off`, `t_previous`, and the class attributes that
v2ecore/v2e_args.py:209,216 reads.

Random numbers come in two modes, selected with the extra keyword
`rng_mode` (or env V2E_AMD_RNG):
  "tape"   (default) the host issues exactly the reference's torch calls in the
           reference's order from torch's global CPU generator -- normal, normal, randn
           on the first frame; randn, one randperm per non-empty iteration, rand on every
           later frame, plus torch.linspace for the timestamps -- and hands the values to
           the kernels.  Event-for-event identical to the reference torch-CPU path under
           the same seed.
  "philox" counter-based Philox4x32-10 inside the kernels, timestamps and the
           per-iteration shuffle computed on device; no host round trip except the final
           copy of the event list.  `generate_events_batch` keeps a whole clip on device.

`show_dvs_model_state` / `save_dvs_model_state` (emulator.py:580-617, 756-764) and `record_single_pixel_states`
(emulator.py:279-300, 985-1009) are host pass-throughs: the named state planes are read back from the device after every frame
(three of them -- log_new_frame, c_minus_s_frame, diff_frame -- are left in extra planes by k_count while one of the two options
is on, v2e_emu_set_model_state_planes) and displayed / saved through cv2 where it is importable, resp. recorded and pickled
as the reference does; both put generate_events on the step-wise kernels.  `hdr=True`
(log-encoded input, emulator.py:304, 666), `photoreceptor_noise=True` (emulator.py:694-703), `scidvs=True` (float64 and float32
state: torch's float32 sinh is restated bit for bit in include/v2e_detmath.h)
and the centre-surround pixel (`cs_lambda_pixels`, emulator.py:1061-1124: the diffuser is stepped on the device between
frames, csrc/csdvs.hip; in generate_events_batch too: the stepping loop is enqueued whole with its stop rule on the device)
are supported.
"""
import atexit
import logging
import math
import os
import random
from typing import Optional

import weakref

import numpy as np
import torch

from . import _capi
from ._capi import EmuParams, RNG_PHILOX, RNG_TAPE
from .engine import EmuEngine

logger = logging.getLogger(__name__)


class _TorchTape:
    """The reference's own random draws, issued on the host (tape mode)."""

    def normal(self, mean, std, shape):
        return torch.normal(mean, std, size=shape, dtype=torch.float32)  # emulator.py:460-471

    def randn(self, shape):
        return torch.randn(shape, dtype=torch.float32)  # emulator.py:501, emulator_utils.py:122

    def rand(self, shape):
        return torch.rand(size=shape, dtype=torch.float32)  # emulator_utils.py:338

    def randperm(self, n, frame=None, it=None):
        return torch.randperm(n)  # emulator.py:868

    def linspace(self, start, end, n):
        return torch.linspace(start=start, end=end, steps=n, dtype=torch.float32)  # emulator.py:793

    def exp_noise_rate(self, cov, randn):
        return torch.exp(math.log(10) * cov * randn)  # emulator.py:504-505

    def exp_scidvs(self, draw):
        return torch.exp(draw)  # emulator.py:481-483


def photoreceptor_noise_vrms(shot_noise_rate_hz, f3db, sample_rate_hz, pos_thr, neg_thr, sigma_thr):
    """Gaussian RMS amplitude (ln units) to inject before the photoreceptor low-pass so that threshold crossings of
    the filtered noise occur at the requested event rate: the curve fit of Graca & Delbruck (2021) evaluated over
    sampled thresholds, then scaled by the noise-equivalent-bandwidth factor of this IIR measured on a white
    sequence (what emulator_utils.py:177-290 computes; like there, the draws are unseeded numpy ones)."""
    rate_per_bw = (shot_noise_rate_hz / f3db) / 2  # the fit is for ON events only
    x = math.log10(rate_per_bw)
    rng = np.random.default_rng()
    n = 300
    thr = np.minimum(pos_thr + sigma_thr * rng.standard_normal(n), neg_thr + sigma_thr * np.random.default_rng().standard_normal(n))
    y = -0.0026 * x ** 3 - 0.036 * x ** 2 - 0.1949 * x + 0.321  # log10(thr / vn)
    vn = float(np.mean(thr / 10 ** y))
    tau = 1 / (f3db * 2 * math.pi)
    dt = 1 / sample_rate_hz
    rin = vn * np.random.default_rng().standard_normal(np.arange(0, 1000 * tau, dt).shape)
    eps = dt / tau
    rout = np.zeros_like(rin)
    for i in range(1, len(rin)):
        rout[i] = rout[i - 1] * (1 - eps) + rin[i] * eps
    return float(np.std(rin) / np.std(rout) * vn)


_THREAD_OVERRIDE = {"users": 0, "prev": None}  # tape-mode emulators alive that rely on the one-thread setting


def _release_host_threads():
    """cleanup() of a tape-mode emulator: the previous thread count comes back when the LAST one that relies on it is gone."""
    o = _THREAD_OVERRIDE
    if o["users"] > 0:
        o["users"] -= 1
        if o["users"] == 0 and o["prev"] is not None:
            torch.set_num_threads(o["prev"])
            o["prev"] = None


def _limit_host_threads(n):
    """Tape mode issues a handful of small torch CPU ops per frame (randn / rand of one frame, a randperm per
    iteration).  Their values do not depend on the intra-op thread count, but on a many-core host fanning each of
    them out over the OpenMP pool costs milliseconds: 250 -> 1 170 frames/s on the MI355X box with one thread.
    The setting is process-global in torch, so it is applied once, when a tape-mode emulator is constructed
    (`tape_host_threads`, default 1; None leaves torch alone), said so in the log, and put back by cleanup().
    Returns the previous value (None if nothing was changed)."""
    if n is None:
        return None
    o = _THREAD_OVERRIDE
    if o["users"] > 0 or torch.get_num_threads() == int(n):  # already in force (or nothing to change): just count this user
        o["users"] += 1
        return True
    prev = torch.get_num_threads()
    o["users"], o["prev"] = 1, prev
    logger.warning("v2e_amd.EventEmulator (tape mode): torch.set_num_threads(%d) for this process (was %d; restored by "
                   "cleanup(); pass tape_host_threads=None to leave torch alone)", int(n), prev)
    torch.set_num_threads(int(n))
    return prev


def _as_f32_tensor(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _cleanup_at_exit(ref):
    emu = ref()
    if emu is not None:
        emu.cleanup()


class _PendingRun:
    """Handle of an enqueued device-resident run (EventEmulator.generate_events_batch_async)."""

    def __init__(self, emu, ev, recs, done, counts, start, return_device, empty, dts=None, ticket=None):
        self.emu, self.ev, self.recs, self._done = emu, ev, recs, done
        self.counts, self.start, self.return_device, self.empty, self.dts = counts, start, return_device, empty, dts
        self.ticket = ticket  # overlapped runs (v2e_emu_run | 1024): what result() waits for on the host (v2e_emu_run_wait)
        self._res = None

    @property
    def done(self):
        """A torch event behind which the run's rows and records are final (for callers that order their own streams behind it)."""
        if self._done is None and self.recs is not None:
            self._done = self.emu._engine.run_done_event()
        return self._done

    def result(self):
        if self._res is None:
            if self.recs is None:
                self._res = ((self.empty if self.return_device else None), self.counts)
            else:
                self._res = self.emu._finish_run(self)
        return self._res


class EventEmulator(object):
    """compute events based on the input frame (MI355X implementation)."""

    # class attributes read by v2ecore/v2e_args.py (emulator.py:41-56)
    l255 = np.log(255)
    gr = (0, 255)
    lg = (0, l255)
    slg = (-l255 / 8, l255 / 8)
    MODEL_STATES = {'new_frame': gr, 'log_new_frame': lg, 'lp_log_frame': lg, 'scidvs_highpass': slg,
                    'photoreceptor_noise_arr': slg, 'cs_surround_frame': lg, 'c_minus_s_frame': slg,
                    'base_log_frame': slg, 'diff_frame': slg}
    MAX_CHANGE_TO_TERMINATE_EULER_SURROUND_STEPPING = 1e-5
    SINGLE_PIXEL_STATES_FILENAME = 'pixel-states.dat'
    SINGLE_PIXEL_MAX_SAMPLES = 10000
    SCIDVS_GAIN = 2
    SCIDVS_TAU_S = .01
    SCIDVS_TAU_COV = 0.5

    def __init__(
            self,
            pos_thres: float = 0.2,
            neg_thres: float = 0.2,
            sigma_thres: float = 0.03,
            cutoff_hz: float = 0.0,
            leak_rate_hz: float = 0.1,
            refractory_period_s: float = 0.0,
            shot_noise_rate_hz: float = 0.0,
            photoreceptor_noise: bool = False,
            leak_jitter_fraction: float = 0.1,
            noise_rate_cov_decades: float = 0.1,
            seed: int = 0,
            output_folder: str = None,
            dvs_h5: str = None,
            dvs_aedat2: str = None,
            dvs_aedat4: str = None,
            dvs_text: str = None,
            show_dvs_model_state: str = None,
            save_dvs_model_state: bool = False,
            output_width: int = None,
            output_height: int = None,
            device: str = "cuda",
            cs_lambda_pixels: float = None,
            cs_tau_p_ms: float = None,
            hdr: bool = False,
            scidvs: bool = False,
            record_single_pixel_states=None,
            label_signal_noise=False,
            *,
            rng_mode: Optional[str] = None,
            shuffle: bool = True,
            tape=None,
            max_iters: int = 64,
            photoreceptor_noise_vrms: Optional[float] = None,
            tape_host_threads: Optional[int] = 1,
    ):
        # research tooling (host pass-throughs, see the module docstring)
        self.dont_show_list = []
        self.show_list = []
        self.video_writers = {}
        self._state_planes = None
        self._cv2 = None
        self._show_warned = False
        if show_dvs_model_state is not None and len(show_dvs_model_state) == 1 and show_dvs_model_state[0] == 'all':
            logger.info(f'will show all model states that exist from {EventEmulator.MODEL_STATES.keys()}')  # emulator.py:365-368
            show_dvs_model_state = list(EventEmulator.MODEL_STATES.keys())
        self.record_single_pixel_states = record_single_pixel_states  # emulator.py:279-300
        self.single_pixel_sample_count = 0
        if self.record_single_pixel_states is None:
            self.single_pixel_states = None
        else:
            if not (type(self.record_single_pixel_states) is tuple):
                raise ValueError(f'--record_single_pixel_states {self.record_single_pixel_states} should be a tuple, e.g. (10,20)')
            if len(self.record_single_pixel_states) != 2:
                raise ValueError(f'--record_single_pixel_states {self.record_single_pixel_states} should have two pixel addresses (x,y)')
            for i in self.record_single_pixel_states:
                if not (type(i) is int):
                    raise ValueError(f'--record_single_pixel_states {self.record_single_pixel_states} should have two '
                                     f'integer-value pixel addresses (x,y)')
            self.single_pixel_states = {k: np.full(self.SINGLE_PIXEL_MAX_SAMPLES, np.nan) for k in (
                'time', 'new_frame', 'base_log_frame', 'lp_log_frame', 'log_new_frame', 'pos_thres', 'neg_thres', 'diff_frame',
                'final_neg_evts_frame', 'final_pos_evts_frame')}

        self.no_events_warning_count = 0
        logger.info("ON/OFF log_e temporal contrast thresholds: {} / {} +/- {}".format(
            pos_thres, neg_thres, sigma_thres))
        self.reset()
        self.t_previous = 0
        self.device = device

        self.sigma_thres = sigma_thres
        self.pos_thres = pos_thres
        self.neg_thres = neg_thres
        self.pos_thres_nominal = pos_thres
        self.neg_thres_nominal = neg_thres
        self.cutoff_hz = cutoff_hz
        self.leak_rate_hz = leak_rate_hz
        self.refractory_period_s = refractory_period_s
        self.shot_noise_rate_hz = shot_noise_rate_hz
        self.photoreceptor_noise = bool(photoreceptor_noise)
        self.photoreceptor_noise_vrms = photoreceptor_noise_vrms  # None: computed like emulator_utils.py:177-290
        self._pn_last_rate = None
        self.photoreceptor_noise_arr = None
        if self.photoreceptor_noise and (shot_noise_rate_hz == 0 or cutoff_hz == 0):  # emulator.py:196-204 (v2e_quit)
            logger.error('--photoreceptor_noise needs a finite --shot_noise_rate_hz and --cutoff_hz')
            raise SystemExit(1)
        self.leak_jitter_fraction = leak_jitter_fraction
        self.noise_rate_cov_decades = noise_rate_cov_decades
        self.SHOT_NOISE_INTEN_FACTOR = 0.25
        self.output_folder = output_folder
        self.output_width = output_width
        self.output_height = output_height
        self.show_dvs_model_state = show_dvs_model_state
        self.save_dvs_model_state = save_dvs_model_state
        self.label_signal_noise = label_signal_noise
        self.log_input = bool(hdr)  # emulator.py:304: frames are log-encoded already
        self.scidvs = bool(scidvs)  # emulator.py:307-309: nonlinear CR high-pass amplified log intensity
        self.scidvs_highpass = self.scidvs_previous_photo = self.scidvs_tau_arr = None
        # CSDVS (emulator.py:245-272)
        self.cs_steps_warning_printed = False
        self.cs_steps_taken = []
        self.cs_alpha_warning_printed = False
        self.cs_tau_p_ms = cs_tau_p_ms
        self.cs_lambda_pixels = cs_lambda_pixels
        self.cs_surround_frame = None
        self.csdvs_enabled = cs_lambda_pixels is not None
        if self.csdvs_enabled:
            self.cs_tau_h_ms = 0 if (cs_tau_p_ms is None or cs_tau_p_ms == 0) else cs_tau_p_ms / (cs_lambda_pixels ** 2)
            logger.info(f'Center-surround parameters:\n\tcs_tau_p_ms: {self.cs_tau_p_ms}\n\tcs_tau_h_ms:  {self.cs_tau_h_ms}\n\t'
                        f'cs_lambda_pixels:  {self.cs_lambda_pixels:.2f}\n\t')
        self.seed = seed

        self.rng_mode = (rng_mode or os.environ.get("V2E_AMD_RNG", "tape")).lower()
        if self.rng_mode not in ("tape", "philox"):
            raise ValueError("rng_mode must be 'tape' or 'philox', got %r" % (self.rng_mode,))
        self.shuffle = bool(shuffle)
        self._tape = tape if tape is not None else _TorchTape()
        self._host_threads_prev = None
        if self.rng_mode == "tape":
            self._host_threads_prev = _limit_host_threads(tape_host_threads)
        self._max_iters = max_iters
        if seed != 0:  # emulator.py:221-224
            torch.manual_seed(seed)
            np.random.seed(seed)
            random.seed(seed)

        # event file writers are outside the hot path: pass through to the reference's own
        # writer classes when they are importable (emulator.py:312-346)
        self.dvs_h5 = self.dvs_aedat2 = self.dvs_aedat4 = self.dvs_text = None
        self.dvs_h5_dataset = self.frame_h5_dataset = self.frame_ts_dataset = self.frame_ev_idx_dataset = None
        self._open_writers(dvs_h5, dvs_aedat2, dvs_aedat4, dvs_text)

        self._engine: Optional[EmuEngine] = None
        self._thres_scalar = None
        # emulator.py:296 registers cleanup with atexit.  Through a weak reference here: a bound method would keep every emulator -- and
        # the gigabytes of device scratch a 1280x720 one holds -- alive until the process ends; one that is dropped is cleaned up then
        # (__del__), one that is still alive at exit by the hook
        atexit.register(_cleanup_at_exit, weakref.ref(self))

    # ------------------------------------------------------------- plumbing
    def _open_writers(self, dvs_h5, dvs_aedat2, dvs_aedat4, dvs_text):
        """Event file sinks (emulator.py:312-346).  AEDAT-2.0 and text: v2e_amd.sinks writers fed from the device-resident
        event buffer (integer conversion on the GPU, byte-identical to the reference writers).  HDF5: the reference's own
        layout through h5py (rows converted on the GPU).  AEDAT-4: v2e_amd.sinks.HostAEDat4Output over the third-party
        dv_processing package (as the reference's writer is), or any writer object the caller passes as `dvs_aedat4`."""
        if not (dvs_h5 or dvs_aedat2 or dvs_aedat4 or dvs_text):
            return

        def suffixed(name, suffix):  # v2e_utils.checkAddSuffix
            return name if name.endswith(suffix) else name + suffix

        folder = self.output_folder or ""
        if dvs_h5:
            try:
                import h5py
            except ImportError as e:
                raise NotImplementedError("dvs_h5 needs h5py: %s" % e)
            self.dvs_h5 = h5py.File(suffixed(os.path.join(folder, dvs_h5), '.h5'), "w")
            self.dvs_h5_dataset = self.dvs_h5.create_dataset(
                name="events", shape=(0, 4), maxshape=(None, 4), dtype="uint32", compression="gzip")
        if dvs_aedat2:
            from .sinks import DeviceAEDat2Output
            self.dvs_aedat2 = DeviceAEDat2Output(suffixed(os.path.join(folder, dvs_aedat2), '.aedat'),
                                                 output_width=self.output_width, output_height=self.output_height,
                                                 label_signal_noise=self.label_signal_noise)
        if dvs_text:
            from .sinks import DeviceTextOutput
            self.dvs_text = DeviceTextOutput(suffixed(os.path.join(folder, dvs_text), '.txt'),
                                             label_signal_noise=self.label_signal_noise)
        if dvs_aedat4:
            if hasattr(dvs_aedat4, "appendEvents"):  # a caller-supplied writer object
                self.dvs_aedat4 = dvs_aedat4
            else:
                from .sinks import HostAEDat4Output  # over dv_processing; NotImplementedError where that is absent
                self.dvs_aedat4 = HostAEDat4Output(suffixed(os.path.join(folder, dvs_aedat4), '.aedat4'))

    def prepare_storage(self, n_frames, frame_ts):  # emulator.py:374-400
        if self.dvs_h5:
            self.frame_h5_dataset = self.dvs_h5.create_dataset(
                name="frame", shape=(n_frames, self.output_height, self.output_width), dtype="uint8",
                compression="gzip")
            frame_ts_arr = np.array(frame_ts, dtype=np.float32) * 1e6
            self.frame_ts_dataset = self.dvs_h5.create_dataset(
                name="frame_ts", shape=(n_frames,), data=frame_ts_arr.astype(np.uint32), dtype="uint32",
                compression="gzip")
            self.frame_ev_idx_dataset = self.dvs_h5.create_dataset(
                name="frame_idx", shape=(n_frames,), dtype="uint64", compression="gzip")
        else:
            self.frame_h5_dataset = self.frame_ts_dataset = self.frame_ev_idx_dataset = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    def cleanup(self):  # emulator.py:402-429
        if len(getattr(self, "cs_steps_taken", ())) > 1:
            logger.info(f'CSDVS steps statistics: mean+std= {np.mean(self.cs_steps_taken):.0f} + {np.std(self.cs_steps_taken):.0f} '
                        f'(median= {np.median(self.cs_steps_taken):.0f})')
        if getattr(self, "_host_threads_prev", None) is not None:
            _release_host_threads()
            self._host_threads_prev = None
        for name, vw in list(getattr(self, "video_writers", {}).items()):  # emulator.py:424-426
            logger.info(f'closing video AVI {name}')
            try:
                vw.release()
            except Exception:
                pass
        if getattr(self, "video_writers", None):
            self.video_writers = {}
        if getattr(self, "record_single_pixel_states", None) is not None:  # emulator.py:428-429
            self.save_recorded_single_pixel_states()
            self.record_single_pixel_states = None
        for w in ("dvs_h5", "dvs_aedat2", "dvs_aedat4", "dvs_text"):
            o = getattr(self, w, None)
            if o is not None:
                try:
                    o.close()
                except Exception:
                    pass
                setattr(self, w, None)

    def set_dvs_params(self, model: str):  # emulator.py:513-556
        if model == 'clean':
            self.pos_thres = 0.2
            self.neg_thres = 0.2
            self.sigma_thres = 0.02
            self.cutoff_hz = 0
            self.leak_rate_hz = 0
            self.leak_jitter_fraction = 0
            self.noise_rate_cov_decades = 0
            self.shot_noise_rate_hz = 0
            self.refractory_period_s = 0
        elif model == 'noisy':
            self.pos_thres = 0.2
            self.neg_thres = 0.2
            self.sigma_thres = 0.05
            self.cutoff_hz = 30
            self.leak_rate_hz = 0.1
            self.shot_noise_rate_hz = 5.0
            self.refractory_period_s = 0
            self.leak_jitter_fraction = 0.1
            self.noise_rate_cov_decades = 0.1
        else:
            logger.warning("dvs_params {} not known: Using commandline assigned options".format(model))

    def reset(self):  # emulator.py:558-578
        self._failed = None
        self._async_slots = {}  # handles of runs enqueued before the reset are no longer collected on the caller's behalf
        self.num_events_total = 0
        self.num_events_on = 0
        self.num_events_off = 0
        self.new_frame = None
        self.log_new_frame = None
        self.lp_log_frame = None
        self.base_log_frame = None
        self.diff_frame = None
        self.frame_counter = 0
        self.timestamp_mem = None
        self.noise_rate_array = None
        self.photoreceptor_noise_arr = None
        self.scidvs_highpass = self.scidvs_previous_photo = self.scidvs_tau_arr = None
        self.cs_surround_frame = None
        self._pn_last_rate = None  # the noise amplitude is recomputed for the first frame pair of the next clip
        self._initialized = False

    # ------------------------------------------------------------- parameters
    def _params(self) -> EmuParams:
        """Snapshot the (mutable) attributes into the C struct, every call."""
        P = EmuParams()
        P.f64_state = 1 if (self.cutoff_hz > 0 or self.log_input) else 0
        P.log_input = 1 if self.log_input else 0
        P.scalar_thres = 1 if self._thres_scalar is not None and self._thres_is_scalar else 0
        P.rng_mode = RNG_PHILOX if self.rng_mode == "philox" else RNG_TAPE
        P.shuffle = 1 if self.shuffle else 0
        P.pos_thres_nominal = float(self.pos_thres_nominal)
        P.neg_thres_nominal = float(self.neg_thres_nominal)
        P.pos_thres_scalar, P.neg_thres_scalar = self._thres_scalar
        P.sigma_thres = float(self.sigma_thres)
        P.cutoff_hz = float(self.cutoff_hz)
        P.leak_rate_hz = float(self.leak_rate_hz)
        P.leak_jitter_fraction = float(self.leak_jitter_fraction)
        P.noise_rate_cov_decades = float(self.noise_rate_cov_decades)
        P.refractory_period_s = float(self.refractory_period_s)
        # photoreceptor noise replaces the shot-event generator and its base reset (emulator.py:893, 940)
        P.shot_noise_rate_hz = 0.0 if self.photoreceptor_noise else float(self.shot_noise_rate_hz)
        P.photoreceptor_noise = 1 if self.photoreceptor_noise else 0
        P.photoreceptor_noise_vrms = float(getattr(self, "_pn_vrms", 0.0) or 0.0)
        P.shot_noise_inten_factor = float(self.SHOT_NOISE_INTEN_FACTOR)
        if P.scalar_thres:  # emulator.py:475-478 with Python-float thresholds
            P.pos_pre_scalar = float(torch.div(self.pos_thres_nominal, P.pos_thres_scalar))
            P.neg_pre_scalar = float(torch.div(self.neg_thres_nominal, P.neg_thres_scalar))
        P.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        return P

    def _ensure_engine(self, H, W):
        if self._engine is not None and (self._engine.H, self._engine.W) != (H, W):
            if self._initialized:
                raise ValueError("frame shape changed from %dx%d to %dx%d after the first frame; call reset() first" % (
                    self._engine.H, self._engine.W, H, W))
            self._engine = None
        if self._engine is None:
            self._engine = EmuEngine(H, W, n_clips=1, device=self.device, max_iters=self._max_iters)
        return self._engine

    # ------------------------------------------------------------- first frame
    def _first_frame(self, frame_dev, H, W, t_frame):
        """emulator.py:681-717 + _init :439-511."""
        eng = self._ensure_engine(H, W)
        # thresholds are scalars (Python floats) until _init draws them; remember the
        # values current at this moment (set_dvs_params may have changed them)
        if torch.is_tensor(self.pos_thres) or torch.is_tensor(self.neg_thres):
            # re-initialisation after reset(): the per-pixel planes of the previous run are dropped and
            # new ones are drawn around the same scalar thresholds (the reference's reset() leaves the old
            # tensors in place, which its own _init cannot consume)
            prev = self._thres_scalar or (float(self.pos_thres_nominal), float(self.neg_thres_nominal))
            self.pos_thres, self.neg_thres = prev
        self._thres_scalar = (float(self.pos_thres), float(self.neg_thres))
        self._thres_is_scalar = not (self.sigma_thres > 0)
        eng.alloc_state(self.cutoff_hz > 0 or self.log_input)
        P = self._params()
        tp = tn = nr = None
        sc_tau_host = None
        if self.scidvs:  # two more state planes + per-pixel time constants (emulator.py:480-483, 719-722)
            sd = torch.float64 if (self.cutoff_hz > 0 or self.log_input) else torch.float32  # zeros_like(lp_log_frame)
            self._sc_planes = [torch.zeros((1, eng.npx_pad), dtype=sd, device=eng.device) for _ in range(2)]
            self._sc_tau = torch.zeros((1, eng.npx_pad), dtype=torch.float32, device=eng.device)
            _capi.check(eng.lib.v2e_emu_set_scidvs(eng._h, self._sc_planes[0].data_ptr(), self._sc_planes[1].data_ptr(),
                                                   self._sc_tau.data_ptr(), int(self.frame_counter)), "v2e_emu_set_scidvs")
        if self.rng_mode == "tape":
            dev = eng.device
            if self.sigma_thres > 0:
                tp = _as_f32_tensor(self._tape.normal(self.pos_thres, self.sigma_thres, (H, W))).to(dev)
                tn = _as_f32_tensor(self._tape.normal(self.neg_thres, self.sigma_thres, (H, W))).to(dev)
            if self.scidvs and self.SCIDVS_TAU_COV > 0:  # drawn between the thresholds and the noise rates
                d = _as_f32_tensor(self._tape.normal(0, self.SCIDVS_TAU_COV, (H, W)))
                sc_tau_host = self.SCIDVS_TAU_S * _as_f32_tensor(self._tape.exp_scidvs(d))
            if self.leak_rate_hz > 0:
                r = _as_f32_tensor(self._tape.randn((H, W)))
                nr = _as_f32_tensor(self._tape.exp_noise_rate(self.noise_rate_cov_decades, r)).to(dev)
        eng.init_state(P, frame_dev, t_frame, tp, tn, nr)
        if self.scidvs:
            if sc_tau_host is not None:
                self._sc_tau[0, :H * W] = sc_tau_host.reshape(-1).to(eng.device)
            self.scidvs_highpass = eng.plane(self._sc_planes[0])
            self.scidvs_previous_photo = eng.plane(self._sc_planes[1])
            self.scidvs_tau_arr = eng.plane(self._sc_tau)
        if self.csdvs_enabled:
            # emulator.py:1062-1063 and :715: the surround starts as a copy of lp_log_frame, so base_log_frame =
            # lp_log_frame - cs_surround_frame is zero; [surround, ping-pong scratch, the coming frame's lp_log_frame]
            self._cs_planes = [eng.lp.clone(), torch.empty_like(eng.lp), torch.empty_like(eng.lp)]
            eng.base.zero_()
            eng.set_csdvs(self._cs_planes[0])
            self.cs_surround_frame = eng.plane(self._cs_planes[0])
        else:
            eng.set_csdvs(None)
        if self._wants_states():
            self._state_planes = [torch.zeros((1, eng.npx_pad), dtype=torch.float64, device=eng.device) for _ in range(3)]
            eng.set_model_state_planes(self._state_planes)
        else:
            self._state_planes = None
            eng.set_model_state_planes(None)
        # public state attributes, as [H,W] device views
        self.lp_log_frame = eng.plane(eng.lp)
        self.base_log_frame = eng.plane(eng.base)
        if self.sigma_thres > 0:
            self.pos_thres = eng.plane(eng.pos_thres)
            self.neg_thres = eng.plane(eng.neg_thres)
        if self.leak_rate_hz > 0:
            self.noise_rate_array = eng.plane(eng.noise_rate)
        if self.refractory_period_s > 0:
            self.timestamp_mem = eng.plane(eng.ts_mem)
        self._initialized = True

    # ------------------------------------------------------------- hot path
    def generate_events(self, new_frame, t_frame):
        """Compute events in new frame (emulator.py:619-1022).

        new_frame: np.ndarray or torch tensor [height, width]; t_frame: seconds.
        Returns np.ndarray [N,4] float32 rows [t, x, y, p(+1/-1)] or None.
        """
        return self._generate_events(new_frame, t_frame)

    def _generate_events(self, new_frame, t_frame):
        if getattr(self, "_failed", None):
            raise _capi.V2EAmdError("a previous device-resident run failed (%s); call reset()" % self._failed)
        if self.frame_h5_dataset is not None:
            fr = new_frame.detach().cpu().numpy() if torch.is_tensor(new_frame) else np.asarray(new_frame)
            self.frame_h5_dataset[self.frame_counter] = fr.astype(np.uint8)
        self.frame_counter += 1
        if t_frame < self.t_previous:
            raise ValueError("this frame time={} must be later than previous frame time={}".format(
                t_frame, self.t_previous))
        shape = tuple(new_frame.shape)
        if len(shape) != 2:
            raise ValueError("new_frame must be [height, width], got shape %r" % (shape,))
        H, W = shape
        eng = self._ensure_engine(H, W)
        # Philox mode after the first frame: the whole frame is ONE C call (v2e_emu_frame), the host frame goes through
        # the handle's pinned staging instead of a torch tensor
        fast = self._initialized and self.rng_mode == "philox" and not self.photoreceptor_noise and not self.csdvs_enabled \
            and not self._wants_states()
        host_frame = None
        if fast and isinstance(new_frame, np.ndarray) and new_frame.dtype in (np.uint8, np.float32, np.float64):
            host_frame = np.ascontiguousarray(new_frame)
        frame_dev = None if host_frame is not None else eng.to_device_frame(new_frame)

        if not self._initialized:
            self._first_frame(frame_dev, H, W, float(t_frame))
            return None  # t_previous intentionally not advanced (emulator.py:717)

        t_prev = float(self.t_previous)
        t_frame = float(t_frame)
        fidx = self.frame_counter - 1
        P = self._params()
        if bool(P.f64_state) != eng.f64_state:
            raise ValueError("cutoff_hz changed sign after the first frame: the state dtype "
                             "(float64 iff cutoff_hz > 0 or hdr) is fixed by the first frame")
        tape = self.rng_mode == "tape"
        dev = eng.device

        counted = False
        if fast:
            ev = eng.event_buffer(max(4 * H * W, 1 << 16))
            rc, out8, events = eng.frame(P, host_frame if host_frame is not None else frame_dev, t_prev, t_frame, fidx, ev)
            n_events, n_on, n_off, n_signal, M = (int(out8[k]) for k in range(5))
            if rc == 0:
                if M > 100:
                    logger.warning(f'Too many events generated for this frame: num_iter={M}>100 events')
                if M == 0 and self.no_events_warning_count < 100:
                    logger.warning(f'no signal events generated for frame #{self.frame_counter:,} at t={t_frame:.4f}s')
                    self.no_events_warning_count += 1
                self.num_events_on += n_on
                self.num_events_off += n_off
                self.num_events_total += n_events
                if events is not None:
                    self._write_events(events, n_signal, events_dev=ev[0, :n_events])
                if self.frame_ev_idx_dataset is not None:
                    self.frame_ev_idx_dataset[self.frame_counter - 1] = self.dvs_h5_dataset.shape[0]
                self.t_previous = t_frame
                return events
            # rc 1: more iterations than the scratch holds; rc 2: more rows than the buffer: the frame is counted, the general
            # path below grows what is needed and finishes it
            counted = True
            if frame_dev is None:
                frame_dev = eng.to_device_frame(new_frame)

        pn_draw = None
        if self.photoreceptor_noise:  # emulator.py:694-703
            if self.photoreceptor_noise_arr is None:  # zeros_like(lp_log_frame) on the first frame (emulator.py:684)
                self._pn_plane = torch.zeros((1, eng.npx_pad), dtype=torch.float64, device=dev)
                self.photoreceptor_noise_arr = eng.plane(self._pn_plane)
            rate = 1.0 / (t_frame - t_prev)
            if self.photoreceptor_noise_vrms is not None:
                self._pn_vrms = float(self.photoreceptor_noise_vrms)
            elif self._pn_last_rate is None or abs(rate / self._pn_last_rate - 1) >= 0.1:  # emulator_utils.py:217-220
                self._pn_vrms = photoreceptor_noise_vrms(self.shot_noise_rate_hz, self.cutoff_hz, rate, self.pos_thres_nominal,
                                                         self.neg_thres_nominal, self.sigma_thres)
                self._pn_last_rate = rate
            P = self._params()
            if tape:
                pn_draw = _as_f32_tensor(self._tape.randn((H, W))).to(dev)  # before the leak draw, as in the reference
            eng.set_pnoise(self._pn_plane, pn_draw)
        leak = None
        if tape and self.leak_rate_hz > 0:
            leak = _as_f32_tensor(self._tape.randn((H, W))).to(dev)
        if self.csdvs_enabled:  # emulator.py:707-708: between the low-pass and the event computation
            self._update_csdvs(P, frame_dev, t_prev, t_frame, fidx)
        if not counted:
            eng.count(P, frame_dev, [t_prev], [t_frame], fidx, leak_randn=leak)
        if self._state_planes is not None:
            self._after_count_states(new_frame, t_frame)
        rec = eng.read_rec(fidx)[0]
        M = int(rec.max_events)
        if M > 100:
            logger.warning(f'Too many events generated for this frame: num_iter={M}>100 events')
        eng.reserve_iters(M)
        n = M if M > 0 else 1
        ts_dev = None
        if tape:
            ts_step = (t_frame - t_prev) / n
            ts = self._tape.linspace(t_prev + ts_step, t_frame, n)
            ts_dev = _as_f32_tensor(ts).to(dev).view(1, n)
        if M == 0 and self.no_events_warning_count < 100:
            logger.warning(f'no signal events generated for frame #{self.frame_counter:,} at t={t_frame:.4f}s')
            self.no_events_warning_count += 1

        eng.rank(P, fidx, ts_dev)
        itc = eng.read_iter_counts(fidx, M)[0]  # [M+1][2]; last row = shot pair
        perms = []
        if tape:
            for i in range(M):
                n_i = int(itc[i, 0]) + int(itc[i, 1])
                perms.append(self._tape.randperm(n_i, fidx, i) if n_i > 0 else None)
            if self.shot_noise_rate_hz > 0 and not self.photoreceptor_noise:
                u = _as_f32_tensor(self._tape.rand((H, W))).to(dev)
                eng.shot(P, frame_dev, fidx, u)
                eng.rank(P, fidx, ts_dev)
                itc = eng.read_iter_counts(fidx, M)[0]
        n_events = int(itc.sum())
        n_on = int(itc[:, 0].sum())
        n_off = int(itc[:, 1].sum())
        n_signal = int(itc[:M].sum())

        events = None
        ev = eng.event_buffer(max(n_events, 1))
        eng.emit(P, fidx, ev, ts_dev, ev_offset0=[0])
        if tape and n_signal > 0 and any(q is not None for q in perms):
            # events_curr_iter[idx] for every iteration (emulator.py:868-869) as ONE gather: the per-iteration
            # permutations, offset to absolute rows, in one pinned upload and one launch
            if eng._events_tmp is None or eng._events_tmp.shape != ev.shape:
                eng._events_tmp = torch.empty_like(ev)
            out = eng._events_tmp
            idx_all = np.empty(n_signal, dtype=np.int32)
            row = 0
            for i in range(M):
                n_i = int(itc[i, 0]) + int(itc[i, 1])
                if n_i > 0:
                    idx = perms[i]
                    idx = idx.numpy() if torch.is_tensor(idx) else np.asarray(idx)
                    np.add(idx, row, out=idx_all[row:row + n_i], casting="unsafe")
                row += n_i
            pin = getattr(self, "_idx_pin", None)
            if pin is None or pin.numel() < n_signal:
                cap_n = max(n_signal, 1 << 16) * 2
                self._idx_pin = pin = torch.empty(cap_n, dtype=torch.int32).pin_memory()
                self._idx_dev = torch.empty(cap_n, dtype=torch.int32, device=dev)
            pin[:n_signal].copy_(torch.from_numpy(idx_all))
            idx_dev = self._idx_dev[:n_signal]
            idx_dev.copy_(pin[:n_signal], non_blocking=True)
            eng.permute(ev, out, idx_dev, 0, n_signal)
            if n_events > n_signal:
                out[0, n_signal:n_events] = ev[0, n_signal:n_events]
            eng._events, eng._events_tmp = out, ev
            ev = out
        if n_events > 0:
            events = ev[0, :n_events].cpu().numpy()

        self.num_events_on += n_on
        self.num_events_off += n_off
        self.num_events_total += n_events

        if events is not None:
            self._write_events(events, n_signal, events_dev=ev[0, :n_events])
        if self.frame_ev_idx_dataset is not None:
            self.frame_ev_idx_dataset[self.frame_counter - 1] = self.dvs_h5_dataset.shape[0]
        if self.record_single_pixel_states is not None:
            self._record_single_pixel(new_frame, t_frame, events, n_signal)
        self.t_previous = t_frame
        return events

    # ------------------------------------------------------------- research tooling (host pass-throughs)
    def _wants_states(self):
        return self.show_dvs_model_state is not None or self.record_single_pixel_states is not None

    def _after_count_states(self, new_frame, t_frame):
        """The attributes the reference holds at emulator.py:756 (after the leak, before the events), as device views; then
        emulator.py:756-767: show / save the requested ones."""
        eng = self._engine
        sd = torch.float64 if eng.f64_state else torch.float32
        self.new_frame = new_frame
        self.log_new_frame = eng.plane(self._state_planes[0]).to(torch.float32)  # lin_log returns float32 (emulator_utils.py:45)
        self.c_minus_s_frame = eng.plane(self._state_planes[1]).to(sd) if self.csdvs_enabled else None
        self.diff_frame = eng.plane(self._state_planes[2]).to(sd)
        if self.show_dvs_model_state is None:
            return
        cv2 = self._host_cv2()
        for s in self.show_dvs_model_state:
            if s in self.dont_show_list:
                continue
            f = getattr(self, s, None)
            if f is None:
                logger.error(f'{s} does not exist so we cannot show it')
                self.dont_show_list.append(s)
            elif cv2 is not None:
                self._show(f, s)
        if cv2 is not None:
            try:
                k = cv2.waitKey(30)
            except Exception:
                k = -1
            if k == 27 or k == ord('x'):
                raise SystemExit(0)  # v2e_quit()

    def _host_cv2(self):
        if self._cv2 is None and not self._show_warned:
            try:
                import cv2
                if not callable(getattr(cv2, "imshow", None)):
                    raise ImportError("cv2 has no imshow")
                self._cv2 = cv2
            except Exception as e:
                logger.warning("v2e_amd.EventEmulator: show_dvs_model_state / save_dvs_model_state need OpenCV on the host (%s): "
                               "nothing is displayed or saved; the state planes stay readable as attributes (%s)", e,
                               ", ".join(self.show_dvs_model_state))
                self._show_warned = True
        return self._cv2

    def _show(self, inp, name):
        """emulator.py:580-617: normalise by MODEL_STATES[name], show, and append to <name>.avi with save_dvs_model_state."""
        cv2 = self._cv2
        img = np.array(inp.detach().cpu().numpy() if torch.is_tensor(inp) else inp, dtype=np.float64)
        (lo, hi) = EventEmulator.MODEL_STATES[name]
        img = (img - lo) / (hi - lo)
        try:
            cv2.namedWindow(name, cv2.WINDOW_NORMAL)
            if name not in self.show_list:
                self.show_list.append(name)
                if self.save_dvs_model_state:
                    fn = os.path.join(self.output_folder or "", name + '.avi')
                    self.video_writers[name] = cv2.VideoWriter(fn, cv2.VideoWriter_fourcc(*'XVID'), 30,
                                                               (int(self.output_width or img.shape[1]),
                                                                int(self.output_height or img.shape[0])))
            for org, col in (((0, img.shape[0]), (0, 0, 0)), ((1, img.shape[0] - 1), (255, 255, 255))):
                cv2.putText(img, f'fr:{self.frame_counter} t:{self.t_previous:.4f}s', org=org, fontScale=1.3, color=col,
                            fontFace=cv2.FONT_HERSHEY_PLAIN, thickness=1)
            cv2.imshow(name, img)
            if self.save_dvs_model_state:
                g = (img * 255).astype(np.uint8)
                self.video_writers[name].write(np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)))
        except Exception as e:  # no display
            logger.warning("v2e_amd.EventEmulator: cannot show %s (%s)", name, e)
            self.dont_show_list.append(name)

    def _record_single_pixel(self, new_frame, t_frame, events, n_signal):
        """emulator.py:985-1009.  The index tuple addresses the [H, W] planes as the reference's does (plane[a, b]); the
        final per-pixel ON / OFF counts are the frame's signal events at that pixel (what final_pos/neg_evts_frame hold)."""
        ij = self.record_single_pixel_states
        if self.single_pixel_sample_count < self.SINGLE_PIXEL_MAX_SAMPLES:
            k = self.single_pixel_sample_count
            if k % 250 == 0:
                logger.info(f'recorded {k} single pixel states')
            st = self.single_pixel_states
            st['time'][k] = t_frame
            st['new_frame'][k] = float(new_frame[ij])
            st['base_log_frame'][k] = float(self.base_log_frame[ij])
            st['lp_log_frame'][k] = float(self.lp_log_frame[ij])
            st['log_new_frame'][k] = float(self.log_new_frame[ij])
            st['pos_thres'][k] = self.pos_thres if type(self.pos_thres) is float else float(self.pos_thres[ij])
            st['neg_thres'][k] = self.neg_thres if type(self.neg_thres) is float else float(self.neg_thres[ij])
            st['diff_frame'][k] = float(self.diff_frame[ij])
            on = off = 0
            if events is not None and n_signal > 0:
                sig = events[:n_signal]
                here = (sig[:, 2] == ij[0]) & (sig[:, 1] == ij[1])  # rows are [t, x = column, y = row, p]
                on = int(np.count_nonzero(here & (sig[:, 3] > 0)))
                off = int(np.count_nonzero(here & (sig[:, 3] < 0)))
            st['final_neg_evts_frame'][k] = off
            st['final_pos_evts_frame'][k] = on
            self.single_pixel_sample_count += 1
        else:
            self.save_recorded_single_pixel_states()
            self.record_single_pixel_states = None

    def save_recorded_single_pixel_states(self):  # emulator.py:431-437
        import pickle
        try:
            with open(self.SINGLE_PIXEL_STATES_FILENAME, 'wb') as outfile:
                pickle.dump(self.single_pixel_states, outfile, protocol=pickle.HIGHEST_PROTOCOL)
                logger.info(f'saved single pixel states with {self.single_pixel_sample_count} samples to '
                            f'{self.SINGLE_PIXEL_STATES_FILENAME}')
        except Exception as e:
            logger.error(f'could not save pickled pixel states, got {e}')

    def _write_events(self, events, n_signal, events_dev=None):
        """emulator.py:953-977: append this frame's events to the open sinks.  The integer conversions of the HDF5 rows
        (emulator.py:957-960), the AEDAT-2.0 words and the text columns run on the GPU on the device-resident rows."""
        if not (self.dvs_h5 is not None or self.dvs_aedat2 is not None or self.dvs_aedat4 is not None or self.dvs_text is not None):
            return
        ev_dev = events_dev if events_dev is not None else torch.from_numpy(np.ascontiguousarray(events, dtype=np.float32)).to(
            self._engine.device)
        sig = n_signal if self.label_signal_noise else None
        if self.dvs_h5 is not None:
            from .sinks import pack_h5
            temp = pack_h5(ev_dev).cpu().numpy().view(np.uint32)
            self.dvs_h5_dataset.resize(self.dvs_h5_dataset.shape[0] + temp.shape[0], axis=0)
            self.dvs_h5_dataset[-temp.shape[0]:] = temp
        if self.dvs_aedat2 is not None:
            self.dvs_aedat2.appendEvents(ev_dev, n_signal=sig)
        if self.dvs_aedat4 is not None:
            label = None
            if self.label_signal_noise:
                label = np.zeros(len(events), dtype=bool)
                label[:n_signal] = True
            self.dvs_aedat4.appendEvents(events, signnoise_label=label)
        if self.dvs_text is not None:
            self.dvs_text.appendEvents(ev_dev, n_signal=sig)

    # ------------------------------------------------------------- centre-surround
    MAX_CHANGE_TO_TERMINATE_EULER_SURROUND_STEPPING = 1e-5  # emulator.py:52
    CSDVS_RUN_MAX_STEPS = 8192  # Euler steps per frame a device-resident run enqueues up front
    CSDVS_RUN_MAX_TOTAL_STEPS = 65536  # ... and per run: every step is a launch (a graph node), taken or not (round-4 advisor)

    def _csdvs_step_params(self, delta_time):
        """emulator.py:1066-1096: (num_steps, alpha_p, alpha_h) of one frame interval, with the reference's warnings and its
        refusal of a diverging diffuser."""
        abs_min_tau_p = 1e-9
        tau_p = abs_min_tau_p if (self.cs_tau_p_ms is None or self.cs_tau_p_ms == 0) else self.cs_tau_p_ms * 1e-3
        tau_h = abs_min_tau_p / (self.cs_lambda_pixels ** 2) if (self.cs_tau_h_ms is None or self.cs_tau_h_ms == 0) \
            else self.cs_tau_h_ms * 1e-3
        min_tau = min(tau_p, tau_h)
        NUM_STEPS_PER_TAU = 5
        num_steps = int(np.ceil((delta_time / min_tau) * NUM_STEPS_PER_TAU))
        actual_delta_time = delta_time / num_steps
        if num_steps > 1000 and not self.cs_steps_warning_printed:
            if self.cs_tau_p_ms == 0:
                logger.warning(f'You set time constant cs_tau_p_ms to zero which set the minimum tau of {abs_min_tau_p}s')
            logger.warning(f'CSDVS timestepping of diffuser could take up to {num_steps} steps per frame for Euler delta time '
                           f'{actual_delta_time:.3g}s; simulation of each frame will terminate when max change is smaller than '
                           f'{self.MAX_CHANGE_TO_TERMINATE_EULER_SURROUND_STEPPING}')
            self.cs_steps_warning_printed = True
        alpha_p = actual_delta_time / tau_p
        alpha_h = actual_delta_time / tau_h
        if alpha_p >= 1 or alpha_h >= 1:
            logger.error(f'CSDVS update alpha (of IIR update) is too large; simulation would explode: '
                         f'alpha_p={alpha_p:.3f} alpha_h={alpha_h:.3f}')
            self.cs_alpha_warning_printed = True
            raise SystemExit(1)  # v2e_quit(1)
        if alpha_p > .25 or alpha_h > .25:
            logger.warning(f'CSDVS update alpha (of IIR update) is too large; simulation will be inaccurate: '
                           f'alpha_p={alpha_p:.3f} alpha_h={alpha_h:.3f}')
            self.cs_alpha_warning_printed = True
        return num_steps, alpha_p, alpha_h

    def _update_csdvs(self, P, frame_dev, t_prev, t_frame, fidx):
        """emulator.py:1061-1124.  The host part (step count, IIR coefficients, warnings, the refusal of a diverging
        diffuser) as the reference writes it; the stepping loop is v2e_csdvs_update on the device planes, driven by the
        coming frame's lp_log_frame (v2e_emu_lp_preview: the low-pass applied to the state as it is)."""
        eng = self._engine
        num_steps, alpha_p, alpha_h = self._csdvs_step_params(t_frame - t_prev)
        sur, scratch, lp_new = self._cs_planes
        eng.lp_preview(P, frame_dev, [t_prev], [t_frame], fidx, lp_new)
        steps = eng.csdvs_update(lp_new, sur, scratch, alpha_p, alpha_h, num_steps, self.MAX_CHANGE_TO_TERMINATE_EULER_SURROUND_STEPPING)
        self.cs_steps_taken.append(steps)

    # ------------------------------------------------------------- device-resident clip
    def generate_events_batch(self, frames, t_frames, return_device=False, use_graph=True, cap=None):
        """Philox mode: run a whole clip [F,H,W] with timestamps t_frames[F] on device.

        Equivalent to calling generate_events once per frame (same events, same order) but
        with no host synchronisation between frames.  Returns (events, counts): events is
        an [N,4] float32 array (or device tensor) of all frames' events concatenated,
        counts[f] the number of events of frame f (0 for the very first frame).
        """
        pipelined = None
        if use_graph is True:
            # the default: one hipGraph per run -- from the SECOND run of a length on.  Capturing and instantiating the graph costs ~2.3 ms,
            # more than it saves a run that is never repeated (an emulator per clip, BASELINE configs[2]: 88.3 -> 86.0 ms per clip)
            seen = self.__dict__.setdefault("_blocking_lengths", {})
            n = int(frames.shape[0])
            if seen.get(n, 0) == 0:
                use_graph, pipelined = 0, False
            seen[n] = seen.get(n, 0) + 1
        return self.generate_events_batch_async(frames, t_frames, return_device=return_device, use_graph=use_graph,
                                                cap=cap, _single_buffer=True, pipelined=pipelined).result()

    def generate_events_batch_async(self, frames, t_frames, return_device=False, use_graph=True, cap=None,
                                    _single_buffer=False, pipelined=None, frames_resident=False):
        """generate_events_batch without waiting for the device: enqueues the run and returns a handle whose
        result() gives (events, counts).  The host can prepare and enqueue the next run (it executes behind this one
        on the same stream) before reading this one's result: two sets of event / record buffers alternate, so a
        result stays valid until the second-next call (a handle still unread at that point is collected by that call, device rows
        copied aside).  Pixel state, frame counter and t_previous advance at enqueue
        time; the event counters (num_events_*) when result() is called; errors (capacity, max_iters) are raised there.
        use_graph: True / 1 one hipGraph per run (what the blocking call uses: lowest latency of a single run), 0 plain launches.
        pipelined (default: on where use_graph is 0): consecutive runs overlap -- the run's upload and first records go out beside the run
        before, its last emission batches finish beside the run after, the current stream carries the chain's launches only
        (v2e_emu_run, 0 | 1024): the throughput mode of a loop that keeps one run enqueued ahead (bench.py).  The current stream then
        orders the pixel state only; result() waits for the rows.  frames_resident=True: the caller vouches that nothing enqueued on
        the current stream still writes `frames`; the run's head then need not wait for the chain of the run before.
        """
        if getattr(self, "_failed", None):
            raise _capi.V2EAmdError("a previous device-resident run failed (%s); call reset()" % self._failed)
        if self.rng_mode != "philox":
            raise ValueError("generate_events_batch needs rng_mode='philox' (tape mode needs the host per frame)")
        if self._wants_states():
            raise ValueError("show_dvs_model_state / record_single_pixel_states read the state after every frame: use "
                             "generate_events per frame")
        if isinstance(frames, np.ndarray):
            if frames.dtype not in (np.uint8, np.float32, np.float64):
                frames = frames.astype(np.float64)
        F = int(frames.shape[0])
        H, W = int(frames.shape[1]), int(frames.shape[2])
        t_frames = [float(t) for t in t_frames]
        if len(t_frames) != F:
            raise ValueError("need one timestamp per frame")
        eng = self._ensure_engine(H, W)
        frames_dev = eng.to_device_frame(frames)
        counts = np.zeros(F, dtype=np.int64)
        start = 0
        if not self._initialized:
            if t_frames[0] < self.t_previous:
                raise ValueError("frame time must not decrease")
            self.frame_counter += 1
            self._first_frame(frames_dev[0], H, W, t_frames[0])
            start = 1
        nrun = F - start
        if nrun <= 0:
            empty = torch.empty((0, 4), dtype=torch.float32, device=eng.device)
            return _PendingRun(self, None, None, None, counts, start, return_device, empty)
        t_prev = []
        tp = float(self.t_previous)
        for f in range(start, F):
            if t_frames[f] < tp:
                raise ValueError("this frame time={} must be later than previous frame time={}".format(
                    t_frames[f], tp))
            t_prev.append(tp)
            tp = t_frames[f]
        if self.photoreceptor_noise:  # emulator.py:694-703, the noise plane stays on the device for the whole run
            if self.photoreceptor_noise_arr is None:
                self._pn_plane = torch.zeros((1, eng.npx_pad), dtype=torch.float64, device=eng.device)
                self.photoreceptor_noise_arr = eng.plane(self._pn_plane)
            rates = [1.0 / (t_frames[f] - t_prev[f - start]) for f in range(start, F)]
            if self.photoreceptor_noise_vrms is not None:
                self._pn_vrms = float(self.photoreceptor_noise_vrms)
            else:
                # the reference recomputes the amplitude when the sample rate has moved by 10 % from the rate it was last
                # computed for (emulator_utils.py:217-220); inside one device-resident run the amplitude is one number, so
                # every frame of the run must stay within 10 % of the rate that number belongs to
                if self._pn_last_rate is None or abs(rates[0] / self._pn_last_rate - 1) >= 0.1:
                    self._pn_vrms = photoreceptor_noise_vrms(self.shot_noise_rate_hz, self.cutoff_hz, rates[0],
                                                             self.pos_thres_nominal, self.neg_thres_nominal, self.sigma_thres)
                    self._pn_last_rate = rates[0]
                if any(abs(r / self._pn_last_rate - 1) >= 0.1 for r in rates):
                    raise ValueError("photoreceptor_noise: the frame interval moves by more than 10 % from the rate the noise "
                                     "amplitude was computed for inside this run; split the run there (or use generate_events "
                                     "per frame)")
            eng.set_pnoise(self._pn_plane, None)
            if isinstance(use_graph, bool) or use_graph in (0, 1):
                use_graph = int(use_graph) | 16  # the pipeline that carries the noise plane
        if self.scidvs and (isinstance(use_graph, bool) or use_graph in (0, 1)):
            use_graph = int(use_graph) | 16  # the kernels that carry the SCIDVS planes
        cs_steps_dev = None
        if self.csdvs_enabled:
            # emulator.py:1061-1124 inside the run: per frame the host part (step count, coefficients) as the reference computes
            # it; the stepping loop is enqueued whole, its stop rule evaluated on the device (v2e_emu_set_csdvs_run)
            pars = [self._csdvs_step_params(t_frames[f] - t_prev[f - start]) for f in range(start, F)]
            if max(q[0] for q in pars) > self.CSDVS_RUN_MAX_STEPS:
                raise ValueError("generate_events_batch with cs_lambda_pixels: %d Euler steps per frame (cs_tau_p_ms / cs_lambda_pixels) "
                                 "would all be enqueued up front; beyond %d use generate_events per frame (its loop ends as soon as "
                                 "the diffuser has settled)" % (max(q[0] for q in pars), self.CSDVS_RUN_MAX_STEPS))
            if sum(q[0] for q in pars) > self.CSDVS_RUN_MAX_TOTAL_STEPS:
                raise ValueError("generate_events_batch with cs_lambda_pixels: %d Euler steps over the run's %d frames would all be "
                                 "enqueued up front (mostly no-op launches once the diffuser has settled); beyond %d split the run or use "
                                 "generate_events per frame" % (sum(q[0] for q in pars), nrun, self.CSDVS_RUN_MAX_TOTAL_STEPS))
            cs_steps_dev = torch.zeros(nrun, dtype=torch.int32, device=eng.device)
            import ctypes as C
            ns = (C.c_int * nrun)(*[q[0] for q in pars])
            ap = (C.c_double * nrun)(*[q[1] for q in pars])
            ah = (C.c_double * nrun)(*[q[2] for q in pars])
            sur, scratch, lp_new = self._cs_planes
            _capi.check(eng.lib.v2e_emu_set_csdvs_run(eng._h, scratch.data_ptr(), lp_new.data_ptr(), ap, ah, ns, nrun,
                                                      float(self.MAX_CHANGE_TO_TERMINATE_EULER_SURROUND_STEPPING),
                                                      cs_steps_dev.data_ptr()), "v2e_emu_set_csdvs_run")
            if isinstance(use_graph, bool) or use_graph in (0, 1):
                use_graph = int(use_graph) | 16  # the per-frame kernels carry the surround
        P = self._params()
        if cap is None:
            # 4 events per pixel and frame (the reference never drops events; a clip that exceeds this raises below and
            # leaves the instance unusable until reset()), bounded to 4 GiB of rows for very long runs
            per_frame = max(4 * H * W, 1 << 16)
            cap = per_frame * min(nrun, max(64, (4 << 30) // (16 * per_frame)))
        # two buffer sets alternate between asynchronous runs; the synchronous call has a third of its own, so that a pending
        # handle whose result() has not been read yet is never overwritten by it
        which = 2 if _single_buffer else self.__dict__.setdefault("_async_flip", 0)
        if not _single_buffer:
            self._async_flip = which ^ 1
        # the run that last used this buffer set and has not been read yet: collect it now -- its rows and records (and, for pipelined
        # runs, its scratch set's pinned records) are about to be reused, and a later result() would hand back this run's data instead
        slots = self.__dict__.setdefault("_async_slots", {})
        old = slots.get(which)
        old = old() if old is not None else None  # (weak: a handle the caller has dropped needs no collecting, and a strong reference
        #                                            here would tie the emulator and its device memory into a cycle only the GC breaks)
        if old is not None and old._res is None and old.recs is not None:
            ev_old, counts_old = old.result()
            if old.return_device and ev_old is not None:
                old._res = (ev_old.clone(), counts_old)
        ev = eng.event_buffer(cap, which)
        recs = eng.alloc_recs(nrun, which)
        ug = int(use_graph)
        if getattr(self, "_refr_mostly_on", False):
            ug |= 128  # one frame per launch: the speculating chain would redo most of its launches
        if (ug & 3) == 0 and (pipelined or pipelined is None) and not _single_buffer:
            ug |= 1024 | (2048 if frames_resident else 0)
        eng.run(P, frames_dev[start:], t_prev, t_frames[start:], self.frame_counter, ev, recs, use_graph=ug)
        ticket = None
        if ug & 1024:
            # the library may have enqueued the run in pieces on streams of its own: result() waits for them on the host; a device-side
            # event (the `done` property) is only built for a caller that asks for one
            done, ticket = None, eng.run_ticket()
        if ticket is None or ticket < 0:
            ticket = None
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(eng.device))
        self.frame_counter += nrun
        self.t_previous = t_frames[-1]
        dts = np.asarray(t_frames[start:]) - np.asarray(t_prev)
        pend = _PendingRun(self, ev, recs, done, counts, start, return_device, None, dts, ticket)
        pend.cs_steps_dev = cs_steps_dev
        slots[which] = weakref.ref(pend)
        return pend

    def _finish_run(self, pend):
        eng = self._engine
        if getattr(self, "_failed", None):  # an earlier pending run failed: the state this run started from is not the clip's
            raise _capi.V2EAmdError("a previous device-resident run failed (%s); this run started from its state: call reset()"
                                    % self._failed)
        if pend.ticket is not None:
            eng.run_wait(pend.ticket)  # (host-blocking; no device-side wait is put into any stream's queue)
            r = eng.run_recs(pend.ticket)[:, 0]
        else:
            r = eng.read_recs_after(pend.recs, pend.done)[:, 0]
        if self.refractory_period_s > 0:  # emulator.py:830 on the frames just run: how often was the rule active?
            m = np.maximum(r["max_events"], 1)
            self._refr_mostly_on = bool(np.mean(self.refractory_period_s > pend.dts / m) > 0.05)
        # the device state has already advanced past the failed run: the instance refuses further frames until reset()
        err = None
        if (r["flags"] & _capi.FLAG_ITERS_CLAMPED).any():
            err = ("a pixel produced more than max_iters=%d events in one frame; construct with a larger max_iters"
                   % eng.max_iters)
        elif (r["flags"] & _capi.FLAG_SYNC_TIMEOUT).any():
            err = ("in-kernel workgroup rendezvous of a redo pass timed out (GPU shared with another process?); "
                   "use_graph | 128 runs one frame per launch, which needs none")
        elif (r["flags"] & _capi.FLAG_EVENTS_DROPPED).any():
            err = "event buffer capacity %d exceeded (needed %d); pass a larger cap" % (
                pend.ev.shape[1], int(r["n_events"].sum()))
        if err is not None:
            self._failed = err
            raise _capi.V2EAmdError(err + " -- the pixel state is past this run: call reset() before feeding more frames")
        pend.counts[pend.start:] = r["n_events"]
        pend.rec_host = r
        if getattr(pend, "cs_steps_dev", None) is not None:
            self.cs_steps_taken.extend(int(v) for v in pend.cs_steps_dev.cpu().tolist())  # emulator.py:1124
        total = int(r["n_events"].sum())
        self.num_events_total += total
        self.num_events_on += int(r["n_on"].sum())
        self.num_events_off += int(r["n_off"].sum())
        out = pend.ev[0, :total]
        if pend.return_device:
            return out, pend.counts
        return (out.cpu().numpy() if total > 0 else None), pend.counts
