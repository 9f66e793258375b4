#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by running the *reference* v2e
(read-only tree at /root/reference, imported in-process via ref_harness.py).

Run here (CPU container):  python tests/golden/make_golden.py
The fixtures travel to the GPU box; the reference does not.

Two kinds of fixture:

* tape_*.npz  -- the reference runs with torch's own global generator (seeded).  Every
  random draw it makes (torch.normal/randn/rand/randperm) and every torch.linspace
  result is recorded, in call order, next to the events of every frame and the final
  state planes.  A test replays the tape through the HIP kernels (tape mode) and must
  reproduce the events bit for bit, including the shuffled order.

* philox_*.npz -- the reference runs unmodified arithmetic but its random *source* is
  swapped for the portable counter-based generator of include/v2e_detmath.h (the same
  numbers the kernels generate in Philox mode): torch.randn/rand/randperm are patched
  for frames >= 1 and the per-pixel parameter planes drawn on frame 0 (thresholds,
  noise rates) are overwritten with the Philox-derived planes.  The in-kernel timestamp
  formula is asserted equal to torch.linspace on every frame at generation time.  Only
  digests are stored for the full-size cases (events are tens of MB).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness as rh  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from v2e_amd.synth import int_gradient_frames  # noqa: E402

DEFAULTS = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01,
                shot_noise_rate_hz=.001, refractory_period_s=.0005)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------- tape kind
class TapeRecorder:
    def __init__(self):
        self.items = []
        self._orig = {}

    def __enter__(self):
        for name in ("normal", "randn", "rand", "randperm", "linspace", "exp"):
            self._orig[name] = getattr(torch, name)
        rec = self

        def normal(*a, **k):
            r = rec._orig["normal"](*a, **k)
            rec.items.append(("normal", r.numpy().copy().ravel()))
            return r

        def randn(*a, **k):
            r = rec._orig["randn"](*a, **k)
            rec.items.append(("randn", r.numpy().copy().ravel()))
            return r

        def rand(*a, **k):
            r = rec._orig["rand"](*a, **k)
            rec.items.append(("rand", r.numpy().copy().ravel()))
            return r

        def randperm(*a, **k):
            r = rec._orig["randperm"](*a, **k)
            rec.items.append(("randperm", r.numpy().copy()))
            return r

        def linspace(*a, **k):
            r = rec._orig["linspace"](*a, **k)
            rec.items.append(("linspace", r.numpy().copy()))
            return r

        def exp(x, *a, **k):
            r = rec._orig["exp"](x, *a, **k)
            # the only torch.exp on the path: noise_rate_array (emulator.py:504)
            rec.items.append(("noise_rate", r.numpy().copy().ravel()))
            return r

        torch.normal, torch.randn, torch.rand = normal, randn, rand
        torch.randperm, torch.linspace, torch.exp = randperm, linspace, exp
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)


def make_tape_fixture(name, frames, times, kw, preset=None, seed=42):
    EE = rh.ref_emulator_cls()
    torch.set_num_threads(1)
    with TapeRecorder() as rec:
        ref = EE(seed=seed, device='cpu', **kw)
        if preset:
            ref.set_dvs_params(preset)
        evs = [ref.generate_events(f, t) for f, t in zip(frames, times)]
    # the noise-rate randn is consumed by exp(): drop it from the tape (the host replays
    # the post-exp plane; see RecordedTape.exp_noise_rate)
    items = []
    for k, (kind, arr) in enumerate(rec.items):
        items.append((kind, arr))
    out = {
        "frames": np.stack(frames),
        "times": np.asarray(times, np.float64),
        "kw": json.dumps(kw), "preset": preset or "", "seed": seed,
        "n_items": len(items),
        "torch_version": torch.__version__,
        "base_final": ref.base_log_frame.numpy(),
        "lp_final": ref.lp_log_frame.numpy(),
        "counters": np.array([ref.num_events_total, ref.num_events_on, ref.num_events_off], np.int64),
    }
    if getattr(ref, "timestamp_mem", None) is not None and ref.refractory_period_s > 0:
        out["ts_mem_final"] = ref.timestamp_mem.numpy()
    if getattr(ref, "scidvs", False):
        out["scidvs_highpass_final"] = ref.scidvs_highpass.numpy()
        out["scidvs_tau"] = ref.scidvs_tau_arr.numpy()
    for k, (kind, arr) in enumerate(items):
        out["tape_%05d_%s" % (k, kind)] = arr
    for k, e in enumerate(evs):
        out["ev_%04d" % k] = e if e is not None else np.zeros((0, 4), np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-34s frames=%d events=%d tape_items=%d  %.0f KB" % (
        name, len(frames), ref.num_events_total, len(items), os.path.getsize(path) / 1024))


# --------------------------------------------------------------- philox kind
class PhiloxSource:
    """Patches torch's random calls so the reference consumes the Philox streams."""

    def __init__(self, seed, H, W):
        self.seed, self.H, self.W = seed, H, W
        self.frame_idx = 0
        self._orig = {}
        self.linspace_checked = 0
        self.pnoise = False     # photoreceptor_noise: the first randn of a frame is the noise draw
        self.randn_calls = 0

    def __enter__(self):
        for name in ("randn", "rand", "randperm", "linspace"):
            self._orig[name] = getattr(torch, name)
        src = self

        def randn(*a, **k):
            src.randn_calls += 1
            if src.pnoise and src.randn_calls == 1:  # emulator.py:698 comes before the leak draw of the frame
                return torch.from_numpy(orc.philox_pnoise(src.seed, 0, src.frame_idx, src.H * src.W).reshape(src.H, src.W))
            r, _ = orc.philox_frame(src.seed, 0, src.frame_idx, src.H * src.W)
            return torch.from_numpy(r.reshape(src.H, src.W))

        def rand(*a, **k):
            _, u = orc.philox_frame(src.seed, 0, src.frame_idx, src.H * src.W)
            return torch.from_numpy(u.reshape(src.H, src.W))

        def randperm(n, *a, **k):
            it = sys._getframe(1).f_locals["i"]  # loop variable of emulator.py:810
            return torch.from_numpy(orc.perm_idx(src.seed, 0, src.frame_idx, it, n))

        def linspace(*a, **k):
            r = src._orig["linspace"](*a, **k)
            f = orc.ts_formula(src.t_prev, src.t_frame, int(k["steps"]))
            if not np.array_equal(r.numpy(), f):
                raise AssertionError("in-kernel timestamp formula != torch.linspace on frame %d" % src.frame_idx)
            src.linspace_checked += 1
            return r

        torch.randn, torch.rand, torch.randperm, torch.linspace = randn, rand, randperm, linspace
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)


def run_reference_philox(frames, times, kw, preset, seed):
    """Reference arithmetic + Philox random source; returns (events per frame, ref)."""
    import math
    EE = rh.ref_emulator_cls()
    torch.set_num_threads(1)
    H, W = frames[0].shape
    ref = EE(seed=seed, device='cpu', **kw)
    if preset:
        ref.set_dvs_params(preset)
    evs = [ref.generate_events(frames[0], times[0])]
    # overwrite the per-pixel parameter planes drawn on frame 0 with the Philox planes
    n_pos, n_neg, n_rate = orc.philox_init(seed, 0, H * W)
    if ref.sigma_thres > 0:
        sig = np.float32(ref.sigma_thres)
        tp = n_pos * sig + np.float32(0.2 if preset else kw["pos_thres"])
        tn = n_neg * sig + np.float32(0.2 if preset else kw["neg_thres"])
        ref.pos_thres = torch.clamp(torch.from_numpy(tp.reshape(H, W)), min=0.01)
        ref.neg_thres = torch.clamp(torch.from_numpy(tn.reshape(H, W)), min=0.01)
        ref.pos_thres_pre_prob = torch.div(ref.pos_thres_nominal, ref.pos_thres)
        ref.neg_thres_pre_prob = torch.div(ref.neg_thres_nominal, ref.neg_thres)
    if getattr(ref, "scidvs", False):  # emulator.py:480-483 with the Philox normal and the deterministic expf
        ref.scidvs_tau_arr = torch.from_numpy(orc.philox_scidvs_tau(seed, 0, H * W).reshape(H, W))
    if ref.leak_rate_hz > 0:
        lnc = np.float32(math.log(10) * ref.noise_rate_cov_decades)
        nr = np.array([orc.lib().v2e_oracle_det_expf(float(np.float32(lnc * v))) for v in n_rate], np.float32)
        ref.noise_rate_array = torch.from_numpy(nr.reshape(H, W))
    with PhiloxSource(seed, H, W) as src:
        src.pnoise = bool(kw.get("photoreceptor_noise"))
        for k in range(1, len(frames)):
            src.frame_idx = k
            src.randn_calls = 0
            src.t_prev = float(ref.t_previous)
            src.t_frame = float(times[k])
            evs.append(ref.generate_events(frames[k], times[k]))
    return evs, ref, src.linspace_checked


def make_philox_fixture(name, frames, times, kw, preset=None, seed=7, store_frames=False, store_events=False,
                        frame_spec=None):
    evs, ref, nlin = run_reference_philox(frames, times, kw, preset, seed)
    out = {
        "times": np.asarray(times, np.float64),
        "kw": json.dumps(kw), "preset": preset or "", "seed": seed,
        "torch_version": torch.__version__,
        "n_events": np.array([0 if e is None else len(e) for e in evs], np.int64),
        "ev_sha": np.array([sha(e) if e is not None else "" for e in evs]),
        "base_sha": sha(ref.base_log_frame.numpy()),
        "lp_sha": sha(ref.lp_log_frame.numpy()),
        "counters": np.array([ref.num_events_total, ref.num_events_on, ref.num_events_off], np.int64),
        "frame_spec": json.dumps(frame_spec or {}),
        "shape": np.array(frames[0].shape, np.int64),
    }
    if ref.refractory_period_s > 0:
        out["ts_mem_sha"] = sha(ref.timestamp_mem.numpy())
    if getattr(ref, "scidvs", False):
        out["scidvs_highpass_final"] = ref.scidvs_highpass.numpy()
        out["base_final"] = ref.base_log_frame.numpy()
    if getattr(ref, "csdvs_enabled", False):
        out["cs_surround_sha"] = sha(ref.cs_surround_frame.numpy())
        out["cs_steps"] = np.asarray(ref.cs_steps_taken, np.int64)
        if store_events:
            out["cs_surround_final"] = ref.cs_surround_frame.numpy()
            out["base_final"] = ref.base_log_frame.numpy()
    if store_frames:
        out["frames"] = np.stack(frames)
    if store_events:
        for k, e in enumerate(evs):
            out["ev_%04d" % k] = e if e is not None else np.zeros((0, 4), np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-34s frames=%d events=%d linspace-checked=%d  %.0f KB" % (
        name, len(frames), ref.num_events_total, nlin, os.path.getsize(path) / 1024))


def moving_dot_frames(n=None):
    md = rh.ref_moving_dot()
    g = md.moving_dot(width=64, height=64, preview=False,
                      arg_list=['--radius', '20', '--t_total', '0.05', '--dt', '1e-4', '--num_particles', '3'])
    frs, tss = [], []
    for _ in range(g.total_frames() if n is None else n):
        f, t = g.next_frame()
        frs.append(np.array(f))
        tss.append(float(t))
    return frs, tss


def main():
    logging_off()
    # ---- tape fixtures (small sensors, full tapes stored)
    fr = int_gradient_frames(12, 40, 48, seed=1, noise=6)
    ts = [i / 300 for i in range(12)]
    kw = dict(DEFAULTS); kw["shot_noise_rate_hz"] = 5.0; kw["leak_rate_hz"] = 0.5
    make_tape_fixture("tape_defaults_40x48", fr, ts, kw)
    make_tape_fixture("tape_noisy_40x48", fr, ts, DEFAULTS, preset="noisy")
    make_tape_fixture("tape_clean_40x48", fr, ts, DEFAULTS, preset="clean")
    kw = dict(DEFAULTS); kw["cutoff_hz"] = 0; kw["shot_noise_rate_hz"] = 5.0; kw["leak_rate_hz"] = 0.5
    make_tape_fixture("tape_f32state_40x48", fr, ts, kw)
    kw = dict(DEFAULTS); kw["sigma_thres"] = 0
    make_tape_fixture("tape_scalarthres_40x48", fr, ts, kw)
    frf = [f.astype(np.float64) * 0.731 + 0.25 for f in int_gradient_frames(12, 33, 37, seed=3, noise=20)]
    tsf = [0.01 + i / 100 for i in range(12)]  # first frame not at t=0
    kw = dict(DEFAULTS); kw["refractory_period_s"] = 0.004; kw["shot_noise_rate_hz"] = 20.0
    make_tape_fixture("tape_refractory_float_33x37", frf, tsf, kw)
    frs, tss = moving_dot_frames(40)
    make_tape_fixture("tape_moving_dot_64x64_40fr", frs, tss, DEFAULTS)

    # ---- philox fixtures
    frs, tss = moving_dot_frames()
    make_philox_fixture("philox_moving_dot_64x64", frs, tss, DEFAULTS, seed=42, store_frames=True, store_events=True)
    spec = dict(gen="int_gradient_frames", n=24, H=260, W=346, seed=1, noise=6)
    fr = int_gradient_frames(24, 260, 346, seed=1, noise=6)
    ts = [i / 300 for i in range(24)]
    make_philox_fixture("philox_defaults_346x260", fr, ts, DEFAULTS, seed=7, frame_spec=spec)
    make_philox_fixture("philox_noisy_346x260", fr, ts, DEFAULTS, preset="noisy", seed=8, frame_spec=spec)
    kw = dict(DEFAULTS); kw["refractory_period_s"] = 0.002; kw["shot_noise_rate_hz"] = 2.0; kw["leak_rate_hz"] = 0.3
    make_philox_fixture("philox_refractory_346x260", fr, ts, kw, seed=9, frame_spec=spec)
    spec = dict(gen="int_gradient_frames", n=6, H=720, W=1280, seed=4, noise=6)
    fr = int_gradient_frames(6, 720, 1280, seed=4, noise=6)
    ts = [i / 600 for i in range(6)]
    make_philox_fixture("philox_noisy_1280x720", fr, ts, DEFAULTS, preset="noisy", seed=10, frame_spec=spec)


def logging_off():
    import logging
    logging.disable(logging.CRITICAL)


if __name__ == "__main__":
    main()
