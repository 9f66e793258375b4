#!/usr/bin/env python
"""Sensor-size fixtures of the drop-in's TAPE-MODE kernels that do not depend on torch's random generator (round-4 review, weak
item 9: on a torch build that draws differently the tape_live_* tests cannot be compared).

The unmodified reference (imported from /root/reference) runs with torch.normal / randn / rand / randperm replaced by
tests/golden/portable_tape.PortableSource; stored: per-frame event counts and SHA-256 digests, digests of the final planes, counters.

  tape_portable_defaults_346x260   12 frames of the integer gradient clip, v2e CLI defaults (refractory rule, shot noise, leak)
  tape_portable_noisy_346x260      12 frames, set_dvs_params('noisy')
(noise_rate_cov_decades = 0 in both: exp(0 * randn) = 1 exactly, so no host's float32 exp enters the digests)
"""
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
import make_golden as mg  # noqa: E402
from portable_tape import PortableSource  # noqa: E402
from v2e_amd.synth import int_gradient_frames  # noqa: E402


def make(name, frames, times, kw, preset=None, seed=77, frame_spec=None):
    EE = rh.ref_emulator_cls()
    torch.set_num_threads(1)
    with PortableSource(seed) as src:
        ref = EE(seed=seed, device='cpu', **kw)
        if preset:
            ref.set_dvs_params(preset)
            ref.noise_rate_cov_decades = 0.0
        evs = [ref.generate_events(f, t) for f, t in zip(frames, times)]
    assert float(ref.noise_rate_array.min()) == 1.0 == float(ref.noise_rate_array.max())
    out = {
        "times": np.asarray(times, np.float64), "kw": json.dumps(kw), "preset": preset or "", "seed": seed,
        "shape": np.asarray(frames[0].shape), "torch_version": torch.__version__, "draw_calls": src.tape.calls,
        "n_events": np.asarray([0 if e is None else len(e) for e in evs], np.int64),
        "ev_sha": np.asarray([mg.sha(e) if e is not None else "" for e in evs]),
        "base_sha": mg.sha(ref.base_log_frame.numpy()), "lp_sha": mg.sha(ref.lp_log_frame.numpy()),
        "counters": np.array([ref.num_events_total, ref.num_events_on, ref.num_events_off], np.int64),
        "frame_spec": json.dumps(frame_spec or {}),
    }
    if getattr(ref, "timestamp_mem", None) is not None and ref.refractory_period_s > 0:
        out["ts_mem_sha"] = mg.sha(ref.timestamp_mem.numpy())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-34s frames=%d events=%d draws=%d  %.0f KB" % (name, len(frames), ref.num_events_total, src.tape.calls, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    spec = dict(gen="int_gradient_frames", n=12, H=260, W=346, seed=5, noise=6)
    frames = int_gradient_frames(spec["n"], spec["H"], spec["W"], seed=spec["seed"], noise=spec["noise"], as_array=True)
    times = [k / 300.0 for k in range(spec["n"])]
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01, shot_noise_rate_hz=2.0,
              refractory_period_s=.0005, noise_rate_cov_decades=0.0)
    make("tape_portable_defaults_346x260", list(frames), times, kw, frame_spec=spec)
    make("tape_portable_noisy_346x260", list(frames), times, dict(noise_rate_cov_decades=0.0), preset="noisy", frame_spec=spec)
