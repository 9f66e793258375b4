#!/usr/bin/env python
"""Sensor-size fixtures of the DEFAULT mode of the drop-in (tape mode: the reference's own torch MT19937 stream).

The reference (imported from /root/reference) runs unmodified with `seed=S`; what is stored is small: per-frame event
counts and SHA-256 digests of the event arrays, digests of the final state planes, the counters, and a digest of every
random draw in order (so that a replay on another torch build can tell a changed generator from a wrong result).  The GPU
test replays with the LIVE seeded generator (EventEmulator(seed=S, rng_mode='tape')): same torch build => same draws.

  tape_live_defaults_346x260     24 frames of the integer gradient clip, v2e CLI defaults
  tape_live_noisy_346x260        ... set_dvs_params('noisy')
  tape_live_noisy_1280x720       4 frames, 'noisy' (3 600 workgroups: multi-workgroup scans)
  tape_live_moving_dot_64x64     BASELINE configs[0] in full: scripts/moving_dot.py 64x64, 500 frames, seed 42 (8 435 events)
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
import make_golden as mg  # noqa: E402
from v2e_amd.synth import int_gradient_frames  # noqa: E402


def draw_probe(seed):
    """digest of a fixed sequence of draws after manual_seed(seed): identifies the generator implementation"""
    torch.manual_seed(seed)
    h = hashlib.sha256()
    for a in (torch.normal(0.2, 0.03, size=(7, 11), dtype=torch.float32), torch.randn((5, 13), dtype=torch.float32),
              torch.rand(size=(3, 17), dtype=torch.float32), torch.randperm(1000), torch.randperm(70000),
              torch.linspace(start=0.0123, end=0.0456, steps=7, dtype=torch.float32)):
        h.update(np.ascontiguousarray(a.numpy()).tobytes())
    return h.hexdigest()


def exp_probe():
    """torch.exp (float32, CPU) is part of the reference's first frame (noise_rate_array, emulator.py:504) and its last bit
    depends on the host CPU (same torch build, same 'AVX512' capability: the build container and the MI355X box differ):
    the final base_log_frame digest is comparable only where this probe matches."""
    torch.manual_seed(5)
    r = torch.randn((64, 64), dtype=torch.float32)
    return hashlib.sha256(torch.exp(0.23025850929940458 * r).numpy().tobytes()).hexdigest()


def make(name, frames, times, kw, preset=None, seed=42, frame_spec=None):
    EE = rh.ref_emulator_cls()
    torch.set_num_threads(1)
    probe = draw_probe(seed)
    with mg.TapeRecorder() as rec:
        ref = EE(seed=seed, device='cpu', **kw)
        if preset:
            ref.set_dvs_params(preset)
        evs = [ref.generate_events(f, t) for f, t in zip(frames, times)]
    h = hashlib.sha256()
    for kind, arr in rec.items:
        if kind != "noise_rate":
            h.update(np.ascontiguousarray(arr).tobytes())
    out = {
        "times": np.asarray(times, np.float64), "kw": json.dumps(kw), "preset": preset or "", "seed": seed,
        "shape": np.asarray(frames[0].shape), "torch_version": torch.__version__, "draw_probe": probe, "exp_probe": exp_probe(), "draws_sha": h.hexdigest(),
        "n_events": np.asarray([0 if e is None else len(e) for e in evs], np.int64),
        "ev_sha": np.asarray([mg.sha(e) if e is not None else "" for e in evs]),
        "base_sha": mg.sha(ref.base_log_frame.numpy()), "lp_sha": mg.sha(ref.lp_log_frame.numpy()),
        "counters": np.array([ref.num_events_total, ref.num_events_on, ref.num_events_off], np.int64),
        "frame_spec": json.dumps(frame_spec or {}),
    }
    if getattr(ref, "timestamp_mem", None) is not None and ref.refractory_period_s > 0:
        out["ts_mem_sha"] = mg.sha(ref.timestamp_mem.numpy())
    if getattr(ref, "csdvs_enabled", False):
        out["cs_surround_sha"] = mg.sha(ref.cs_surround_frame.numpy())
        out["cs_steps"] = np.asarray(ref.cs_steps_taken, np.int64)
    if frame_spec is None:
        out["frames"] = np.stack(frames)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-34s frames=%d events=%d max/frame=%d  %.0f KB" % (name, len(frames), ref.num_events_total,
                                                              int(out["n_events"].max()), os.path.getsize(path) / 1024))


def main():
    mg.logging_off()
    spec = dict(gen="int_gradient_frames", n=24, H=260, W=346, seed=1, noise=6)
    fr = int_gradient_frames(24, 260, 346, seed=1, noise=6)
    ts = [i / 300 for i in range(24)]
    make("tape_live_defaults_346x260", fr, ts, mg.DEFAULTS, seed=42, frame_spec=spec)
    make("tape_live_noisy_346x260", fr, ts, mg.DEFAULTS, preset="noisy", seed=43, frame_spec=spec)
    spec = dict(gen="int_gradient_frames", n=4, H=720, W=1280, seed=4, noise=6)
    fr = int_gradient_frames(4, 720, 1280, seed=4, noise=6)
    make("tape_live_noisy_1280x720", fr, [i / 600 for i in range(4)], mg.DEFAULTS, preset="noisy", seed=44, frame_spec=spec)
    frs, tss = mg.moving_dot_frames()
    make("tape_live_moving_dot_64x64", frs, tss, mg.DEFAULTS, seed=42)


if __name__ == "__main__":
    main()
