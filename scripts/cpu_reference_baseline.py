#!/usr/bin/env python
"""The CPU baseline BASELINE.json asks for: the UNMODIFIED reference (imported from /root/reference through
tests/golden/ref_harness.py, which stubs only imports that the hot path never calls) timed on this host.

  emulator  v2ecore.emulator.EventEmulator.generate_events (emulator.py:619) on the configs[1] clip: 346x260
            sin*cos gradient video, dt = 1/300 s, v2e CLI default DVS parameters, device='cpu'
  slomo     the slomo.py:338-345, 404-433 arithmetic driven through v2ecore.model.UNet / backWarp at 320x256, U=10
            (tests/golden/make_golden_slomo.ref_interp), seeded random-init weights

each at torch.set_num_threads(1) and at all cores; warm-up, then median of 3 repeats; only the reference calls are
inside the timed region (no frame synthesis, no PNG/disk).  The reference cannot travel to the GPU box
(/root/reference exists only in the build container), so the result is RECORDED: profiles/r06_cpu_reference.json (host, time and repository commit stamped),
which bench.py attaches to its JSON line as cpu_baseline.reference (host named).

usage: python scripts/cpu_reference_baseline.py [--frames 120] [--quick]
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness as rh  # noqa: E402
from make_golden_slomo import ref_interp  # noqa: E402
from v2e_amd.synth import portable_unet_state_dict, sincos_gradient_frames  # noqa: E402

H, W, DT = 260, 346, 1.0 / 300.0
KW = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01,
          shot_noise_rate_hz=.001, refractory_period_s=.0005)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor()


def time_emulator(frames, threads, reps=3, warm=5):
    torch.set_num_threads(threads)
    cls = rh.ref_emulator_cls()
    runs = []
    for r in range(reps):
        emu = cls(seed=1 + r, device="cpu", output_folder=None, **KW)
        n_ev = 0
        for i in range(warm + 1):
            emu.generate_events(frames[i], i * DT)
        t0 = time.perf_counter()
        for i in range(warm + 1, len(frames)):
            e = emu.generate_events(frames[i], i * DT)
            n_ev += 0 if e is None else len(e)
        sec = time.perf_counter() - t0
        nfr = len(frames) - warm - 1
        runs.append((n_ev / sec / 1e6, nfr / sec, n_ev / nfr))
    mev = statistics.median(x[0] for x in runs)
    fps = statistics.median(x[1] for x in runs)
    return {"threads": threads, "Mevents_per_s": round(mev, 3), "frames_per_s": round(fps, 2),
            "events_per_frame": round(runs[0][2], 1), "timed_frames": len(frames) - warm - 1, "repeats": reps}


def time_slomo(B, U, threads, reps=3):
    torch.set_num_threads(threads)
    model = rh.ref_model()
    hh, ww = 256, 320
    flow_net, interp_net = model.UNet(2, 4), model.UNet(12, 5)
    flow_net.load_state_dict({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, 101).items()})
    interp_net.load_state_dict({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()})
    warper = model.backWarp(ww, hh, "cpu")
    g = torch.Generator()
    g.manual_seed(2)
    I0 = torch.rand((B, 1, hh, ww), generator=g) - 0.428
    I1 = torch.rand((B, 1, hh, ww), generator=g) - 0.428
    ts = [(k + 0.5) / U for k in range(U)]
    secs = []
    with torch.no_grad():
        flow_net(torch.cat((I0[:1], I1[:1]), dim=1))  # warm-up (thread pool, mkldnn primitive cache)
        for _ in range(reps):
            t0 = time.perf_counter()
            ref_interp(model, flow_net, interp_net, warper, I0, I1, ts)
            secs.append(time.perf_counter() - t0)
    sec = statistics.median(secs)
    return {"threads": threads, "batch_pairs": B, "U": U, "interpolated_frames_per_s": round(B * U / sec, 3),
            "s_per_batch": round(sec, 2), "repeats": reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=126)
    ap.add_argument("--quick", action="store_true", help="fewer repeats / no single-thread SloMo B=8")
    args = ap.parse_args()
    ncores = os.cpu_count()
    frames = list(sincos_gradient_frames(args.frames, H, W, seed=1))
    out = {"host": {"cpu": cpu_model(), "cores": ncores, "torch": torch.__version__, "numpy": np.__version__,
                    "where": "build container (the reference tree exists only here)"},
           "reference": "SensorsINI/v2e at /root/reference, unmodified (v2ecore.emulator.EventEmulator, v2ecore.model)",
           "emulator": {"workload": "BASELINE configs[1]: 346x260 sin*cos gradient video, dt=1/300 s, CLI default DVS params",
                        "runs": []},
           "slomo": {"workload": "slomo.py:338-433 arithmetic at 320x256 (346x260 source), U=10, random-init weights",
                     "runs": []}}
    reps = 1 if args.quick else 3
    for th in (1, ncores):
        r = time_emulator(frames, th, reps=reps)
        print("emulator", r, flush=True)
        out["emulator"]["runs"].append(r)
    for th, B, rp in ((ncores, 1, reps), (ncores, 8, reps), (1, 1, 1)):
        r = time_slomo(B, 10, th, reps=rp)
        print("slomo", r, flush=True)
        out["slomo"]["runs"].append(r)
    import subprocess
    try:
        out["host"]["repo_commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        out["host"]["repo_commit"] = None
    out["host"]["recorded"] = time.strftime("%Y-%m-%d %H:%M:%S")
    out["host"]["hostname"] = platform.node()
    path = os.path.join(ROOT, "profiles", "r06_cpu_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
