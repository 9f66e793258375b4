#!/usr/bin/env python
"""dev tool: the chain kernel's launches as they run in the TIMED configuration (graph replay, side streams beside it), from the
device time stamps of v2e_emu_launch_stamps -- no profiler attached.  Usage: chain_stamps.py [steps] [show_runs]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from v2e_amd import EventEmulator
from v2e_amd.benchutil import run_steps


def read_stamps(eng, runs):
    n, lpr = C.c_int(), C.c_int()
    buf = np.zeros((runs, 128, 2), dtype=np.uint64)
    rc = eng.lib.v2e_emu_launch_stamps(eng._h, runs, buf.ctypes.data_as(C.c_void_p), runs, C.byref(n), C.byref(lpr))
    assert rc == 0 and lpr.value == 128
    return buf[:n.value]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    F = bench.FRAMES_PER_STEP
    frames = bench.gen_frames_device(min(steps + 5, bench.CLIP_STEPS) * F + 1, 1, dev)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **bench.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    eng = emu._engine
    el, ne = run_steps(emu, frames, F, bench.DT, steps, 5, None, None, dev)
    print("plain: %.1f Mev/s, %.1f us/step" % (ne / el / 1e6, el / steps * 1e6))
    assert eng.lib.v2e_emu_launch_stamps(eng._h, steps, None, 0, None, None) == 0
    el, ne = run_steps(emu, frames, F, bench.DT, steps, 0, None, None, dev, first_step=steps + 5)
    print("stamped: %.1f Mev/s, %.1f us/step" % (ne / el / 1e6, el / steps * 1e6))
    st = read_stamps(eng, steps).astype(np.int64)
    nl = int((st[0, :, 1] > 0).sum())
    st = st[:, :nl] / 1e3  # us
    dur = st[:, :, 1] - st[:, :, 0]
    gap = st[:, 1:, 0] - st[:, :-1, 1]
    step_gap = st[1:, 0, 0] - st[:-1, -1, 1]   # last launch's end -> next run's first launch's start
    period = st[1:, 0, 0] - st[:-1, 0, 0]
    print("launches per run %d; step period mean %.1f us (p50 %.1f)" % (nl, period.mean(), np.median(period)))
    print("per run: sum of launch durations %.1f, gaps inside the run %.1f, last end -> next run's first start %.1f" % (
        dur.sum(1).mean(), gap.sum(1).mean(), step_gap.mean()))
    print("duration by launch index (mean):", " ".join("%.1f" % v for v in dur.mean(0)))
    print("gap before launch index (mean):  - " + " ".join("%.1f" % v for v in gap.mean(0)))
    full = dur[:, :nl - 2].ravel()
    print("full launches: mean %.2f p10 %.2f p50 %.2f p90 %.2f; all launches mean %.2f" % (
        full.mean(), np.percentile(full, 10), np.median(full), np.percentile(full, 90), dur.mean()))
    for r in range(len(st) - show, len(st)):
        t0 = st[r, 0, 0]
        print("run %d:" % r, " ".join("[%.0f +%.0f]" % (a - t0, b - a) for a, b in st[r]))


if __name__ == "__main__":
    main()
