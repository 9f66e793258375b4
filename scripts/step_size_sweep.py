#!/usr/bin/env python
"""dev tool: time per step of the benchmark loop against frames per step: t(F) = a + b F separates the per-run boundary (a) from the
steady-state time per frame (b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
from v2e_amd.benchutil import run_steps
dev = torch.device("cuda")
frames = B.gen_frames_device(7201, 1, dev)
res = []
for F in (96, 160, 300, 608, 1216, 2400):
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    steps = max(6, 12000 // F)
    best = None
    for rep in range(3):
        el, ne = run_steps(emu, frames, F, B.DT, steps, 3 if rep == 0 else 0, None, None, dev, first_step=rep * steps + (3 if rep else 0))
        us = el / steps * 1e6
        best = us if best is None else min(best, us)
    res.append((F, best))
    print("F = %4d frames/step: %8.1f us/step = %.3f us/frame" % (F, best, best / F), flush=True)
x = np.array([r[0] for r in res], float); y = np.array([r[1] for r in res])
b, a = np.polyfit(x, y, 1)
print("fit: t(F) = %.1f us + %.4f us x F  (boundary %.1f us per run; steady state %.3f us/frame = %.2f Gev/s at 35.4 k events/frame)" % (a, b, a, b, 35.4e3 / b / 1e3))
