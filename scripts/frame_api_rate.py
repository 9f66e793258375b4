#!/usr/bin/env python
"""dev tool: rate of the drop-in frame-at-a-time API (host numpy frame in, numpy events out; PCIe-inclusive)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import DEFAULT_KW
from v2e_amd import EventEmulator
from v2e_amd.synth import sincos_gradient_frames
fr = sincos_gradient_frames(421, 260, 346, seed=1)
for mode in ("philox", "tape"):
    emu = EventEmulator(device="cuda", seed=1, rng_mode=mode, **DEFAULT_KW)
    emu.generate_events(fr[0], 0.0)
    for i in range(1, 21):
        emu.generate_events(fr[i], i / 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for i in range(21, 421):
        e = emu.generate_events(fr[i], i / 300)
        n += 0 if e is None else len(e)
    dt = time.perf_counter() - t0
    print("%-7s frame API: %8.1f frames/s  %7.2f Mev/s  (%.0f us/frame)" % (mode, 400 / dt, n / dt / 1e6, dt / 400 * 1e6))
