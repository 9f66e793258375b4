#!/usr/bin/env python
"""dev tool: cProfile of the frame-at-a-time drop-in API (tape mode by default)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
mode = sys.argv[1] if len(sys.argv) > 1 else "tape"
fr = B.gen_frames_device(401, 1, torch.device("cuda")).cpu().numpy()
emu = EventEmulator(device="cuda", seed=1, rng_mode=mode, **B.DEFAULT_KW)
for i in range(30):
    emu.generate_events(fr[i], i * B.DT)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(30, 400):
    emu.generate_events(fr[i], i * B.DT)
pr.disable()
dt = time.perf_counter() - t0
print("%s: %.1f frames/s (%.0f us/frame under cProfile)" % (mode, 370 / dt, dt / 370 * 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
