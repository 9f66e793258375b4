#!/usr/bin/env python
"""dev tool: short runs of the three emulator workloads of the bench line, for rocprofv3 (kernel trace / PMC passes).
usage: emu_workloads.py [headline] [batched] [hd]   (default: all three)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as B  # noqa: E402
from v2e_amd import EventEmulator  # noqa: E402
from v2e_amd.benchutil import batched_emulator_bench, hd_noisy_emulator_bench  # noqa: E402

which = sys.argv[1:] or ["headline", "batched", "hd"]
dev = torch.device("cuda")
if "headline" in which:
    F, steps = B.FRAMES_PER_STEP, 4
    frames = B.gen_frames_device(steps * F + 1, 1, dev)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    n = 0
    for s in range(steps):
        lo = 1 + s * F
        ev, c = emu.generate_events_batch(frames[lo:lo + F].contiguous(), [(lo + i) * B.DT for i in range(F)],
                                          return_device=True, use_graph=True)
        n += int(c.sum())
    torch.cuda.synchronize()
    print("headline: %d events in %d frames" % (n, steps * F))
if "batched" in which:
    print("batched:", batched_emulator_bench(dev))
if "hd" in which:
    print("hd:", hd_noisy_emulator_bench(dev))
