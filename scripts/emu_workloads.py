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
    # the TIMED configuration: bench.py's loop (run_steps: pipelined runs, the next run enqueued while this one executes)
    from v2e_amd.benchutil import run_steps  # noqa: E402
    F, steps = B.FRAMES_PER_STEP, 8
    frames = B.gen_frames_device(steps * F + 1, 1, dev)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    el, n = run_steps(emu, frames, F, B.DT, steps, 0, None, None, dev)
    print("headline: %d events in %d frames" % (n, steps * F))
if "batched" in which:
    print("batched:", batched_emulator_bench(dev))
if "hd" in which:
    print("hd:", hd_noisy_emulator_bench(dev))
