#!/usr/bin/env python
"""dev tool: instrumented run of the decoupled pipeline: step-chain time per launch and emission-batch time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import gen_frames_device, DEFAULT_KW, H, W, DT
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = 300
frames = gen_frames_device(3 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
for s in range(3):
    lo = 1 + s * F
    buf = frames[lo:lo + F].contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev, c = emu.generate_events_batch(buf, [(lo + i) * DT for i in range(F)], return_device=True, use_graph=2)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pr = emu._engine.last_profile()
    print("run %d: wall %.2f ms; step chain %.3f ms = %.2f us/frame (%d launches); emission batches: %.3f ms total in %d" % (
        s, wall * 1e3, pr["count"], pr["count"] / F * 1e3, pr["step_launches"], pr["emit"], pr["emit_batches"]))
