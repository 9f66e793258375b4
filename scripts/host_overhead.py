#!/usr/bin/env python
"""dev tool: where the wall time of one generate_events_batch step goes (host preparation vs GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F, steps = B.FRAMES_PER_STEP, 10
frames = B.gen_frames_device(steps * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
eng = emu._engine
for s in range(steps):
    lo = 1 + s * F
    buf.copy_(frames[lo:lo + F])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    times = [(lo + i) * B.DT for i in range(F)]
    t1 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    P = emu._params()
    t_prev = [emu.t_previous] + times[:-1]
    ev = eng.event_buffer(max(4 * B.H * B.W, 1 << 16) * 64)
    recs = eng.alloc_recs(F)
    t2 = time.perf_counter()
    eng.run(P, buf, t_prev, times, emu.frame_counter, ev, recs, use_graph=1)
    t3 = time.perf_counter()
    e1.record()
    r = eng.recs_to_numpy(recs)[:, 0]
    t4 = time.perf_counter()
    emu.frame_counter += F; emu.t_previous = times[-1]
    if s >= 2:
        print("step %d: times list %.0f us, params/buffers %.0f us, eng.run call %.0f us, wait+readback %.0f us; GPU %.0f us (e0->e1)" % (
            s, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, e0.elapsed_time(e1) * 1e3))
