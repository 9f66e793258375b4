#!/usr/bin/env python
"""dev tool: headline workload wall time per frame under the current environment switches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F, steps = B.FRAMES_PER_STEP, 12
frames = B.gen_frames_device(steps * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
ug = int(os.environ.get("UG", "1"))
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
def step(s):
    lo = 1 + s * F
    buf.copy_(frames[lo:lo + F])  # fixed buffer: the run's hipGraph bakes the pointer in
    ev, c = emu.generate_events_batch(buf, [(lo + i) * B.DT for i in range(F)], return_device=True, use_graph=ug)
    return int(c.sum())
for s in range(2): step(s)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for s in range(2, steps): n += step(s)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%s: %.3f us/frame, %.0f Mev/s" % (os.environ.get("TAG", ""), dt / ((steps - 2) * F) * 1e6, n / dt / 1e6))
