#!/usr/bin/env python
"""dev tool: k_main per frame (default at this size) vs k_step chain vs unfused pipeline at 1280x720 noisy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
H, W, F = 720, 1280, 40
fr = B.gen_frames_device(F + 1, 4, dev, h=H, w=W)
for label, ug in (("k_main graph", 33), ("k_step chain graph", 65), ("legacy graph", 17)):
    emu = EventEmulator(device=dev, seed=4, rng_mode="philox", **B.DEFAULT_KW)
    emu.set_dvs_params("noisy")
    dt = 1 / 600.0
    emu.generate_events(fr[0], 0.0)
    buf = fr[1:].contiguous()
    cap = 16_000_000
    emu.generate_events_batch(buf, [(1 + i) * dt for i in range(F)], return_device=True, cap=cap, use_graph=ug)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for k in range(1, 4):
        ev, c = emu.generate_events_batch(buf, [(1 + k * F + i) * dt for i in range(F)], return_device=True, cap=cap, use_graph=ug)
        n += int(c.sum())
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    print("%-14s %8.2f us/frame %8.1f Mev/s" % (label, sec / (3 * F) * 1e6, n / sec / 1e6))
