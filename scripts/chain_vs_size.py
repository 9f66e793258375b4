#!/usr/bin/env python
"""dev tool: clean k_chain launch time against the number of workgroups (is the chain bound by the SIMDs that hold two waves?).
Instrumented runs (every kernel alone on one stream); prints the per-launch times of a 300-frame step for several sensor sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
sizes = [(128, 256), (256, 256), (260, 346), (256, 512), (384, 512), (512, 512), (512, 768), (512, 1024)]
for (h, w) in sizes:
    frames = B.gen_frames_device(2 * F + 1, 1, dev, h=h, w=w)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    emu.generate_events_batch(frames[1:1 + F], [(1 + i) * B.DT for i in range(F)], return_device=True)
    eng, P = emu._engine, emu._params()
    buf = frames[1 + F:1 + 2 * F].contiguous()
    t_prev = [emu.t_previous + i * B.DT for i in range(F)]
    t_frame = [emu.t_previous + (i + 1) * B.DT for i in range(F)]
    ev = eng.event_buffer(1)
    recs = eng.alloc_recs(F)
    eng.run(P, buf, t_prev, t_frame, emu.frame_counter, ev, recs, use_graph=2)
    prof = eng.last_profile()
    kname, fpl, fpb = eng.last_pipeline()
    us = prof.get("chain_launch_us", [])
    full = sorted(us[:F // fpl])
    print("%4dx%-4d wgs %5d  %s  min %.1f med %.1f  all %s" % (h, w, (h * w + 255) // 256, kname.split("(")[0], full[0], full[len(full) // 2], [round(u, 1) for u in us]), flush=True)
    del emu, eng, frames, buf, ev, recs
    torch.cuda.empty_cache()
