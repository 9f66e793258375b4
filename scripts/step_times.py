#!/usr/bin/env python
"""dev tool: host-side completion time of every step of the headline loop (where do slow blocks come from?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
frames_all = B.gen_frames_device(24 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames_all[0], 0.0)
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
def enq(s):
    lo = 1 + (s % 24) * F
    buf.copy_(frames_all[lo:lo + F])
    return emu.generate_events_batch_async(buf, [(1 + s * F + i) * B.DT for i in range(F)], return_device=True)
import gc
if os.environ.get('NOGC'): gc.collect(); gc.disable()
ts, ne = [], []
pend = enq(0)
t0 = time.perf_counter()
for s in range(1, n):
    nxt = enq(s)
    ev, c = pend.result()
    ts.append(time.perf_counter()); ne.append(int(c.sum()))
    pend = nxt
d = np.diff(np.array(ts)) * 1e3
print("median %.3f ms; steps slower than 1.5x median: %s" % (np.median(d), [(i + 1, round(float(v), 2)) for i, v in enumerate(d) if v > 1.5 * np.median(d)][:60]))
print("per 25 steps mean ms:", [round(float(d[i:i + 25].mean()), 3) for i in range(0, len(d), 25)])
