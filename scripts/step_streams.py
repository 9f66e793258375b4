#!/usr/bin/env python
"""dev tool: does the idle time between the hipGraphs of consecutive steps (~130 us in the kernel trace) go away when the steps are
launched on two alternating streams joined by an event instead of back to back on one stream?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = sys.argv[2] if len(sys.argv) > 2 else "one"
frames_all = B.gen_frames_device(24 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames_all[0], 0.0)
bufs = [torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
prev_done = None
def enq(s):
    global prev_done
    lo = 1 + (s % 24) * F
    st = streams[s % 2] if mode == "two" else streams[0]
    with torch.cuda.stream(st):
        if prev_done is not None and mode == "two":
            st.wait_event(prev_done)
        buf = bufs[s % 2] if mode == "two" else bufs[0]
        buf.copy_(frames_all[lo:lo + F])
        p = emu.generate_events_batch_async(buf, [(1 + s * F + i) * B.DT for i in range(F)], return_device=True)
    prev_done = p.done
    return p
gc.collect(); gc.freeze()
pend = enq(0)
for s in range(1, 20):
    nxt = enq(s); pend.result(); pend = nxt
torch.cuda.synchronize()
t0 = time.perf_counter(); ne = 0
for s in range(20, 20 + n):
    nxt = enq(s); ev, c = pend.result(); ne += int(c.sum()); pend = nxt
pend.result(); torch.cuda.synchronize()
sec = time.perf_counter() - t0
print("%s stream(s): %.4f ms per step, %.1f Mev/s" % (mode, sec / n * 1e3, ne / sec / 1e6))
