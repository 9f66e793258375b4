#!/usr/bin/env python
"""dev tool: GPU-side busy time and idle gap per step of the benchmark loop, from HIP events on the run's stream (no profiler):
e0[s] recorded before step s is enqueued, e1[s] behind it.  busy = e0[s] -> e1[s]; gap = e1[s] -> e0[s+1]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
frames = B.gen_frames_device(24 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
stage = os.environ.get("V2E_AMD_BENCH_STAGE_FRAMES") == "1"
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
e0 = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
e1 = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
import gc; gc.collect(); gc.freeze()
pend = None
h_async = h_res = 0.0
torch.cuda.synchronize()
T0 = time.perf_counter()
for s in range(N):
    lo = 1 + (s % 24) * F
    e0[s].record()
    src = frames[lo:lo + F]
    if stage:
        buf.copy_(src); src = buf
    a = time.perf_counter()
    nxt = emu.generate_events_batch_async(src, [(1 + s * F + i) * B.DT for i in range(F)], return_device=True, use_graph=1)
    b = time.perf_counter()
    e1[s].record()
    if pend is not None:
        pend.result()
    c = time.perf_counter()
    pend = nxt
    h_async += b - a; h_res += c - b
pend.result()
torch.cuda.synchronize()
T1 = time.perf_counter()
busy = np.array([e0[s].elapsed_time(e1[s]) for s in range(10, N)]) * 1e3
gap = np.array([e1[s].elapsed_time(e0[s + 1]) for s in range(10, N - 1)]) * 1e3
print("%s: wall/step %.1f us | GPU busy/step mean %.1f p50 %.1f p10 %.1f p90 %.1f | GPU gap mean %.1f p50 %.1f p90 %.1f | host async %.0f result %.0f"
      % ("staged" if stage else "inplace", (T1 - T0) / N * 1e6, busy.mean(), np.median(busy), np.percentile(busy, 10), np.percentile(busy, 90),
         gap.mean(), np.median(gap), np.percentile(gap, 90), h_async / N * 1e6, h_res / N * 1e6))
