#!/usr/bin/env python
"""dev tool: is the HOST the bottleneck of bench.py's step loop?  Times run_steps' pieces with perf_counter around each call
(enqueue: buf.copy_, generate_events_batch_async; finish: result()) and prints a cProfile of 60 steps."""
import cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
from v2e_amd.benchutil import run_steps
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
frames = B.gen_frames_device(12 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
run_steps(emu, frames, F, B.DT, 20, 5, None, None, dev)
# manual loop with timers
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
t_copy = t_async = t_res = 0.0
pend = None
torch.cuda.synchronize()
T0 = time.perf_counter()
N = 60
for s in range(25, 25 + N):
    a = time.perf_counter()
    lo = 1 + (s % 12) * F
    buf.copy_(frames[lo:lo + F])
    b = time.perf_counter()
    nxt = emu.generate_events_batch_async(buf, [(1 + s * F + i) * B.DT for i in range(F)], return_device=True, use_graph=1)
    c = time.perf_counter()
    if pend is not None:
        pend.result()
    d = time.perf_counter()
    pend = nxt
    t_copy += b - a; t_async += c - b; t_res += d - c
pend.result()
torch.cuda.synchronize()
T1 = time.perf_counter()
print("per step: wall %.0f us; host: buf.copy_ %.0f us, generate_events_batch_async %.0f us, result() (incl. waiting) %.0f us"
      % ((T1 - T0) / N * 1e6, t_copy / N * 1e6, t_async / N * 1e6, t_res / N * 1e6))
pr = cProfile.Profile()
pr.enable()
run_steps(emu, frames, F, B.DT, 60, 0, None, None, dev, first_step=100)
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
print(st.getvalue()[:6000])
