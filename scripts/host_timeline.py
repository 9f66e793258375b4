#!/usr/bin/env python
"""dev tool: where the HOST's time goes in the pipelined loop (bench.py's run_steps): enqueue of run s + 1, result() of run s (blocked +
bookkeeping).  A loop whose result() never blocks is bound by the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from v2e_amd import EventEmulator
dev = torch.device("cuda", 0); F = 300; N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames = bench.gen_frames_device(bench.CLIP_STEPS * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **bench.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
te, tr, tw = [], [], []
pend = None
torch.cuda.synchronize()
t_all = time.perf_counter()
for s in range(N):
    lo = 1 + (s % bench.CLIP_STEPS) * F
    t0 = time.perf_counter()
    p = emu.generate_events_batch_async(frames[lo:lo + F], [(1 + s * F + i) * bench.DT for i in range(F)], return_device=True, use_graph=0, frames_resident=True)
    t1 = time.perf_counter()
    if pend is not None:
        eng = emu._engine
        eng.run_wait(pend.ticket)
        t2 = time.perf_counter()
        pend.result()
        t3 = time.perf_counter()
        tw.append(t2 - t1); tr.append(t3 - t2)
    te.append(t1 - t0)
    pend = p
pend.result()
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
sk = 5
print("steps %d: %.1f us/step; host per step: enqueue %.1f us (p90 %.1f), blocked waiting for the run before %.1f us, result() bookkeeping %.1f us" % (
    N, tot / N * 1e6, np.mean(te[sk:]) * 1e6, np.percentile(te[sk:], 90) * 1e6, np.mean(tw[sk:]) * 1e6, np.mean(tr[sk:]) * 1e6))
# ---- what the enqueue is made of
import cProfile, pstats, io
pr = cProfile.Profile()
pend = []
pr.enable()
for s in range(N, N + 30):
    lo = 1 + (s % bench.CLIP_STEPS) * F
    pend.append(emu.generate_events_batch_async(frames[lo:lo + F], [(1 + s * F + i) * bench.DT for i in range(F)], return_device=True, use_graph=0, frames_resident=True))
    if len(pend) > 1:
        pend.pop(0).result()
pend.pop(0).result()
pr.disable()
so = io.StringIO()
pstats.Stats(pr, stream=so).sort_stats("cumulative").print_stats(22)
print(so.getvalue()[:4500])
