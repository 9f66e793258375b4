#!/usr/bin/env python
"""dev tool: the pipelined step loop (benchutil.run_steps) repeated on fresh emulators; every repetition must produce the same per-step
event counts and digests (refr = 4 ms: the pipeline switches to one frame per launch between steps).  usage: stress_steps.py [reps] [refr]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench as B
from v2e_amd import EventEmulator
from v2e_amd.benchutil import run_steps


class Sink:
    def __init__(self): self.steps = []
    def submit(self, ev, n, ready_event=None, run_bound=None):
        self.steps.append((int(n), hashlib.sha256(ev[:n].cpu().numpy().tobytes()).hexdigest()[:12]))
    def wait(self): pass


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
refr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
dev = torch.device("cuda")
kw = dict(B.DEFAULT_KW); kw["refractory_period_s"] = refr
frames = B.gen_frames_device(2 * B.FRAMES_PER_STEP + 1, 1, dev)
# (other allocations on the device, as in a long test session)
junk = [torch.empty((1 << 28,), dtype=torch.uint8, device=dev) for _ in range(4)]
ref = None
bad = 0
for r in range(reps):
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **kw)
    emu.generate_events(frames[0], 0.0)
    sink = Sink()
    run_steps(emu, frames, B.FRAMES_PER_STEP, B.DT, 5, 1, sink, None, dev)
    if ref is None:
        ref = sink.steps
    elif sink.steps != ref:
        bad += 1
        print("repetition %d differs:" % r, [(i, a[0], b[0]) for i, (a, b) in enumerate(zip(sink.steps, ref)) if a != b])
print("refr %g: %d repetitions, %d differing; counts %s" % (refr, reps, bad, [s[0] for s in ref]))
