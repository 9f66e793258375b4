#!/usr/bin/env python
"""Within-process A/B of emulator pipeline variants on the headline workload (dev tool).
use_graph bits: 1 = hipGraph; |16 = 4-kernel legacy; |32 = fused k_main per frame; default = k_step chain + deferred emission."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import gen_frames_device, DEFAULT_KW, H, W, DT
from v2e_amd import EventEmulator

def run(label, kw, use_graph, F=300, steps=6, keep=False, **ekw):
    dev = torch.device("cuda")
    frames = gen_frames_device(F * (steps + 1) + 1, 1, dev)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **kw, **ekw)
    emu.generate_events(frames[0], 0.0)
    buf = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    kept = []
    def step(s):
        lo = 1 + s * F
        buf.copy_(frames[lo:lo + F])
        ev, c = emu.generate_events_batch(buf, [(lo + i) * DT for i in range(F)], return_device=True, use_graph=use_graph)
        if keep and s < 2:
            kept.append((ev.clone(), c.copy()))
        return int(c.sum())
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for s in range(1, steps + 1):
        n += step(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %7.2f us/frame  %8.1f Mev/s" % (label, dt / (steps * F) * 1e6, n / dt / 1e6), flush=True)
    return kept

def same(a, b):
    return all(np.array_equal(x[1], y[1]) and torch.equal(x[0], y[0]) for x, y in zip(a, b))

if __name__ == "__main__":
    kw = dict(DEFAULT_KW)
    a = run("step chain x2 + deferred emission, graph", kw, 1, keep=True)
    b = run("fused k_main per frame, graph", kw, 33, keep=True)
    print("   identical event streams:", same(a, b))
    c = run("step chain x1, graph", kw, 129, keep=True)
    print("   identical event streams:", same(a, c))
    run("step chain x2, plain launches", kw, 0)
    run("legacy 4-kernel graph", kw, 17)
    k2 = dict(kw); k2["refractory_period_s"] = 0.0
    run("step chain, refractory 0", k2, 1)
    k3 = dict(k2); k3["leak_rate_hz"] = 0.0; k3["shot_noise_rate_hz"] = 0.0
    run("step chain, no refr/leak/shot (no Philox)", k3, 1)
    k4 = dict(kw); k4["refractory_period_s"] = 0.002  # rule active on most frames
    run("step chain x1, refractory 2 ms (rule active)", k4, 129)
    a = run("step chain x2, refractory 2 ms (rule active)", k4, 1, keep=True)
    b = run("fused k_main, refractory 2 ms", k4, 33, keep=True)
    print("   identical event streams:", same(a, b))
