#!/usr/bin/env python
"""dev tool: in-kernel timeline of one mid-run k_chain launch (s_memrealtime stamps, 100 MHz)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import gen_frames_device, DEFAULT_KW, H, W, DT
from v2e_amd import EventEmulator, _capi
dev = torch.device("cuda")
F = 160
frames = gen_frames_device(2 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
lib = C.CDLL(_capi.LIB_PATH)
ng = C.c_int()
lib.v2e_emu_debug_timeline(emu._engine._h, None, C.byref(ng))
buf = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
for s in range(2):
    lo = 1 + s * F
    buf.copy_(frames[lo:lo + F])
    emu.generate_events_batch(buf, [(lo + i) * DT for i in range(F)], return_device=True, use_graph=1)
out = (C.c_uint64 * (16 * ng.value))()
lib.v2e_emu_debug_timeline(emu._engine._h, out, None)
t = np.frombuffer(out, dtype=np.uint64).reshape(ng.value, 16).astype(np.int64)
t0 = t[:, 0].min()
print("blocks start spread: %.2f us" % ((t[:, 0].max() - t0) / 100.0))
names = ["start", "state loaded", "pass inputs staged"] + ["frame %d done" % k for k in range(12)] + ["end"]
prev = None
for i, n in enumerate(names):
    col = t[:, i]; ok = col > 0
    if not ok.any(): continue
    m = (col[ok] - t[ok, 0]).mean() / 100.0
    print("%-20s mean since own start %7.2f us   delta %6.2f" % (n, m, m - (prev if prev is not None else 0)))
    prev = m
