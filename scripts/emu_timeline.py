#!/usr/bin/env python
"""dev tool: in-kernel timeline (s_memrealtime stamps, 100 MHz) of one mid-run launch: k_step of the default
pipeline, or k_main of the fused one (argument fused=1)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import gen_frames_device, DEFAULT_KW, H, W, DT
from v2e_amd import EventEmulator, _capi

kw = dict(DEFAULT_KW)
fused = False
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "fused":
        fused = bool(int(v))
    else:
        kw[k] = float(v)
dev = torch.device("cuda")
F = 100
frames = gen_frames_device(2 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **kw)
emu.generate_events(frames[0], 0.0)
lib = C.CDLL(_capi.LIB_PATH)
ng = C.c_int()
lib.v2e_emu_debug_timeline(emu._engine._h, None, C.byref(ng))
for s in range(2):
    lo = 1 + s * F
    emu.generate_events_batch(frames[lo:lo + F].contiguous(), [(lo + i) * DT for i in range(F)], return_device=True, use_graph=33 if fused else 1)
out = (C.c_uint64 * (16 * ng.value))()
lib.v2e_emu_debug_timeline(emu._engine._h, out, None)
t = np.frombuffer(out, dtype=np.uint64).reshape(ng.value, 16).astype(np.int64)
t0 = t[:, 0].min()
if not fused:
    print("blocks start spread: %.2f us" % ((t[:, 0].max() - t0) / 100.0))
    for i, n in [(0, "start"), (5, "RNG done"), (6, "loads arrived"), (1, "M(f-1) known"), (2, "finalise done"), (3, "count math done"), (4, "end")]:
        col = t[:, i]
        ok = col > 0
        if not ok.any():
            continue
        print("%-18s mean %+7.2f us  (min %+6.2f max %+6.2f) since first block start; mean since own start %6.2f" % (
            n, (col[ok] - t0).mean() / 100.0, (col[ok] - t0).min() / 100.0, (col[ok] - t0).max() / 100.0, (col[ok] - t[ok, 0]).mean() / 100.0))
    sys.exit(0)
names = ["start", "loads issued+keytotals", "M known", "pass1 done", "barrier", "pass2 done", "emit loop end", "emit done", "count math done", "end"]
print("blocks start spread: %.2f us" % ((t[:, 0].max() - t0) / 100.0))
names += ["tg ready (10)", "emit setup done (11)", "late key totals done (12)"]
order = [0, 1, 2, 10, 11, 12, 3, 4, 5, 6, 7, 8, 9]
for i, n in [(k, names[k]) for k in order]:
    col = t[:, i]
    ok = col > 0
    print("%-26s mean %+7.2f us  (min %+6.2f max %+6.2f) since first block start; mean since own start %6.2f" % (
        n, (col[ok] - t0).mean() / 100.0, (col[ok] - t0).min() / 100.0, (col[ok] - t0).max() / 100.0, (col[ok] - t[ok, 0]).mean() / 100.0))
