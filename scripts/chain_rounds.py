#!/usr/bin/env python
"""dev tool: what the redo passes of the chain did in one step of the benchmark clip -- per 32-frame launch the frames its
own pass flagged rule-on (row 0 of its rule-on rows, with the maxima) and the rows the redo passes on it left (row r: what
pass r flagged on the frames after the one it fixed), read back from the device after the run."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
F = B.FRAMES_PER_STEP
step = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames_all = B.gen_frames_device(B.CLIP_STEPS * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames_all[0], 0.0)
for s in range(step + 1):
    lo = 1 + (s % B.CLIP_STEPS) * F
    ev, counts = emu.generate_events_batch(frames_all[lo:lo + F], [(1 + s * F + i) * B.DT for i in range(F)], return_device=True, use_graph=257)
eng = emu._engine
nl, K = C.c_int(), C.c_int()
eng.lib.v2e_emu_debug_chain_rows.restype = C.c_int
eng.lib.v2e_emu_debug_chain_rows(eng._h, None, C.c_size_t(0), C.byref(nl), C.byref(K))
n = nl.value * (K.value + 1) * K.value
buf = (C.c_uint32 * n)()
eng.lib.v2e_emu_debug_chain_rows(eng._h, buf, C.c_size_t(n), C.byref(nl), C.byref(K))
rows = np.frombuffer(buf, np.uint32).reshape(nl.value, K.value + 1, K.value)
print("step %d of the benchmark clip: %d events; rule threshold M >= 7" % (step, int(counts.sum())))
for L in range(nl.value):
    r0 = rows[L, 0]
    if not r0.any() and not rows[L, 1:].any():
        print("launch %2d: no frame flagged" % L); continue
    print("launch %2d: own pass flagged %s" % (L, {int(k): int(v) for k, v in enumerate(r0) if v}))
    for r in range(1, K.value + 1):
        if rows[L, r].any():
            print("            redo pass %d flagged %s" % (r, {int(k): int(v) for k, v in enumerate(rows[L, r]) if v}))
