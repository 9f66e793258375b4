#!/usr/bin/env python
"""dev tool: where a step's time goes ON THE GPU without a profiler attached.  Sequence-numbered device time stamps (s_memrealtime,
100 MHz): B = a one-thread kernel on the run's stream before the step is enqueued, Z = the first node of the run's graph, J = its
last node (behind the joins), A = a one-thread kernel behind the graph launch.  Per step: B->Z upload + graph start latency, Z->J the
graph, J->A its completion latency, A->B' between steps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device("cuda")
N = 160
st = torch.zeros((8 * N + 16,), dtype=torch.int64, device=dev)
os.environ["V2E_AMD_DBG_STAMP_PTR"] = str(st.data_ptr())
import bench as B
from v2e_amd import EventEmulator
rl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "slp_repro", "libslp_repro.so"))
F = int(os.environ.get("F", B.FRAMES_PER_STEP))
frames = B.gen_frames_device(24 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
def stamp(): assert rl.slp_stamp_seq(C.c_void_p(st.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
import gc; gc.collect(); gc.freeze()
pend = None
for s in range(N):
    lo = 1 + (s % 24) * F
    stamp()
    nxt = emu.generate_events_batch_async(frames[lo:lo + F], [(1 + s * F + i) * B.DT for i in range(F)], return_device=True, use_graph=1)
    stamp()
    if pend is not None: pend.result()
    pend = nxt
pend.result(); torch.cuda.synchronize()
t = st.cpu().numpy()
n = int(t[0]); v = t[1:1 + n].astype(np.float64) / 100.0
assert n == 4 * N, n
v = v.reshape(N, 4)[20:]
Bs, Z, J, A = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
def q(x): return "mean %.1f p50 %.1f p10 %.1f p90 %.1f" % (x.mean(), np.median(x), np.percentile(x, 10), np.percentile(x, 90))
print("F=%d  period %s" % (F, q(A[1:] - A[:-1])))
print("  B->Z upload + graph start: %s" % q(Z - Bs))
print("  Z->J the graph:            %s" % q(J - Z))
print("  J->A completion:           %s" % q(A - J))
print("  A->B' between steps:       %s" % q(Bs[1:] - A[:-1]))
