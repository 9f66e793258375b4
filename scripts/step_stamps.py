#!/usr/bin/env python
"""dev tool: where a step's time goes ON THE GPU without a profiler attached: one-thread stamp kernels (s_memrealtime, 100 MHz) on the
run's stream before (B) and behind (A) every step of the benchmark loop.  A(s) - B(s) = the step as the stream executes it (upload,
graph launch latency, the run); B(s+1) - A(s) = what lies between two steps (the caller's frame copy when staged, idle)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from v2e_amd import EventEmulator
dev = torch.device("cuda")
rl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "slp_repro", "libslp_repro.so"))
F = int(os.environ.get("F", B.FRAMES_PER_STEP))
frames = B.gen_frames_device(24 * F + 1, 1, dev)
emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
emu.generate_events(frames[0], 0.0)
stage = os.environ.get("V2E_AMD_BENCH_STAGE_FRAMES") == "1"
buf = torch.empty((F, B.H, B.W), dtype=torch.uint8, device=dev)
N = 160
st = torch.zeros((2 * N,), dtype=torch.int64, device=dev)
def stamp(i): assert rl.slp_stamp(C.c_void_p(st.data_ptr()), i, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
import gc; gc.collect(); gc.freeze()
pend = None
for s in range(N):
    lo = 1 + (s % 24) * F
    src = frames[lo:lo + F]
    if stage:
        buf.copy_(src); src = buf
    stamp(2 * s)
    nxt = emu.generate_events_batch_async(src, [(1 + s * F + i) * B.DT for i in range(F)], return_device=True, use_graph=1)
    stamp(2 * s + 1)
    if pend is not None: pend.result()
    pend = nxt
pend.result(); torch.cuda.synchronize()
t = st.cpu().numpy().astype(np.float64) / 100.0  # us
Bs, As = t[0::2], t[1::2]
run = (As - Bs)[20:]; between = (Bs[1:] - As[:-1])[20:]; period = (As[1:] - As[:-1])[20:]
print("%s F=%d: period mean %.1f p50 %.1f | step on the stream (B->A) mean %.1f p50 %.1f p10 %.1f | between steps (A->B) mean %.1f p50 %.1f p90 %.1f"
      % ("staged" if stage else "inplace", F, period.mean(), np.median(period), run.mean(), np.median(run), np.percentile(run, 10), between.mean(), np.median(between), np.percentile(between, 90)))
