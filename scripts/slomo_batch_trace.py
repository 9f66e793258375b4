#!/usr/bin/env python
"""dev tool: a few SloMoEngine.interpolate batches as bench.py's slomo leg runs them (B = 8 pairs, U = 10, 320x256, look-ahead), for
rocprofv3 --kernel-trace (scripts/dump_timeline.py prints the launches with their hardware queues)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.slomo import SloMoEngine
from v2e_amd.synth import portable_unet_state_dict
dev = torch.device("cuda")
sd_f, sd_i = portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102)
eng = SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()}, {k: torch.from_numpy(v) for k, v in sd_i.items()}, dev)
g = torch.Generator(device=dev); g.manual_seed(2)
B, U, H, W = 8, 10, 256, 320
I0 = torch.rand((B, 1, H, W), device=dev, generator=g) - 0.428
I1 = torch.rand((B, 1, H, W), device=dev, generator=g) - 0.428
ts = [(k + 0.5) / U for k in range(U)]
for _ in range(2): eng.interpolate(I0, I1, ts, next_pair=(I0, I1))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): eng.interpolate(I0, I1, ts, next_pair=(I0, I1))
e1.record(); torch.cuda.synchronize()
print("ms per batch %.3f" % (e0.elapsed_time(e1) / 4))
