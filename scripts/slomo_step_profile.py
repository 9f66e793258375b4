#!/usr/bin/env python
"""dev tool: one SloMo batch (flow UNet on B pairs, prep, interpolation UNet on U*B samples, fuse) a few times, for
rocprofv3 --kernel-trace --stats: where the time outside the interpolation UNet goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.benchutil import slomo_bench
r = slomo_bench(torch.device("cuda"), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
print(r["value"], r["ms_per_batch"])
