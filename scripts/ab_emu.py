#!/usr/bin/env python
"""dev tool: the batched (64 clips) and 1280x720 noisy emulator legs of the bench line, for A/B runs (V2E_AMD_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.benchutil import batched_emulator_bench, hd_noisy_emulator_bench
dev = torch.device("cuda")
for _ in range(2):
    b = batched_emulator_bench(dev)
    h = hd_noisy_emulator_bench(dev)
    print(os.environ.get("TAG", ""), "batched %.1f (%.3f)  hd %.1f (%.3f)" % (b["value"], b["hbm_frac"], h["value"], h["hbm_frac"]))
