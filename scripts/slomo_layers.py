#!/usr/bin/env python
"""dev tool: run the interpolation UNet a few times (for rocprofv3 --kernel-trace).  usage: slomo_layers.py [n [H W]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 320)
dev = torch.device("cuda")
net = HipUNet({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}, 12, 5, dev,
              os.environ.get("V2E_AMD_CONV_MATH", "bf16x3"))
x = torch.rand((n, 12, H, W), device=dev) - 0.4
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
print("done")
