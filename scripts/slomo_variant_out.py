#!/usr/bin/env python
"""dev tool: the interpolation UNet's output under the current V2E_AMD_S3_VARIANT, as a digest (to compare tile variants bit for bit)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
dev = torch.device("cuda")
net = HipUNet({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}, 12, 5, dev)
g = torch.Generator(device="cpu").manual_seed(3)
for shape in ((3, 12, 256, 320), (2, 12, 64, 96), (1, 12, 128, 160)):
    x = (torch.rand(shape, generator=g) - 0.4).to(dev)
    y = net.forward(x).cpu().numpy()
    print(os.environ.get("V2E_AMD_S3_VARIANT", "0"), shape, hashlib.sha256(y.tobytes()).hexdigest()[:16], float(abs(y).max()))
