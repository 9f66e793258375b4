#!/usr/bin/env python
"""dev tool: wall time of the interpolation UNet forward at n samples (HIP events).  usage: slomo_time.py [n] [conv_math ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
from v2e_amd.benchutil import unet_flops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
maths = sys.argv[2:] or ["auto", "fp16x2", "bf16x3"]
dev = torch.device("cuda")
x = torch.rand((n, 12, 256, 320), device=dev) - 0.4
for m in maths:
    net = HipUNet({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}, 12, 5, dev, m)
    for _ in range(2): net.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    IT = int(os.environ.get('ITERS', '5'))
    for _ in range(IT): net.forward(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / IT
    print("%s %s n=%d: %.3f ms, %.1f TF f32-equivalent" % (os.environ.get("TAG", ""), m, n, ms, n * unet_flops(12, 5, 256, 320) / ms / 1e9))
