#!/usr/bin/env python
"""Reproduction of the round-4 finding behind -fno-slp-vectorize in v2e_amd/csrc/Makefile.

A float32 UNet pass (victim, current stream) runs while ONE split-operand convolution (bf16 / f16 MFMA) is launched 40 times
on another stream.  With the library built WITH packed-float32 VALU instructions (make -B EXTRA=-fslp-vectorize) the victim's
result differs from its stand-alone result in 10 runs of 10 -- by ~2e-3 beside the bf16 MFMA kernel, ~5e-4 beside the f16 one --
although victim and noise share no memory; the kernel that goes wrong is k_upsample2 (24 v_pk_mul/add_f32), chains of
convolutions alone are fine, and so is the whole net with the upsampling fused into the convolutions' loaders (V2E_AMD_FUSE_UP=31),
with AMD_SERIALIZE_KERNEL=3, with GPU_MAX_HW_QUEUES=1, or with a float32-MFMA convolution as the noise.  Built as committed
(no packed-float32 instructions) every line prints 0/10."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
dev = torch.device("cuda")
sd_i = {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}
side = torch.cuda.Stream(dev)
vnet = HipUNet(sd_i, 12, 5, dev, "f32")
nz = HipUNet(sd_i, 12, 5, dev, "bf16x3")
lib = nz.lib
def P(t): return C.c_void_p(t.data_ptr())
d = nz.descs[9]
xn = (torch.rand((2, d.cin, 16, 24), device=dev) - 0.4); yn = torch.empty((2, d.cout, 16, 24), device=dev)
def noise():
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.v2e_conv2d_lrelu(P(xn), d.cin, None, 0, 0, C.byref(d), P(yn), 2, 16, 24, st) == 0
def chain(layers, n, h, w):
    """dependent convs: layer i reads the output of layer i-1 (cin must match cout of the one before)"""
    ds = [vnet.descs[i] for i in layers]
    x = torch.rand((n, ds[0].cin, h, w), device=dev) - 0.4
    bufs = [torch.empty((n, q.cout, h, w), device=dev) for q in ds]
    def run():
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        src = x
        for q, y in zip(ds, bufs):
            assert lib.v2e_conv2d_lrelu(P(src), q.cin, None, 0, 0, C.byref(q), P(y), n, h, w, st) == 0
            src = y
        return bufs[-1]
    return run
xi = torch.rand((4, 12, 64, 96), device=dev) - 0.4
victims = {
    "whole f32 net": lambda: vnet.forward(xi),
    "chain conv1->conv2 @64x96": chain([0, 1], 4, 64, 96),
    "chain 3,3,3,3 (64->64 k5) @32x48": chain([3, 3, 3, 3], 4, 32, 48),
    "chain 5x6 (128->128 k3) @16x24": chain([5] * 6, 4, 16, 24),
    "chain 11x8 (512->512 k3) @2x3": chain([11] * 8, 4, 2, 3),
    "chain 11x8 (512->512 k3) @4x6": chain([11] * 8, 4, 4, 6),
    "chain 11x8 (512->512 k3) @8x12": chain([11] * 8, 4, 8, 12),
    "chain 7x8 (256->256 k3) @8x12": chain([7] * 8, 4, 8, 12),
}
for name, vf in victims.items():
    r = vf().clone(); torch.cuda.synchronize()
    bad = 0
    for rep in range(10):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(40): noise()
        o = vf().clone()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o, r))
    print("victim %-45s wrong %d/10" % (name, bad), flush=True)
