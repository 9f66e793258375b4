#!/usr/bin/env python
"""Experiment 3: bisect the noise.  Victim = v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] alone (experiment 2: the only packed form that
goes wrong, and only its LOW result).  Noises: mixes of the split-operand convolution's ingredients (scripts/slp_repro/host.hip
k_noise_mix) and the library's convolutions per conv math.  Also dumps what the wrong low halves are.  -> gpurun_out/slp_repro3.*"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
dev = torch.device("cuda")
rl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libslp_repro.so"))
sd_i = {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}
nets = {m: HipUNet(sd_i, 12, 5, dev, m) for m in ("bf16x3", "fp16x2", "f32")}
lib = nets["f32"].lib
def P(t): return C.c_void_p(t.data_ptr())
def ST(): return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
nout = torch.empty((1024 * 256,), device=dev)
gsrc = torch.rand((1 << 20,), device=dev)
MIX = ["bf16 MFMA x4 independent", "bf16 MFMA + v_cvt_pk_bf16_f32", "bf16 MFMA fed by ds_read_b128", "bf16 MFMA + accvgpr write/read",
       "ds_write_b128 + s_barrier + ds_read_b128", "f32 MFMA 32x32x2 x4", "f16 MFMA x4 independent", "bf16 MFMA + global_load stream"]
MIX_ITERS = [60000, 60000, 40000, 60000, 150000, 30000, 60000, 40000]
def conv_noise(math, layer, n, h, w):
    d = nets[math].descs[layer]
    x = torch.rand((n, d.cin, h, w), device=dev) - 0.4
    y = torch.empty((n, d.cout, h, w), device=dev)
    def run():
        for _ in range(60):
            assert lib.v2e_conv2d_lrelu(P(x), d.cin, None, 0, 0, C.byref(d), P(y), n, h, w, ST()) == 0
    return run
noises = [("none", lambda: None)]
for k, nm in enumerate(MIX):
    noises.append((nm, (lambda k=k: rl.slp_launch_noise_mix(k, P(nout), P(gsrc), 1024, MIX_ITERS[k], ST()))))
for math in ("bf16x3", "fp16x2", "f32"):
    noises.append(("conv down4.conv2 %s @16x24" % math, conv_noise(math, 9, 2, 16, 24)))
    noises.append(("conv conv2 k7 32->32 %s @64x96" % math, conv_noise(math, 1, 4, 64, 96)))
g = torch.Generator(device=dev); g.manual_seed(9)
NB = 2048
xv = torch.rand((NB * 256 * 4,), device=dev, generator=g) + 0.25
mism = torch.zeros((2,), dtype=torch.int32, device=dev)
dump = torch.zeros((NB * 256, 8), device=dev)
side = torch.cuda.Stream(dev)
lines, saved = [], {}
for nm, nf in noises:
    tot, still = 0, 0
    for rep in range(4):
        mism.zero_(); dump.zero_()
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            nf()
            ev = torch.cuda.Event(); ev.record()
        for _ in range(6):
            assert rl.slp_launch_victim_dump(P(xv), NB, 3000, P(dump), P(mism), ST()) == 0
        torch.cuda.current_stream().synchronize()
        still += int(not ev.query())
        torch.cuda.synchronize()
        n = int(mism[0].item())
        tot += n
        if n and nm not in saved:
            d = dump.cpu().numpy()
            saved[nm] = d[d[:, 7] > 0][:20000]
    lines.append("noise %-45s wrong results of the swapped v_pk_mul_f32: %10d of %.3g  (noise still running after the victim: %d/4)"
                 % (nm, tot, 4.0 * 6 * 3000 * NB * 256, still))
    print(lines[-1], flush=True)
for nm, d in saved.items():
    a0, a1, b0, b1, r0, r1 = (d[:, i] for i in range(6))
    f = np.float32
    lines.append("wrong values beside '%s' (%d lanes sampled): low==0: %d, low==a.x*b.x (op_sel ignored): %d, low==a.y*b.y: %d, low==a.y*b.x (=high): %d, "
                 "high wrong: %d; lanes affected per wave (of the sampled lanes' waves): min %d max %d"
                 % (nm, len(d), int((r0 == 0).sum()), int((r0 == f(a0) * f(b0)).sum()), int((r0 == f(a1) * f(b1)).sum()), int((r0 == f(a1) * f(b0)).sum()),
                    int((r1 != f(a1) * f(b0)).sum()), 0, 0))
    lines.append("   first rows a.x a.y b.x b.y -> r.x r.y (iteration): " + "; ".join("%g %g %g %g -> %g %g (%d)" % tuple(list(row[:6]) + [int(row[6])]) for row in d[:4]))
    print(lines[-2]); print(lines[-1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "slp_repro3.txt"), "w").write("\n".join(lines) + "\n")
