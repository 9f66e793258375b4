#!/usr/bin/env python
"""Experiment 2 of the SLP / MFMA concurrency finding: which INSTRUCTION of the split-operand convolutions disturbs which FORM of a
packed-float32 instruction.  Noises: loops of one instruction each (v_cvt_pk_bf16_f32, v_cvt_pk_f16_f32, ds_read_b128,
v_cvt_f32_f16 sdwa, ds_bpermute_b32, v_mul_f32) and, for calibration, the library's bf16x3 convolution.  Victims: loops of one packed
instruction form each, every result compared in the same lane with the scalar instructions on the same registers; and the real
k_upsample2 built with SLP.  -> gpurun_out/slp_repro2.txt"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from v2e_amd.slomo import HipUNet
from v2e_amd.synth import portable_unet_state_dict
dev = torch.device("cuda")
rl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libslp_repro.so"))
sd_i = {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}
net = HipUNet(sd_i, 12, 5, dev, "bf16x3")
lib = net.lib
def P(t): return C.c_void_p(t.data_ptr())
def ST(): return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
d9 = net.descs[9]
xn = torch.rand((2, d9.cin, 16, 24), device=dev) - 0.4
yn = torch.empty((2, d9.cout, 16, 24), device=dev)
nout = torch.empty((1024 * 256,), device=dev)
NOISES = ["none", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "ds_read_b128", "v_cvt_f32_f16_sdwa", "ds_bpermute_b32", "v_mul_f32", "conv_bf16x3"]
ITERS = {"v_cvt_pk_bf16_f32": 1500000, "v_cvt_pk_f16_f32": 1500000, "ds_read_b128": 150000, "v_cvt_f32_f16_sdwa": 1500000,
         "ds_bpermute_b32": 150000, "v_mul_f32": 1500000}
def noise(kind):
    if kind == "none": return
    if kind == "conv_bf16x3":
        for _ in range(60):
            assert lib.v2e_conv2d_lrelu(P(xn), d9.cin, None, 0, 0, C.byref(d9), P(yn), 2, 16, 24, ST()) == 0
        return
    assert rl.slp_launch_noise_one(NOISES.index(kind) - 1, P(nout), 1024, ITERS[kind], ST()) == 0
FORMS = ["pk_mul", "pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_add", "pk_mul sgpr-pair", "pk_fma", "v_sub->hi half, swapped pk_mul"]
g = torch.Generator(device=dev); g.manual_seed(9)
xv = torch.rand((2048 * 256 * 4,), device=dev, generator=g) + 0.25
mism = torch.zeros((2,), dtype=torch.int32, device=dev)
NC, Hh, Ww = 256, 64, 96
xu = (torch.rand((NC, Hh // 2, Ww // 2), device=dev, generator=g) - 0.4) * 3.0
yu = torch.empty((NC, Hh, Ww), device=dev)
ref = torch.empty_like(yu)
assert rl.slp_launch_victim(0, P(xu), P(ref), NC, Hh, Ww, ST()) == 0
torch.cuda.synchronize()
side = torch.cuda.Stream(dev)
lines = []
for nk in NOISES:
    for form in range(len(FORMS) + 1):
        tot = [0, 0, 0]
        still = 0
        for rep in range(4):
            mism.zero_()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                noise(nk)
                ev = torch.cuda.Event(); ev.record()
            if form < len(FORMS):
                for _ in range(6):
                    assert rl.slp_launch_victim_one(form, P(xv), 2048, 3000, P(mism), ST()) == 0
                torch.cuda.current_stream().synchronize()
                still += int(not ev.query())
                torch.cuda.synchronize()
                m = mism.cpu().numpy()
                tot[0] += int(m[0]); tot[1] += int(m[1])
            else:
                bad = 0
                for _ in range(12):
                    assert rl.slp_launch_victim(0, P(xu), P(yu), NC, Hh, Ww, ST()) == 0
                    torch.cuda.current_stream().synchronize()
                    bad += int((yu.view(torch.int32) != ref.view(torch.int32)).sum())
                still += int(not ev.query())
                torch.cuda.synchronize()
                tot[2] += bad
        name = FORMS[form] if form < len(FORMS) else "k_upsample2 (SLP build)"
        lines.append("noise %-20s victim %-40s wrong low halves %9d, wrong high halves %9d, wrong outputs %7d  (noise still running after the victim: %d/4)"
                     % (nk, name, tot[0], tot[1], tot[2], still))
        print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "slp_repro2.txt"), "w").write("\n".join(lines) + "\n")
