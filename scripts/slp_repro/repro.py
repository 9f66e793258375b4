#!/usr/bin/env python
"""Root-cause experiment for the round-4 concurrency finding (profiles/r04_concurrency_finding.txt).

victims: v2e_amd/csrc/slomo.hip's k_upsample2 built WITH the SLP vectoriser (24 v_pk_mul_f32 / v_pk_add_f32), WITHOUT it, and
         WITH it plus -mllvm -amdgpu-waitcnt-forcezero (an s_waitcnt 0 behind every instruction), scripts/slp_repro/build.sh
noises (second stream): one convolution of the library per conv math (bf16x3 / fp16x2 / f32), launched 40 times, and a bare
         v_mfma_f32_32x32x16_bf16 loop on toggling operands
A victim launch is 'wrong' if any output differs bitwise from the same kernel's stand-alone output on the same input.
Output: one line per (victim, noise, fresh-input) and, for the first wrong launch of a pair, the mismatching values with what
the scalar build gives -- gpurun_out/slp_repro.txt / .npz."""
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
NOISE_LAYER = 9  # down4.conv2 (512 -> 512, k3) @16x24: the noise of table 3
xn = torch.rand((2, nets["f32"].descs[NOISE_LAYER].cin, 16, 24), device=dev) - 0.4
yn = torch.empty((2, nets["f32"].descs[NOISE_LAYER].cout, 16, 24), device=dev)
bare_out = torch.empty((1024 * 256,), device=dev)
def noise(kind):
    if kind == "none": return
    if kind == "bare_bf16_mfma":
        assert rl.slp_launch_noise(P(bare_out), 1024, 60000, ST()) == 0
        return
    d = nets[kind].descs[NOISE_LAYER]
    for _ in range(40):
        assert lib.v2e_conv2d_lrelu(P(xn), d.cin, None, 0, 0, C.byref(d), P(yn), 2, 16, 24, ST()) == 0
NC, Hh, Ww = 256, 64, 96  # up5's x2 in the 4 x 12 x 64 x 96 victim net: 1.57 M outputs
g = torch.Generator(device=dev); g.manual_seed(5)
x_src = (torch.rand((NC, Hh // 2, Ww // 2), device=dev, generator=g) - 0.4) * 3.0
x = x_src.clone()
NV = 8
ys = [torch.empty((NC, Hh, Ww), device=dev) for _ in range(NV)]
def victim(which, y, fresh):
    if fresh: x.copy_(x_src)  # the input was written by the kernel right before, as in the net
    assert rl.slp_launch_victim(which, P(x), P(y), NC, Hh, Ww, ST()) == 0
side = torch.cuda.Stream(dev)
names = ["slp", "noslp", "slp_waitcnt0"]
refs = []
for w in range(3):
    r = torch.empty((NC, Hh, Ww), device=dev); victim(w, r, False); torch.cuda.synchronize(); refs.append(r.clone())
lines = ["stand-alone: slp == noslp bitwise: %s, slp_waitcnt0 == noslp: %s" % (torch.equal(refs[0], refs[1]), torch.equal(refs[2], refs[1]))]
dump = {}
for nk in ("none", "f32", "bf16x3", "fp16x2", "bare_bf16_mfma"):
    for w in range(3):
        for fresh in (False, True):
            bad_launches, bad_vals, worst = 0, 0, 0.0
            for rep in range(10):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    noise(nk)
                for y in ys: victim(w, y, fresh)
                torch.cuda.synchronize()
                for y in ys:
                    ne = (y.view(torch.int32) != refs[w].view(torch.int32))
                    n = int(ne.sum())
                    if n:
                        bad_launches += 1; bad_vals += n
                        worst = max(worst, float(((y - refs[w]).abs() / refs[w].abs().clamp_min(1e-6))[ne].max()))
                        key = "%s|%s" % (names[w], nk)
                        if key not in dump:
                            idx = ne.flatten().nonzero().flatten()[:4096]
                            dump[key] = (idx.cpu().numpy(), y.flatten()[idx].cpu().numpy(), refs[w].flatten()[idx].cpu().numpy())
            lines.append("victim %-13s noise %-15s fresh_input %d: wrong launches %2d/%d, wrong values %d, worst rel %.3g"
                         % (names[w], nk, int(fresh), bad_launches, 10 * NV, bad_vals, worst))
            print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "slp_repro.txt"), "w").write("\n".join(lines) + "\n")
np.savez(os.path.join(ROOT, "gpurun_out", "slp_repro.npz"), x=x_src.cpu().numpy(),
         **{k.replace("|", "__") + "__" + f: v[i] for k, v in dump.items() for i, f in enumerate(("idx", "got", "ref"))})
