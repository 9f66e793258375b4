#!/usr/bin/env python
"""dev tool / evidence: distance of the three convolution maths to the reference's float32 goldens (max |a - b| / max(1, |b|),
the north-star metric, bar 1e-5) and, on the stress fixtures, to the float64 result in units of the reference's own float32
distance (rms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from v2e_amd.slomo import SloMoEngine
from v2e_amd.synth import portable_unet_state_dict
from test_slomo_oracle_golden import bench_shape_inputs, _scaled_state_dicts, noise_ratio_rms, GOLDEN


def relerr(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))))


def pairs(z):
    fr = z["frames"]; n = len(fr) - 1
    return ((fr[:n].astype(np.float32) / np.float32(255.0))[:, None] - np.float32(0.428),
            (fr[1:n + 1].astype(np.float32) / np.float32(255.0))[:, None] - np.float32(0.428))


for math in ("bf16x3", "fp16x2", "f32"):
    out = []
    z = np.load(os.path.join(GOLDEN, "slomo_unet_64x96.npz")); I0, I1 = pairs(z); ts = list(z["ts"])
    eng = SloMoEngine({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, 101).items()},
                      {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}, "cuda", conv_math=math)
    Ft = eng.interpolate(torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda(), ts).cpu().numpy()
    out.append("64x96: flow %.2e Ft %.2e" % (relerr(eng.last["flow"].cpu().numpy(), z["flow"]), relerr(Ft, z["Ft"])))
    z = np.load(os.path.join(GOLDEN, "slomo_320x256.npz")); I0, I1 = bench_shape_inputs(z); ts = list(z["ts"]); sf, si = (int(v) for v in z["seeds"])
    eng = SloMoEngine({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, sf).items()},
                      {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, si).items()}, "cuda", conv_math=math)
    Ft = eng.interpolate(torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda(), ts).cpu().numpy()
    out.append("320x256: flow %.2e Ft %.2e" % (relerr(eng.last["flow"].cpu().numpy(), z["flow"]), relerr(Ft, z["Ft"])))
    for fx in ("slomo_trained_scale_64x96", "slomo_allscale_64x96"):
        z = np.load(os.path.join(GOLDEN, fx + ".npz")); I0, I1 = pairs(z); ts = list(z["ts"]); sd_f, sd_i = _scaled_state_dicts(z)
        eng = SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()}, {k: torch.from_numpy(v) for k, v in sd_i.items()}, "cuda", conv_math=math)
        Ft = eng.interpolate(torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda(), ts).cpu().numpy()
        got = {"flow": eng.last["flow"].cpu().numpy(), "intrp": eng.last["intrp"].cpu().numpy().reshape(len(ts), I0.shape[0], 5, 64, 96), "Ft": Ft}
        out.append("%s: |x - f64| / |ref_f32 - f64| rms flow %.2f intrp %.2f Ft %.2f" % (fx.replace("slomo_", "").replace("_64x96", ""),
                   *[noise_ratio_rms(got[k], z, k) for k in ("flow", "intrp", "Ft")]))
    print("%-7s %s" % (math, " | ".join(out)))
