#!/usr/bin/env python
"""Golden vectors for the SuperSloMo path from the reference's own modules
(v2ecore.model.UNet / backWarp, imported from /root/reference) and the inner math of
v2ecore/slomo.py:338-345, 404-433 (driven line by line: the SuperSloMo class itself needs a
checkpoint file and cv2/torchvision).  Weights are the portable seeded random init of
v2e_amd.synth.portable_unet_state_dict (the trained checkpoint is not available offline).

  slomo_unet_64x96.npz   flow UNet and interpolation UNet outputs for B=2 pairs, 3 time points
  slomo_warp_64x96.npz   backWarp / blend / fusion with large random flows (hits the zero-padding edges)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402
from v2e_amd.synth import int_gradient_frames, portable_unet_state_dict  # noqa: E402


def ref_interp(model, flow_net, interp_net, warper, I0, I1, ts, flowOut=None, intrp_override=None):
    """slomo.py:343-345, 404-433 verbatim on given tensors; returns dict of intermediates."""
    out = {}
    if flowOut is None:
        flowOut = flow_net(torch.cat((I0, I1), dim=1))
    F_0_1 = flowOut[:, :2, :, :]
    F_1_0 = flowOut[:, 2:, :, :]
    out["flow"] = flowOut
    Fts, x12s, intrps = [], [], []
    for ti, t in enumerate(ts):
        temp = -t * (1 - t)
        fCoeff = [temp, t * t, (1 - t) * (1 - t), temp]
        F_t_0 = fCoeff[0] * F_0_1 + fCoeff[1] * F_1_0
        F_t_1 = fCoeff[2] * F_0_1 + fCoeff[3] * F_1_0
        g_I0_F_t_0 = warper(I0, F_t_0)
        g_I1_F_t_1 = warper(I1, F_t_1)
        x12 = torch.cat((I0, I1, F_0_1, F_1_0, F_t_1, F_t_0, g_I1_F_t_1, g_I0_F_t_0), dim=1)
        intrpOut = interp_net(x12) if intrp_override is None else intrp_override[ti]
        F_t_0_f = intrpOut[:, :2, :, :] + F_t_0
        F_t_1_f = intrpOut[:, 2:4, :, :] + F_t_1
        V_t_0 = torch.sigmoid(intrpOut[:, 4:5, :, :])
        V_t_1 = 1 - V_t_0
        g_I0_F_t_0_f = warper(I0, F_t_0_f)
        g_I1_F_t_1_f = warper(I1, F_t_1_f)
        wCoeff = [1 - t, t]
        Ft_p = (wCoeff[0] * V_t_0 * g_I0_F_t_0_f + wCoeff[1] * V_t_1 * g_I1_F_t_1_f) / \
               (wCoeff[0] * V_t_0 + wCoeff[1] * V_t_1)
        Fts.append(Ft_p)
        x12s.append(x12)
        intrps.append(intrpOut)
    out["Ft"] = torch.stack(Fts)          # [n_t, B, 1, H, W]
    out["x12"] = torch.stack(x12s)        # [n_t, B, 12, H, W]
    out["intrp"] = torch.stack(intrps)    # [n_t, B, 5, H, W]
    return out


def main():
    model = rh.ref_model()
    torch.set_num_threads(8)
    H, W, B = 64, 96, 2
    fr = int_gradient_frames(B + 1, H, W, seed=11, noise=10, as_array=True)
    I0 = torch.from_numpy(fr[:B].astype(np.float32) / 255.0).unsqueeze(1) - 0.428
    I1 = torch.from_numpy(fr[1:B + 1].astype(np.float32) / 255.0).unsqueeze(1) - 0.428
    flow_net = model.UNet(2, 4)
    interp_net = model.UNet(12, 5)
    flow_net.load_state_dict({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, 101).items()})
    interp_net.load_state_dict({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()})
    warper = model.backWarp(W, H, "cpu")
    ts = [(k + 0.5) / 3 for k in range(3)]
    with torch.no_grad():
        o = ref_interp(model, flow_net, interp_net, warper, I0, I1, ts)
    np.savez_compressed(os.path.join(HERE, "slomo_unet_64x96.npz"), frames=fr, ts=np.asarray(ts),
                        flow=o["flow"].numpy(), intrp=o["intrp"].numpy(), Ft=o["Ft"].numpy(),
                        torch_version=torch.__version__)
    print("slomo_unet_64x96: |flow|max %.4f |intrp|max %.4f Ft range [%.3f, %.3f]" % (
        o["flow"].abs().max(), o["intrp"].abs().max(), o["Ft"].min(), o["Ft"].max()))

    # warps with large flows and synthetic interpolation-net outputs
    rng = np.random.Generator(np.random.PCG64(5))
    def ri(shape, scale):
        return torch.from_numpy(((rng.integers(0, 1 << 16, size=shape).astype(np.float32) / 32768.0) - 1.0) * np.float32(scale))
    flow = ri((1, 4, H, W), 9.0)
    intrp = [ri((1, 5, H, W), 3.0) for _ in ts]
    with torch.no_grad():
        o = ref_interp(model, None, None, warper, I0[:1], I1[:1], ts, flowOut=flow, intrp_override=intrp)
    np.savez_compressed(os.path.join(HERE, "slomo_warp_64x96.npz"), frames=fr[:2], ts=np.asarray(ts), flow=flow.numpy(),
                        intrp=torch.stack(intrp).numpy(), x12_tail=o["x12"].numpy()[:, :, 6:12], Ft=o["Ft"].numpy(),
                        torch_version=torch.__version__)
    print("slomo_warp_64x96: Ft range [%.3f, %.3f]" % (o["Ft"].min(), o["Ft"].max()))
    for f in ("slomo_unet_64x96.npz", "slomo_warp_64x96.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
