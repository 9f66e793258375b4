#!/usr/bin/env python
"""Golden vectors for the SuperSloMo path AT THE BENCHMARKED SHAPE, from the reference's own
modules (v2ecore.model.UNet / backWarp imported from /root/reference; slomo.py:343-345, 404-433
driven line by line by make_golden_slomo.ref_interp).

  slomo_320x256.npz            346x260 source frames -> PIL LANCZOS 320x256 (dataloader.py:122-147),
                               B=2 pairs, U=10 time points (the v2e CLI's 10x slowdown): `flow`, `Ft`
                               in full, `intrp` on a stride-8 lattice plus SHA-256 of the full tensor.
                               The GPU test tiles the two pairs to B=8 so the interpolation UNet sees the
                               80 samples of the bench line (wide 64-channel k_conv tiles are dispatched).
  slomo_trained_scale_64x96.npz  same weights with conv3 of both nets scaled so that |flow| reaches ~30 px
                               (the magnitude a trained checkpoint produces): warps far outside
                               the image, visibility logits saturating the sigmoid.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402
from make_golden_slomo import ref_interp  # noqa: E402
from v2e_amd.synth import int_gradient_frames, portable_unet_state_dict  # noqa: E402

SRC_H, SRC_W, H, W, B, U = 260, 346, 256, 320, 2, 10
SEED_FRAMES, SEED_F, SEED_I = 21, 501, 502


def network_inputs(frames_u8, dim):
    """dataloader.py:136-147 + slomo.py:148-151: PIL LANCZOS resize, ToTensor, Normalize(0.428, 1)."""
    from PIL import Image
    rs = np.stack([np.asarray(Image.fromarray(f).resize(dim, Image.LANCZOS)) for f in frames_u8])
    t = torch.from_numpy(rs.astype(np.float32) / 255.0).unsqueeze(1) - 0.428
    return t[:-1].contiguous(), t[1:].contiguous()


def nets(model, seed_f, seed_i, scale_f=1.0, scale_i=1.0):
    sd_f, sd_i = portable_unet_state_dict(2, 4, seed_f), portable_unet_state_dict(12, 5, seed_i)
    for sd, s in ((sd_f, scale_f), (sd_i, scale_i)):
        if s != 1.0:
            sd["conv3.weight"] = (sd["conv3.weight"] * np.float32(s)).astype(np.float32)
            sd["conv3.bias"] = (sd["conv3.bias"] * np.float32(s)).astype(np.float32)
    flow_net, interp_net = model.UNet(2, 4), model.UNet(12, 5)
    flow_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    interp_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_i.items()})
    return flow_net, interp_net


def main():
    model = rh.ref_model()
    torch.set_num_threads(8)
    fr = int_gradient_frames(B + 1, SRC_H, SRC_W, seed=SEED_FRAMES, noise=10, as_array=True)
    I0, I1 = network_inputs(fr, (W, H))
    flow_net, interp_net = nets(model, SEED_F, SEED_I)
    warper = model.backWarp(W, H, "cpu")
    ts = [(k + 0.5) / U for k in range(U)]
    with torch.no_grad():
        o = ref_interp(model, flow_net, interp_net, warper, I0, I1, ts)
    intrp = o["intrp"].numpy()
    np.savez_compressed(os.path.join(HERE, "slomo_320x256.npz"),
                        frame_args=np.asarray([B + 1, SRC_H, SRC_W, SEED_FRAMES, 10]), seeds=np.asarray([SEED_F, SEED_I]),
                        ts=np.asarray(ts), flow=o["flow"].numpy(), Ft=o["Ft"].numpy(),
                        intrp_lattice=np.ascontiguousarray(intrp[:, :, :, ::8, ::8]),
                        intrp_sha256=hashlib.sha256(np.ascontiguousarray(intrp).tobytes()).hexdigest(),
                        torch_version=torch.__version__)
    print("slomo_320x256: |flow|max %.4f |intrp|max %.4f Ft range [%.3f, %.3f]" % (
        o["flow"].abs().max(), o["intrp"].abs().max(), o["Ft"].min(), o["Ft"].max()))

    # trained-scale outputs at a small shape
    h, w, b = 64, 96, 2
    fr2 = int_gradient_frames(b + 1, h, w, seed=11, noise=10, as_array=True)
    J0 = torch.from_numpy(fr2[:b].astype(np.float32) / 255.0).unsqueeze(1) - 0.428
    J1 = torch.from_numpy(fr2[1:b + 1].astype(np.float32) / 255.0).unsqueeze(1) - 0.428
    sf, si = 200.0, 40.0
    flow_net, interp_net = nets(model, 101, 102, sf, si)
    warper = model.backWarp(w, h, "cpu")
    ts2 = [(k + 0.5) / 3 for k in range(3)]
    with torch.no_grad():
        o2 = ref_interp(model, flow_net, interp_net, warper, J0, J1, ts2)
    np.savez_compressed(os.path.join(HERE, "slomo_trained_scale_64x96.npz"), frames=fr2, ts=np.asarray(ts2),
                        conv3_scale=np.asarray([sf, si]), flow=o2["flow"].numpy(), intrp=o2["intrp"].numpy(),
                        Ft=o2["Ft"].numpy(), torch_version=torch.__version__)
    print("slomo_trained_scale_64x96: |flow|max %.3f |intrp|max %.3f Ft range [%.3f, %.3f] outside[0,1]-0.428: %.3f" % (
        o2["flow"].abs().max(), o2["intrp"].abs().max(), o2["Ft"].min(), o2["Ft"].max(),
        float(((o2["Ft"] + 0.428 < 0) | (o2["Ft"] + 0.428 > 1)).float().mean())))
    for f in ("slomo_320x256.npz", "slomo_trained_scale_64x96.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
