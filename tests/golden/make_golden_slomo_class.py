#!/usr/bin/env python
"""PNG-level golden of the SloMo stage from the reference CLASS itself: v2ecore.slomo.SuperSloMo.interpolate run on
a 40x70 clip (dataloader.FramesDirectory: np.load -> PIL LANCZOS 64x32 -> ToTensor -> Normalize; flow + interpolation
UNets; revNormalize -> ToPILImage -> PIL BILINEAR 70x40 -> <idx>.png), seeded random-init checkpoint written to a file.
The reference picks its transforms by device (slomo.py:154-161: the CPU branch skips the normalisation); the drop-in
mirrors the GPU branch, so the class runs here with `device = torch.device("cpu")`, which is not == "cpu" and selects that branch
(`pngs`); the CPU branch's output is stored too (`pngs_cpu_branch`).
torchvision is not in this image: ref_harness.install_torchvision_stub restates the three transforms.

  slomo_class_40x70.npz  source frames, the 15 PNG frames as written, interpTimes, avgUpsampling
"""
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402
from v2e_amd.synth import int_gradient_frames, portable_unet_state_dict  # noqa: E402


def main():
    SuperSloMo = rh.ref_slomo_cls()
    torch.set_num_threads(8)
    Hs, Ws, n, U = 40, 70, 6, 3
    fr = int_gradient_frames(n, Hs, Ws, seed=9, noise=6, as_array=True)
    with tempfile.TemporaryDirectory() as td:
        src, dst = os.path.join(td, "src"), os.path.join(td, "dst")
        os.mkdir(src); os.mkdir(dst)
        for i, f in enumerate(fr):
            np.save(os.path.join(src, "%08d.npy" % i), f)
        sd_f, sd_i = portable_unet_state_dict(2, 4, 401), portable_unet_state_dict(12, 5, 402)
        ckpt = os.path.join(td, "ckpt.pt")
        torch.save({"state_dictFC": {k: torch.from_numpy(v) for k, v in sd_f.items()},
                    "state_dictAT": {k: torch.from_numpy(v) for k, v in sd_i.items()}}, ckpt)
        out = {}
        for branch in ("gpu", "cpu"):
            dst_b = os.path.join(td, "dst_" + branch)
            os.mkdir(dst_b)
            sm = SuperSloMo(model=ckpt, auto_upsample=False, upsampling_factor=U, batch_size=2)
            if branch == "gpu":
                # != "cpu" (a str): the GPU branch of slomo.py:154-161 (Normalize / revNormalize); the transforms are
                # built in the constructor (slomo.py:118), so build them again
                sm.device = torch.device("cpu")
                sm.to_tensor, sm.to_image = sm._SuperSloMo__transform()
            times, avg = sm.interpolate(src, dst_b, (Ws, Hs))
            out[branch] = np.stack([np.asarray(Image.open(os.path.join(dst_b, "%d.png" % i))) for i in range((n - 1) * U)])
        pngs = out["gpu"]
    np.savez_compressed(os.path.join(HERE, "slomo_class_40x70.npz"), frames=fr, pngs=pngs, pngs_cpu_branch=out["cpu"], times=np.asarray(times),
                        avg=np.float64(avg), U=U, seeds=np.asarray([401, 402]), torch_version=torch.__version__)
    print("slomo_class_40x70: %d PNG frames %s, times %s.., avg %.1f" % (len(pngs), pngs.shape[1:], times[:4], avg))


if __name__ == "__main__":
    main()
