#!/usr/bin/env python
"""BASELINE configs[2] END TO END from the reference: the reference `SuperSloMo` CLASS (v2ecore/slomo.py, GPU branch of its
transforms as in make_golden_slomo_class.py) over a 31-frame 346x260 clip, U = 10, batch_size 8 -> 300 PNG frames, then the
reference `EventEmulator` over those frames exactly as v2e.py:798-834 feeds them (read_image: 8-bit grey -> float32; frame
times interpTimes * duration / (max - min)), with the reference's random SOURCE swapped for the portable Philox streams as
make_golden.py does (run_reference_philox).  Seeded random-init checkpoint (the trained one is not obtainable offline).

  config3_ref_346x260.npz   per interpolated frame: SHA-256 of the PNG pixels, event count, SHA-256 of the event rows;
                            every 10th PNG frame in full; per-pixel ON / OFF event totals of the whole clip; the times.

The HIP path meets the SloMo tolerance (1e-5 on floats), after which the frames are TRUNCATED to 8 bits: a float within 1e-5
of a quantisation boundary can land on the other side, so PNG frames agree to <= 1 LSB on a small fraction of the pixels and
the event streams downstream cannot be bit-equal everywhere.  tests/test_config3_vs_reference.py measures exactly that
(fraction of bit-identical PNG frames / event frames, total and per-pixel count divergence) and asserts the measured bounds.
"""
import hashlib
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
from make_golden import DEFAULTS, run_reference_philox, sha  # noqa: E402
from v2e_amd.synth import int_gradient_frames, portable_unet_state_dict  # noqa: E402

HS, WS, N_SRC, U, BATCH = 260, 346, 31, 10, 8
SEED_FRAMES, SEED_F, SEED_I, SEED_EMU = 31, 101, 102, 5
SRC_FPS = 30.0


def main():
    import logging
    logging.disable(logging.CRITICAL)
    SuperSloMo = rh.ref_slomo_cls()
    torch.set_num_threads(8)
    fr = int_gradient_frames(N_SRC, HS, WS, seed=SEED_FRAMES, noise=10, as_array=True)
    with tempfile.TemporaryDirectory() as td:
        src, dst = os.path.join(td, "src"), os.path.join(td, "dst")
        os.mkdir(src); os.mkdir(dst)
        for i, f in enumerate(fr):
            np.save(os.path.join(src, "%08d.npy" % i), f)
        sd_f, sd_i = portable_unet_state_dict(2, 4, SEED_F), portable_unet_state_dict(12, 5, SEED_I)
        ckpt = os.path.join(td, "ckpt.pt")
        torch.save({"state_dictFC": {k: torch.from_numpy(v) for k, v in sd_f.items()},
                    "state_dictAT": {k: torch.from_numpy(v) for k, v in sd_i.items()}}, ckpt)
        sm = SuperSloMo(model=ckpt, auto_upsample=False, upsampling_factor=U, batch_size=BATCH)
        sm.device = torch.device("cpu")  # != "cpu": the GPU branch of slomo.py:154-161, see make_golden_slomo_class.py
        sm.to_tensor, sm.to_image = sm._SuperSloMo__transform()
        with torch.no_grad():
            interp_times, avg = sm.interpolate(src, dst, (WS, HS))
        n = (N_SRC - 1) * U
        pngs = np.stack([np.asarray(Image.open(os.path.join(dst, "%d.png" % i))) for i in range(n)])
    assert pngs.shape == (n, HS, WS) and pngs.dtype == np.uint8 and len(interp_times) == n
    # v2e.py:786-797: normalise the interpolated frame times to the processed duration of the source video
    duration = (N_SRC - 1) / SRC_FPS
    times = (duration / (np.max(interp_times) - np.min(interp_times))) * np.asarray(interp_times)
    frames = [p.astype(np.float32) for p in pngs]  # v2e_utils.read_image: cv2.IMREAD_GRAYSCALE -> float32
    evs, ref, nlin = run_reference_philox(frames, [float(t) for t in times], dict(DEFAULTS), None, SEED_EMU)
    on = np.zeros((HS, WS), np.int32)
    off = np.zeros((HS, WS), np.int32)
    for e in evs:
        if e is not None and len(e):
            x, y = e[:, 1].astype(np.int64), e[:, 2].astype(np.int64)
            np.add.at(on, (y[e[:, 3] > 0], x[e[:, 3] > 0]), 1)
            np.add.at(off, (y[e[:, 3] < 0], x[e[:, 3] < 0]), 1)
    out = os.path.join(HERE, "config3_ref_346x260.npz")
    np.savez_compressed(out, frame_args=np.asarray([N_SRC, HS, WS, SEED_FRAMES, 10]), seeds=np.asarray([SEED_F, SEED_I, SEED_EMU]),
                        U=U, batch=BATCH, src_fps=SRC_FPS, interp_times=np.asarray(interp_times), times=times, avg=np.float64(avg),
                        png_sha=np.array([sha(p) for p in pngs]), pngs_every10=pngs[::10].copy(),
                        n_events=np.array([0 if e is None else len(e) for e in evs], np.int64),
                        ev_sha=np.array([sha(e) if e is not None else "" for e in evs]),
                        on_map=on.astype(np.int16), off_map=off.astype(np.int16),
                        counters=np.array([ref.num_events_total, ref.num_events_on, ref.num_events_off], np.int64),
                        torch_version=torch.__version__)
    print("config3_ref_346x260: %d PNG frames, %d events (%d on / %d off), linspace-checked %d, %d KB" % (
        n, ref.num_events_total, ref.num_events_on, ref.num_events_off, nlin, os.path.getsize(out) // 1024))


if __name__ == "__main__":
    main()
