"""BASELINE configs[2] end to end AGAINST THE REFERENCE (round-3 review, "What's weak" 2): tests/golden/make_golden_config3.py ran
the reference `SuperSloMo` class over a 31-frame 346x260 clip (U = 10 -> 300 PNG frames) and the reference `EventEmulator`
(Philox source swapped in) over those frames as v2e.py feeds them.  Both product paths are run over the same clip here -- the
drop-in classes through PNG files, and VideoToEvents entirely in HBM -- and compared frame by frame.

What can be expected: the SloMo floats agree with the reference to ~1e-6 (bar 1e-5), then `(x + mean) * 255` is TRUNCATED to
8 bits, so a value within 1e-6 of an integer can land on the other side: the frames agree to <= 1 grey level on a small
fraction of the pixels, and a frame with such a pixel can move an event.  The test prints what it measures and asserts the
bounds measured on MI355X (with margin): it pins how far from "event for event" the end-to-end run is."""
import os

import numpy as np
import pytest

from fixtures import GOLDEN, sha

pytestmark = pytest.mark.gpu


def _load():
    from v2e_amd.synth import int_gradient_frames
    z = np.load(os.path.join(GOLDEN, "config3_ref_346x260.npz"))
    n, h, w, seed, noise = (int(v) for v in z["frame_args"])
    return z, int_gradient_frames(n, h, w, seed=seed, noise=noise, as_array=True)


def _maps(evs, h, w):
    on, off = np.zeros((h, w), np.int64), np.zeros((h, w), np.int64)
    for e in evs:
        if e is not None and len(e):
            x, y = e[:, 1].astype(np.int64), e[:, 2].astype(np.int64)
            np.add.at(on, (y[e[:, 3] > 0], x[e[:, 3] > 0]), 1)
            np.add.at(off, (y[e[:, 3] < 0], x[e[:, 3] < 0]), 1)
    return on, off


def _report(tag, z, pngs, evs):
    h, w = pngs.shape[1:]
    png_exact = np.array([sha(p) == str(s) for p, s in zip(pngs, z["png_sha"])])
    ref10 = z["pngs_every10"]
    d = np.abs(pngs[::10].astype(int) - ref10.astype(int))
    n_ev = np.array([0 if e is None else len(e) for e in evs], np.int64)
    ev_exact = np.array([(n == 0 and not str(s)) or (n > 0 and sha(e) == str(s)) for e, n, s in zip(evs, n_ev, z["ev_sha"])])
    ref_n = z["n_events"]
    on, off = _maps(evs, h, w)
    px = ((on != z["on_map"]) | (off != z["off_map"])).mean()
    tot, ref_tot = int(n_ev.sum()), int(ref_n.sum())
    m = dict(png_frames_exact=float(png_exact.mean()), png_max_lsb=int(d.max()), png_pixels_off=float((d > 0).mean()),
             event_frames_exact=float(ev_exact.mean()), first_event_frame_off=int(np.argmin(ev_exact)) if not ev_exact.all() else -1,
             events=tot, events_ref=ref_tot, total_rel=abs(tot - ref_tot) / ref_tot,
             per_frame_count_max_rel=float(np.max(np.abs(n_ev - ref_n) / np.maximum(ref_n, 1))), pixels_with_other_totals=float(px))
    print("config 3 vs reference [%s]: %s" % (tag, m))
    return m


def _check(m):
    assert m["png_max_lsb"] <= 1 and m["png_pixels_off"] < 1e-3, m       # frames: one grey level on < 0.1 % of the pixels
    assert m["total_rel"] < 2e-3, m                                        # total events within 0.2 %
    assert m["per_frame_count_max_rel"] < 0.02, m                          # every frame's count within 2 %
    assert m["pixels_with_other_totals"] < 0.01, m                         # < 1 % of the pixels end with other ON / OFF totals


def test_drop_in_classes_against_the_reference_run(tmp_path):
    """v2e_amd.SuperSloMo.interpolate (PNG files) + v2e_amd.EventEmulator.generate_events per frame, as v2e.py drives them."""
    import torch
    from PIL import Image
    from v2e_amd import EventEmulator, SuperSloMo
    from v2e_amd.synth import portable_unet_state_dict
    import bench as B
    z, fr = _load()
    U, batch = int(z["U"]), int(z["batch"])
    sf, si, se = (int(v) for v in z["seeds"])
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    for i, f in enumerate(fr):
        np.save(str(src / ("%08d.npy" % i)), f)
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"state_dictFC": {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, sf).items()},
                "state_dictAT": {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, si).items()}}, str(ckpt))
    sm = SuperSloMo(model=str(ckpt), auto_upsample=False, upsampling_factor=U, batch_size=batch)
    interp_times, avg = sm.interpolate(str(src), str(dst), (fr.shape[2], fr.shape[1]))
    assert np.array_equal(interp_times, z["interp_times"]) and avg == float(z["avg"])
    n = (len(fr) - 1) * U
    pngs = np.stack([np.asarray(Image.open(str(dst / ("%d.png" % i)))) for i in range(n)])
    duration = (len(fr) - 1) / float(z["src_fps"])
    times = (duration / (np.max(interp_times) - np.min(interp_times))) * np.asarray(interp_times)  # v2e.py:786-797
    assert np.array_equal(times, z["times"])
    emu = EventEmulator(device="cuda", seed=se, rng_mode="philox", **B.DEFAULT_KW)
    evs = [emu.generate_events(p.astype(np.float32), float(t)) for p, t in zip(pngs, times)]  # v2e.py:826-834
    _check(_report("drop-in classes, PNG files", z, pngs, evs))
    # fed the REFERENCE's frames, the emulator gives the reference's events bit for bit (the divergence above is the SloMo
    # stage's 1-LSB pixels, nothing else): checked on the stored every-10th frames' neighbourhood via a short clip
    emu2 = EventEmulator(device="cuda", seed=se, rng_mode="philox", **B.DEFAULT_KW)
    if bool(np.array([sha(p) == str(s) for p, s in zip(pngs[:20], z["png_sha"][:20])]).all()):
        for k in range(20):
            e = emu2.generate_events(pngs[k].astype(np.float32), float(times[k]))
            assert (0 if e is None else len(e)) == z["n_events"][k] and (e is None or sha(e) == str(z["ev_sha"][k])), k


def test_device_pipeline_against_the_reference_run():
    """VideoToEvents: the same clip without leaving HBM (what bench.py's end_to_end object times)."""
    import torch
    from v2e_amd import EventEmulator
    from v2e_amd.pipeline import VideoToEvents, interp_frame_times
    from v2e_amd.slomo import SloMoEngine
    from v2e_amd.synth import portable_unet_state_dict
    import bench as B
    z, fr = _load()
    U, batch = int(z["U"]), int(z["batch"])
    sf, si, se = (int(v) for v in z["seeds"])
    eng = SloMoEngine({k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, sf).items()},
                      {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, si).items()}, "cuda")
    assert np.array_equal(interp_frame_times(len(fr), U, 1 / float(z["src_fps"]), batch), z["times"])
    pipe = VideoToEvents(eng, EventEmulator(device="cuda", seed=se, rng_mode="philox", **B.DEFAULT_KW), U, batch_size=batch)
    src = torch.from_numpy(fr).cuda()
    up = pipe.upsample(src).cpu().numpy()
    ev, counts, nfr = pipe.run(src, 1 / float(z["src_fps"]))
    assert nfr == len(z["n_events"])
    rows = np.cumsum(np.concatenate(([0], counts)))
    evs = [ev[rows[k]:rows[k + 1]] if counts[k] else None for k in range(nfr)]
    _check(_report("VideoToEvents, in HBM", z, up, evs))
