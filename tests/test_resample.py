"""The Pillow-exact resampling tables (v2e_amd/resample.py) against PIL itself (CPU), and the
HIP resample / quantise kernels against PIL (GPU)."""
import numpy as np
import pytest
from PIL import Image

CASES = [(260, 346, 256, 320, "lanczos"), (256, 320, 260, 346, "bilinear"), (40, 70, 32, 64, "lanczos"),
         (32, 64, 40, 70, "bilinear"), (37, 64, 32, 64, "lanczos"), (64, 96, 64, 100, "bilinear")]
PILF = {"lanczos": Image.LANCZOS, "bilinear": Image.BILINEAR}


@pytest.mark.parametrize("case", CASES)
def test_tables_reproduce_pillow(case):
    from v2e_amd.resample import resample_u8_numpy
    ih, iw, oh, ow, f = case
    rng = np.random.Generator(np.random.PCG64(ih * 7 + ow))
    img = rng.integers(0, 256, size=(ih, iw), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), PILF[f]))
    assert np.array_equal(resample_u8_numpy(img, (oh, ow), f), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [(720, 1280, 704, 1280, "lanczos")])
def test_hip_resample_equals_pillow(case):
    import torch
    from v2e_amd.pipeline import DeviceResampler
    ih, iw, oh, ow, f = case
    rng = np.random.Generator(np.random.PCG64(ih + ow))
    imgs = rng.integers(0, 256, size=(3, ih, iw), dtype=np.uint8)
    rs = DeviceResampler((ih, iw), (oh, ow), f, "cuda")
    got = rs(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(3):
        ref = np.asarray(Image.fromarray(imgs[i]).resize((ow, oh), PILF[f]))
        assert np.array_equal(got[i], ref)


@pytest.mark.gpu
def test_device_pipeline_equals_png_pipeline(tmp_path):
    """VideoToEvents (everything in HBM) == SuperSloMo.interpolate (PNG files) + per-frame emulator,
    frame for frame (uint8) and event for event."""
    import os
    import torch
    from v2e_amd import EventEmulator, SuperSloMo
    from v2e_amd.pipeline import VideoToEvents
    from v2e_amd.synth import int_gradient_frames, portable_unet_state_dict
    Hs, Ws, n, U = 40, 70, 6, 3
    fr = int_gradient_frames(n, Hs, Ws, seed=9, noise=6, as_array=True)
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    for i, f in enumerate(fr):
        np.save(str(src / ("%08d.npy" % i)), f)
    sd_f, sd_i = portable_unet_state_dict(2, 4, 401), portable_unet_state_dict(12, 5, 402)
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"state_dictFC": {k: torch.from_numpy(v) for k, v in sd_f.items()},
                "state_dictAT": {k: torch.from_numpy(v) for k, v in sd_i.items()}}, str(ckpt))
    sm = SuperSloMo(model=str(ckpt), auto_upsample=False, upsampling_factor=U, batch_size=2)
    times, _ = sm.interpolate(str(src), str(dst), (Ws, Hs))
    pngs = [np.asarray(Image.open(str(dst / ("%d.png" % i)))) for i in range((n - 1) * U)]
    kw = dict(pos_thres=0.15, neg_thres=0.15, sigma_thres=0.03, cutoff_hz=200, leak_rate_hz=0.1, shot_noise_rate_hz=2.0,
              refractory_period_s=0.001)
    dt_src = 1 / 30.0
    ref_emu = EventEmulator(device="cuda", seed=5, rng_mode="philox", **kw)
    duration = (n - 1) * dt_src  # v2e.py:786-797: the interpTimes are stretched to span the processed duration
    vt = (duration / (np.max(times) - np.min(times))) * times
    ref_ev = [ref_emu.generate_events(p.astype(np.float32), float(t)) for p, t in zip(pngs, vt)]  # v2e.py:832-834
    ref_ev = np.concatenate([e for e in ref_ev if e is not None])
    pipe = VideoToEvents(sm.engine, EventEmulator(device="cuda", seed=5, rng_mode="philox", **kw), U, batch_size=2)
    up = pipe.upsample(torch.from_numpy(fr).cuda())
    assert np.array_equal(up.cpu().numpy(), np.stack(pngs)), "device frames differ from the PNG frames"
    ev, counts, nfr = pipe.run(torch.from_numpy(fr).cuda(), dt_src)
    assert nfr == (n - 1) * U and int(counts.sum()) == len(ref_ev)
    assert np.array_equal(ev, ref_ev)


@pytest.mark.gpu
def test_quantisation_equals_torch_cpu_byte_conversion_out_of_range():
    """revNormalize + ToPILImage (slomo.py:153-161, 437): (x + 0.428).mul(255).byte() as torch computes it on the CPU,
    including values below 0 and above 1 (a float -> uint8 conversion that wraps instead of saturating): the kernel both
    product paths use (PNG files: v2e_amd.SuperSloMo.interpolate; in HBM: VideoToEvents) gives the same bytes."""
    import ctypes as C
    import torch
    from v2e_amd import _capi
    from v2e_amd._capi import check
    lib = _capi.lib()
    rng = np.random.Generator(np.random.PCG64(77))
    U, B, h, w = 3, 2, 8, 40
    x = (rng.integers(0, 1 << 16, size=(U, B, 1, h, w)).astype(np.float32) / np.float32(65536.0) * np.float32(2.6) - np.float32(1.2))
    x.flat[:6] = [-0.428, 1.0 - 0.428, 0.0, 1.0, -0.4280001, 255.9999 / 255 - 0.428]  # 0, 1 and their neighbours after + mean
    t = torch.from_numpy(x)
    expect = (t + 0.428).mul(255).byte().numpy()                      # torchvision F.normalize(mean=-0.428) + to_pil_image
    assert (x + 0.428 < 0).any() and (x + 0.428 > 1).any()
    dev = torch.device("cuda")
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    td = t.to(dev)
    q = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    check(lib.v2e_f32_to_u8_trunc(C.c_void_p(td.data_ptr()), C.c_void_p(q.data_ptr()), U, B, h * w, 0.428, 0, s), "trunc")
    assert np.array_equal(q.cpu().numpy(), expect)
    q2 = torch.empty((B * U, h, w), dtype=torch.uint8, device=dev)    # time-ordered variant of the PNG-free pipeline
    check(lib.v2e_f32_to_u8_trunc(C.c_void_p(td.data_ptr()), C.c_void_p(q2.data_ptr()), U, B, h * w, 0.428, 1, s), "trunc")
    assert np.array_equal(q2.cpu().numpy().reshape(B, U, h, w), expect[:, :, 0].transpose(1, 0, 2, 3))


@pytest.mark.gpu
def test_product_paths_match_reference_class_pngs(tmp_path):
    """The frames the REFERENCE class wrote (v2ecore.slomo.SuperSloMo.interpolate, tests/golden/make_golden_slomo_class.py)
    against both product paths: the drop-in class (PNG files) and VideoToEvents.upsample (in HBM).  uint8 frames: at most
    one grey level on < 1 % of the pixels (a 1e-7 float difference can cross a truncation boundary)."""
    import os
    import torch
    from v2e_amd import EventEmulator, SuperSloMo
    from v2e_amd.pipeline import VideoToEvents
    from v2e_amd.synth import portable_unet_state_dict
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slomo_class_40x70.npz"))
    fr, ref, U = z["frames"], z["pngs"], int(z["U"])
    n, Hs, Ws = fr.shape
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    for i, f in enumerate(fr):
        np.save(str(src / ("%08d.npy" % i)), f)
    sf, si = (int(v) for v in z["seeds"])
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"state_dictFC": {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, sf).items()},
                "state_dictAT": {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, si).items()}}, str(ckpt))
    sm = SuperSloMo(model=str(ckpt), auto_upsample=False, upsampling_factor=U, batch_size=2)
    times, avg = sm.interpolate(str(src), str(dst), (Ws, Hs))
    assert np.array_equal(times, z["times"]) and avg == float(z["avg"])
    pngs = np.stack([np.asarray(Image.open(str(dst / ("%d.png" % i)))) for i in range((n - 1) * U)])
    pipe = VideoToEvents(sm.engine, EventEmulator(device="cuda", seed=5, rng_mode="philox"), U, batch_size=2)
    up = pipe.upsample(torch.from_numpy(fr).cuda()).cpu().numpy()
    for got in (pngs, up):
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01
    assert np.array_equal(pngs, up)
