"""CPU: the drop-in binding of INTEGRATION.md as code.  The reference's v2e.py is imported (non-hot-path imports stubbed:
cv2, argcomplete, ...), v2e_amd.launcher.bind() rebinds the two class names, and the classes are constructed with the
keyword sets v2e.py itself uses (v2e.py:471-478, 544-563).  Without a GPU the first call that needs the device raises
V2EAmdError (there is no CPU fallback).  Skipped where the reference tree does not exist (the GPU box)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_harness as rh  # noqa: E402

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


def test_launcher_binds_the_drop_in_into_v2e(tmp_path):
    import torch
    import v2e_amd
    from v2e_amd.launcher import bind
    rh.install_stubs()
    rh.install_torchvision_stub()
    if "argcomplete" not in sys.modules:
        rh._mod("argcomplete", __getattr__=rh._ga)
    import v2ecore.emulator
    import v2ecore.slomo
    saved = (v2ecore.emulator.EventEmulator, v2ecore.slomo.SuperSloMo)
    try:
        _check_binding(bind(rh.REF_ROOT), tmp_path)
    finally:  # other tests of this process import the reference's own classes
        v2ecore.emulator.EventEmulator, v2ecore.slomo.SuperSloMo = saved
        v2e = sys.modules.get("v2e")
        if v2e is not None:
            v2e.EventEmulator, v2e.SuperSloMo = saved


def _check_binding(v2e, tmp_path):
    import torch
    import v2e_amd
    import v2ecore.emulator
    import v2ecore.slomo
    assert v2e.EventEmulator is v2e_amd.EventEmulator and v2ecore.emulator.EventEmulator is v2e_amd.EventEmulator
    assert v2e.SuperSloMo is v2e_amd.SuperSloMo and v2ecore.slomo.SuperSloMo is v2e_amd.SuperSloMo
    assert callable(v2e.main)
    # class attributes v2ecore/v2e_args.py:209,216 reads
    assert v2e.EventEmulator.MODEL_STATES and v2e.EventEmulator.SINGLE_PIXEL_STATES_FILENAME
    # v2e.py:544-563, keyword for keyword (the CLI defaults)
    emu = v2e.EventEmulator(
        pos_thres=0.2, neg_thres=0.2, sigma_thres=0.03, cutoff_hz=300, leak_rate_hz=0.01, shot_noise_rate_hz=0.001,
        photoreceptor_noise=False, leak_jitter_fraction=0.1, noise_rate_cov_decades=0.1, refractory_period_s=0.0005, seed=0,
        output_folder=str(tmp_path), dvs_h5=None, dvs_aedat2=None, dvs_aedat4=None, dvs_text=None, show_dvs_model_state=None,
        save_dvs_model_state=False, output_width=346, output_height=260, device="cuda" if torch.cuda.is_available() else "cpu",
        cs_lambda_pixels=None, cs_tau_p_ms=None, hdr=False, scidvs=False, record_single_pixel_states=None,
        label_signal_noise=False)
    emu.set_dvs_params("noisy")  # v2e.py:565-570
    assert emu.t_previous == 0 and emu.num_events_total == 0
    # v2e.py:471-478
    sm = v2e.SuperSloMo(model="no-such-checkpoint.pth", auto_upsample=False, upsampling_factor=10, video_path=None, vid_orig=None,
                        vid_slomo=None, preview=False, batch_size=8)
    with pytest.raises(ValueError):
        v2e.SuperSloMo(model="x", auto_upsample=False, upsampling_factor=1)  # slomo.py:91-94
    if not torch.cuda.is_available():
        with pytest.raises(v2e_amd.V2EAmdError):
            emu.generate_events(np.zeros((260, 346), np.uint8), 0.0)
        with pytest.raises(v2e_amd.V2EAmdError):
            sm.interpolate(str(tmp_path), str(tmp_path), (346, 260))


class _FakeCapture:
    """cv2.VideoCapture over generated grey 346x260 frames (the size the CLI asks for, monochrome: stage 1 then neither
    resizes nor converts -- v2e.py:713-731)."""
    N = 6

    def __init__(self, path):
        self.i = 0

    def get(self, prop):
        return {"fps": 30.0, "count": self.N, "w": 346, "h": 260, "mono": 1}[prop]

    def read(self):
        if self.i >= self.N:
            return False, None
        self.i += 1
        yy, xx = np.mgrid[0:260, 0:346]
        return True, ((xx + 3 * self.i + yy) % 256).astype(np.uint8)

    def release(self):
        pass


def test_v2e_main_with_the_cli_defaults_reaches_the_first_device_call(tmp_path, monkeypatch):
    """v2e.main() as a user starts it -- video output ON (no --skip_video_output), preview ON (no --no_preview): v2e.py:471-478
    then passes video_path / vid_orig / vid_slomo / preview to SuperSloMo.  The run must get through both constructors and
    stage 1 and stop only where the device is first needed (here, without a GPU: V2EAmdError from SuperSloMo.interpolate)."""
    import torch
    import v2e_amd
    from v2e_amd.launcher import bind
    rh.install_stubs()
    rh.install_torchvision_stub()
    if "argcomplete" not in sys.modules:
        rh._mod("argcomplete", __getattr__=rh._ga)
    import v2ecore.emulator
    import v2ecore.slomo
    saved = (v2ecore.emulator.EventEmulator, v2ecore.slomo.SuperSloMo)
    video = tmp_path / "in.avi"
    video.write_bytes(b"not a real video: the capture below is a fake")
    out = tmp_path / "out"
    built = {}
    try:
        v2e = bind(rh.REF_ROOT)
        cv2 = sys.modules["cv2"]
        for k, v in dict(VideoCapture=_FakeCapture, CAP_PROP_FPS="fps", CAP_PROP_FRAME_COUNT="count", CAP_PROP_FRAME_WIDTH="w",
                         CAP_PROP_FRAME_HEIGHT="h", CAP_PROP_MONOCHROME="mono").items():
            monkeypatch.setattr(cv2, k, v, raising=False)
        real_slomo = v2e.SuperSloMo

        def spy(*a, **k):
            built["slomo_kwargs"] = dict(k)
            built["slomo"] = real_slomo(*a, **k)
            return built["slomo"]
        monkeypatch.setattr(v2e, "SuperSloMo", spy)
        monkeypatch.setattr(v2e, "v2e_quit", lambda *a: (_ for _ in ()).throw(SystemExit(a[0] if a else 0)), raising=False)
        monkeypatch.setattr(sys, "argv", ["v2e.py", "-i", str(video), "--output_folder", str(out), "--overwrite", "--dvs346",
                                          "--ignore-gooey"])
        if torch.cuda.is_available():
            with pytest.raises(FileNotFoundError):  # the default checkpoint input/SuperSloMo39.ckpt is not here
                v2e.main()
        else:
            with pytest.raises(v2e_amd.V2EAmdError, match="ROCm GPU"):
                v2e.main()
    finally:
        v2ecore.emulator.EventEmulator, v2ecore.slomo.SuperSloMo = saved
        v2e = sys.modules.get("v2e")
        if v2e is not None:
            v2e.EventEmulator, v2e.SuperSloMo = saved
    kw = built["slomo_kwargs"]
    assert kw["video_path"] == str(out) and kw["vid_orig"] and kw["vid_slomo"] and kw["preview"] is True  # the CLI defaults
    sm = built["slomo"]
    assert sm.video_path == str(out) and sm.preview is True
    sm.cleanup()


def test_slomo_constructor_accepts_the_video_keywords():
    """slomo.py:44-54: every keyword of the reference constructor; none of them raises (the AVI writers are host work that is
    skipped with a warning when OpenCV is absent)."""
    import v2e_amd
    sm = v2e_amd.SuperSloMo(model="x.pth", auto_upsample=True, upsampling_factor=2, batch_size=4, video_path="/tmp",
                            vid_orig="original.avi", vid_slomo="slomo.avi", preview=True, avi_frame_rate=30)
    assert (sm.video_path, sm.vid_orig, sm.vid_slomo, sm.preview, sm.avi_frame_rate) == ("/tmp", "original.avi", "slomo.avi", True, 30)
    sm._open_writers((346, 260))  # no cv2 here: one warning, no writers, no exception
    sm.cleanup()
