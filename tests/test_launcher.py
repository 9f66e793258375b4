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
