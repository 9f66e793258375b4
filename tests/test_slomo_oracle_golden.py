"""CPU: the SloMo oracle (oracle/slomo_oracle.c) against golden vectors produced by the
reference's own modules (tests/golden/make_golden_slomo.py).  Tolerance: BASELINE north_star
'float intermediates within 1e-5', stated as |a-b| <= 1e-5 * max(1,|b|)."""
import os

import numpy as np

from fixtures import GOLDEN

TOL = 1e-5


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def load_pairs(z):
    fr = z["frames"]
    n = len(fr) - 1
    I0 = (fr[:n].astype(np.float32) / np.float32(255.0))[:, None] - np.float32(0.428)
    I1 = (fr[1:n + 1].astype(np.float32) / np.float32(255.0))[:, None] - np.float32(0.428)
    return I0, I1


def test_oracle_unet_and_interpolation_match_reference(oracle_lib):
    from v2e_amd.synth import portable_unet_state_dict
    z = np.load(os.path.join(GOLDEN, "slomo_unet_64x96.npz"))
    I0, I1 = load_pairs(z)
    ts = list(z["ts"])
    o = oracle_lib.slomo_interpolate(I0, I1, ts, portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102))
    assert close(o["flow"], z["flow"]) < TOL
    nt, b = len(ts), I0.shape[0]
    assert close(o["intrp"].reshape(nt, b, 5, 64, 96), z["intrp"]) < TOL
    assert close(o["Ft"], z["Ft"]) < TOL


def test_oracle_warp_blend_fusion_match_reference(oracle_lib):
    z = np.load(os.path.join(GOLDEN, "slomo_warp_64x96.npz"))
    I0, I1 = load_pairs(z)
    ts = list(z["ts"])
    x12 = oracle_lib.slomo_prep(I0, I1, z["flow"], ts)
    assert close(x12.reshape(len(ts), 1, 12, 64, 96)[:, :, 6:12], z["x12_tail"]) < TOL
    intrp = z["intrp"].reshape(len(ts) * 1, 5, 64, 96)
    Ft = oracle_lib.slomo_fuse(I0, I1, x12, intrp, ts)
    assert close(Ft, z["Ft"]) < TOL
