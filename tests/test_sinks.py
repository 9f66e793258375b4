"""Event-sink integer formats: the HIP packers against bytes produced by the
reference's own AEDAT-2.0 writer and its HDF5 conversion lines (tests/golden/make_golden_sinks.py)."""
import os

import numpy as np
import pytest

from fixtures import GOLDEN


def test_golden_is_consistent_with_documented_rules():
    """CPU: restate aedat2_output.py:155-173 / emulator.py:957-960 in numpy and compare with the recorded bytes."""
    z = np.load(os.path.join(GOLDEN, "sinks.npz"))
    for (w, h, xs, ys, ps) in ((346, 260, 12, 22, 11), (640, 480, 1, 11, 0)):
        ev = z["ev_%dx%d" % (w, h)]
        t = (1e6 * ev[:, 0]).astype(np.int32)
        x = (w - 1) - ev[:, 1].astype(np.int32)
        y = (h - 1) - ev[:, 2].astype(np.int32)
        p = ((ev[:, 3] + 1) / 2).astype(np.int32)
        a = (x << xs | y << ys | p << ps)
        a[4000:] |= 1 << 10
        out = np.empty(2 * len(ev), np.int32)
        out[0::2], out[1::2] = a, t
        assert np.array_equal(np.frombuffer(out.byteswap().tobytes(), np.uint8), z["aedat2_%dx%d" % (w, h)])


@pytest.mark.gpu
def test_hip_packers_match_reference_writer_bytes():
    import torch
    from v2e_amd.sinks import pack_aedat2, pack_h5
    z = np.load(os.path.join(GOLDEN, "sinks.npz"))
    for (w, h) in ((346, 260), (640, 480)):
        ev = torch.from_numpy(z["ev_%dx%d" % (w, h)]).cuda()
        got = pack_aedat2(ev, w, h, noise_from=4000).cpu().numpy()
        assert np.array_equal(got, z["aedat2_%dx%d" % (w, h)])
        h5 = pack_h5(ev).cpu().numpy().view(np.uint32)
        assert np.array_equal(h5, z["h5_%dx%d" % (w, h)])
    with pytest.raises(ValueError):
        pack_aedat2(ev, 100, 100)


@pytest.mark.gpu
def test_event_frame_accumulator_matches_renderer_math():
    """renderer.py:368-400 restated with numpy (hist2d_numba_seq semantics), incl. down-scaled output bins."""
    import torch
    from v2e_amd.sinks import EventFrameAccumulator
    z = np.load(os.path.join(GOLDEN, "sinks.npz"))
    ev = z["ev_346x260"]
    for (bh, bw) in ((260, 346), (130, 173)):
        acc = EventFrameAccumulator(bh, bw, full_scale_count=3, device="cuda")
        cur = np.zeros((bh, bw))
        for sl in (slice(0, 2000), slice(2000, 5000)):
            e = ev[sl]
            got = acc.accumulate(torch.from_numpy(e).cuda(), 260, 346).cpu().numpy()
            on = e[:, 3] == 1
            H = np.zeros((bh, bw))
            for pol, sgn in ((on, 1), (~on, -1)):
                i = (e[pol, 2].astype(np.float64) - 0) * (1 / ((260 - 0) / bh))
                j = (e[pol, 1].astype(np.float64) - 0) * (1 / ((346 - 0) / bw))
                ok = (i >= 0) & (i < bh) & (j >= 0) & (j < bw)
                np.add.at(H, (i[ok].astype(int), j[ok].astype(int)), sgn)
            cur = np.clip(cur + H, -3, 3)
            assert np.array_equal(got, cur)
