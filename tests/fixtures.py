"""Helpers to load tests/golden/*.npz (see tests/golden/make_golden.py)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TAPE_FIXTURES = ["tape_defaults_40x48", "tape_noisy_40x48", "tape_clean_40x48", "tape_f32state_40x48",
                 "tape_scalarthres_40x48", "tape_refractory_float_33x37", "tape_moving_dot_64x64_40fr",
                 "tape_hdr_40x48", "tape_hdr_nocutoff_40x48",  # hdr: tests/golden/make_golden_hdr.py
                 "tape_pnoise_40x48"]  # photoreceptor noise: tests/golden/make_golden_pnoise.py
PHILOX_FIXTURES = ["philox_moving_dot_64x64", "philox_defaults_346x260", "philox_noisy_346x260",
                   "philox_refractory_346x260", "philox_noisy_1280x720", "philox_hdr_97x131",
                   "philox_pnoise_97x131"]
# 1280x720 runs long enough for full 32-frame chain launches, a validating successor and a wrap of the ring of frame
# slots (104 frames, noisy preset), and with a refractory period of 2 ms (26 frames): tests/golden/make_golden_hd_long.py
PHILOX_HD_FIXTURES = ["philox_noisy_1280x720_long", "philox_refractory_1280x720"]
PNOISE_VRMS = 0.03125  # the noise amplitude the photoreceptor-noise fixtures were generated with


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class TapeFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.name = name
        self.frames = z["frames"]
        self.times = z["times"]
        self.kw = json.loads(str(z["kw"]))
        if self.kw.get("photoreceptor_noise"):
            self.kw["photoreceptor_noise_vrms"] = PNOISE_VRMS
        self.preset = str(z["preset"]) or None
        self.seed = int(z["seed"])
        n = int(z["n_items"])
        keys = sorted(k for k in z.files if k.startswith("tape_"))
        assert len(keys) == n
        self.items = [(k.split("_", 2)[2], z[k]) for k in keys]
        self.events = [z["ev_%04d" % i] for i in range(len(self.frames))]
        self.base_final = z["base_final"]
        self.lp_final = z["lp_final"]
        self.ts_mem_final = z["ts_mem_final"] if "ts_mem_final" in z.files else None
        self.counters = z["counters"]


class PhiloxFixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.name = name
        self.times = z["times"]
        self.kw = json.loads(str(z["kw"]))
        if self.kw.get("photoreceptor_noise"):
            self.kw["photoreceptor_noise_vrms"] = PNOISE_VRMS
        self.preset = str(z["preset"]) or None
        self.seed = int(z["seed"])
        self.n_events = z["n_events"]
        self.ev_sha = [str(s) for s in z["ev_sha"]]
        self.base_sha = str(z["base_sha"])
        self.lp_sha = str(z["lp_sha"])
        self.ts_mem_sha = str(z["ts_mem_sha"]) if "ts_mem_sha" in z.files else None
        self.counters = z["counters"]
        self.shape = tuple(int(v) for v in z["shape"])
        spec = json.loads(str(z["frame_spec"]))
        if "frames" in z.files:
            self.frames = z["frames"]
        else:
            from v2e_amd.synth import int_gradient_frames
            assert spec["gen"] == "int_gradient_frames"
            self.frames = int_gradient_frames(spec["n"], spec["H"], spec["W"], seed=spec["seed"],
                                              noise=spec["noise"], as_array=True)
        self.events = None
        if "ev_0000" in z.files:
            self.events = [z["ev_%04d" % i] for i in range(len(self.times))]


TAPE_LIVE_FIXTURES = ["tape_live_defaults_346x260", "tape_live_noisy_346x260", "tape_live_noisy_1280x720",
                      "tape_live_moving_dot_64x64"]  # tests/golden/make_golden_tape_live.py


class LiveTapeFixture(PhiloxFixture):
    """Digest-only fixture of the reference run with its own seeded torch generator (the drop-in's default mode)."""

    def __init__(self, name):
        super().__init__(name)
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.draw_probe = str(z["draw_probe"])
        self.exp_probe = str(z["exp_probe"])
        self.torch_version = str(z["torch_version"])

    def generator_matches(self):
        """True iff this host's torch produces the draws the fixture was generated with (same MT19937 consumption by
        normal / randn / rand / randperm / linspace); otherwise the replay cannot be compared and the test is skipped."""
        import hashlib
        import torch
        torch.manual_seed(self.seed)
        h = hashlib.sha256()
        for a in (torch.normal(0.2, 0.03, size=(7, 11), dtype=torch.float32), torch.randn((5, 13), dtype=torch.float32),
                  torch.rand(size=(3, 17), dtype=torch.float32), torch.randperm(1000), torch.randperm(70000),
                  torch.linspace(start=0.0123, end=0.0456, steps=7, dtype=torch.float32)):
            h.update(np.ascontiguousarray(a.numpy()).tobytes())
        return h.hexdigest() == self.draw_probe

    def host_exp_matches(self):
        """torch.exp (float32, CPU: noise_rate_array, emulator.py:504) gives the fixture host's bits on this host; where it
        does not (its last bit depends on the CPU), base_log_frame differs in the last bits from the stored digest."""
        import hashlib
        import torch
        torch.manual_seed(5)
        r = torch.randn((64, 64), dtype=torch.float32)
        return hashlib.sha256(torch.exp(0.23025850929940458 * r).numpy().tobytes()).hexdigest() == self.exp_probe


TAPE_PORTABLE_FIXTURES = ["tape_portable_defaults_346x260", "tape_portable_noisy_346x260"]  # tests/golden/make_golden_tape_portable.py


class PortableTapeFixture(PhiloxFixture):
    """Digest-only fixture of the reference run with tests/golden/portable_tape.PortableSource as its random source: pins the
    tape-mode path at sensor size on ANY torch build (the draws are integer hashing; see portable_tape.py)."""

    def tape(self):
        from portable_tape import PortableTape
        return PortableTape(self.seed)


def require_same_generator(fx):
    """The tape_live_* fixtures replay torch's own MT19937 stream: on a torch build that draws differently they cannot be compared.
    That used to be a silent skip (round-4 review: 'parity evidence that can evaporate silently'); now it is a FAILURE unless
    V2E_ALLOW_TORCH_DRIFT=1 says the difference is known -- the tape_portable_* fixtures pin the same kernels without torch's
    generator either way."""
    import os
    import pytest
    import torch
    if fx.generator_matches():
        return
    msg = ("torch %s draws differently from the fixture's torch %s: the default (tape) mode of the drop-in cannot be compared with "
           "the reference's recorded run on this host" % (torch.__version__, fx.torch_version))
    if os.environ.get("V2E_ALLOW_TORCH_DRIFT") == "1":
        pytest.skip(msg + " (V2E_ALLOW_TORCH_DRIFT=1)")
    pytest.fail(msg + "; set V2E_ALLOW_TORCH_DRIFT=1 to accept (tape_portable_* still pin the kernels)")


def events_equal(a, b):
    if a is None:
        a = np.zeros((0, 4), np.float32)
    if b is None:
        b = np.zeros((0, 4), np.float32)
    return a.shape == b.shape and np.array_equal(a, b)


# ---- `_update_csdvs` in isolation (tests/golden/make_golden_csdvs.py: csdvs_steps.npz holds the reference's digests)
CSDVS_STEP_SHAPE = (200, 208)
CSDVS_STEP_CASES = ["f32", "f64", "f32_early", "f64_early"]


def csdvs_step_case(name):
    """(photoreceptor plane p, surround plane h0) of a step case: seeded noise, or (`*_early`) a nearly settled diffuser
    whose loop ends on max_change <= 1e-5 well before num_steps."""
    H, W = CSDVS_STEP_SHAPE
    dt = np.float32 if name.startswith("f32") else np.float64
    if name.endswith("early"):
        yy, xx = np.mgrid[0:H, 0:W]
        # (+ - * / only: correctly rounded everywhere, unlike numpy's vectorised sin / exp)
        p = (3 + 0.002 * ((xx % 34) / 17.0 - 1) * ((yy % 46) / 23.0 - 1)).astype(dt)
        h0 = (p + 2e-4 / (1 + ((xx - 90) ** 2 + (yy - 70) ** 2) / 50.0)).astype(dt)
    else:
        rng = np.random.default_rng(5 if dt == np.float32 else 6)
        p = (rng.standard_normal((H, W)) * 0.5 + 3).astype(dt)
        h0 = (p + rng.standard_normal((H, W)) * 0.05).astype(dt)
    return p, h0
