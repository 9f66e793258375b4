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


def events_equal(a, b):
    if a is None:
        a = np.zeros((0, 4), np.float32)
    if b is None:
        b = np.zeros((0, 4), np.float32)
    return a.shape == b.shape and np.array_equal(a, b)
