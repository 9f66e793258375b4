"""CPU: the oracle (oracle/emu_oracle.c) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py).  Bit-exact: integer and float32 event
rows, final float64/float32 state planes."""
import numpy as np
import pytest

from fixtures import (PHILOX_FIXTURES, PHILOX_HD_FIXTURES, TAPE_FIXTURES, TAPE_LIVE_FIXTURES, LiveTapeFixture, PhiloxFixture, TapeFixture,
                      events_equal, sha)


@pytest.mark.parametrize("name", TAPE_FIXTURES)
def test_oracle_replays_reference_tape(name, oracle_lib):
    fx = TapeFixture(name)
    emu = oracle_lib.OracleEmulator(seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items), **fx.kw)
    if fx.preset:
        emu.set_dvs_params(fx.preset)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        assert events_equal(ev, fx.events[k]), "frame %d differs" % k
    assert emu.tape.pos == len(fx.items), "tape not fully consumed"
    assert emu.base_log_frame.dtype == fx.base_final.dtype
    assert np.array_equal(emu.base_log_frame, fx.base_final)
    assert np.array_equal(emu.lp_log_frame, fx.lp_final)
    if fx.ts_mem_final is not None:
        assert np.array_equal(emu.timestamp_mem, fx.ts_mem_final)
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("name", PHILOX_FIXTURES + PHILOX_HD_FIXTURES)
def test_oracle_philox_matches_reference_with_philox_source(name, oracle_lib):
    fx = PhiloxFixture(name)
    emu = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="philox", **fx.kw)
    if fx.preset:
        emu.set_dvs_params(fx.preset)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    assert sha(emu.base_log_frame) == fx.base_sha
    assert sha(emu.lp_log_frame) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(emu.timestamp_mem) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("name", TAPE_LIVE_FIXTURES)
def test_oracle_live_tape_at_sensor_size(name, oracle_lib):
    """The reference's own seeded MT19937 stream at 346x260 / 1280x720 and BASELINE configs[0] in full (500 frames,
    8 435 events): the oracle draws live from torch with the same seed and must give the reference's events."""
    import torch
    fx = LiveTapeFixture(name)
    if not fx.generator_matches():
        pytest.skip("torch %s draws differently from the fixture's torch %s" % (torch.__version__, fx.torch_version))
    torch.set_num_threads(1)
    emu = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="tape", **fx.kw)
    if fx.preset:
        emu.set_dvs_params(fx.preset)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    if fx.host_exp_matches():
        assert sha(emu.base_log_frame) == fx.base_sha
    assert sha(emu.lp_log_frame) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(emu.timestamp_mem) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)
    if "moving_dot" in name:
        assert emu.num_events_total == 8435


def test_oracle_scidvs_replays_reference_tape(oracle_lib):
    """scidvs=True (emulator.py:56-80, 719-725, 747; float64 state): the reference's recorded torch draws, incl. the per-pixel
    time-constant normal and its exp.  Events bit for bit; scidvs_highpass to 1e-12 (torch's vectorised float64 sinh and libm's
    differ in the last bit on a few percent of the arguments, which never reaches an event)."""
    import os
    from fixtures import GOLDEN
    fx = TapeFixture("tape_scidvs_40x48")
    emu = oracle_lib.OracleEmulator(seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items), **fx.kw)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        assert events_equal(emu.generate_events(f, float(t)), fx.events[k]), "frame %d differs" % k
    assert emu.tape.pos == len(fx.items), "tape not fully consumed"
    z = np.load(os.path.join(GOLDEN, "tape_scidvs_40x48.npz"))
    assert np.array_equal(emu.scidvs_tau_arr, z["scidvs_tau"])
    assert np.max(np.abs(emu.scidvs_highpass - z["scidvs_highpass_final"])) <= 1e-12 * max(1.0, np.max(np.abs(z["scidvs_highpass_final"])))
    assert np.array_equal(emu.lp_log_frame, fx.lp_final)


def test_oracle_scidvs_philox_matches_reference(oracle_lib):
    import os
    from fixtures import GOLDEN
    fx = PhiloxFixture("philox_scidvs_97x131")
    emu = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="philox", **fx.kw)
    evs = [emu.generate_events(f, float(t)) for f, t in zip(fx.frames, fx.times)]
    assert [0 if e is None else len(e) for e in evs] == list(fx.n_events)
    for k, e in enumerate(evs):
        assert events_equal(e, fx.events[k] if len(fx.events[k]) else None), "frame %d differs" % k
    z = np.load(os.path.join(GOLDEN, "philox_scidvs_97x131.npz"))
    assert np.max(np.abs(emu.scidvs_highpass - z["scidvs_highpass_final"])) <= 1e-12 * max(1.0, np.max(np.abs(z["scidvs_highpass_final"])))
    assert np.max(np.abs(emu.base_log_frame - z["base_final"])) <= 1e-12 * max(1.0, np.max(np.abs(z["base_final"])))
