"""The centre-surround DVS (cs_lambda_pixels; emulator.py:245-272, 707-716, 753-754, 1061-1124).

CPU: the oracle's restatement of `_update_csdvs`'s stepping loop against the reference's digests (csdvs_steps.npz: all
num_steps steps, and loops that end early on max_change <= 1e-5; float32 and float64 state) and, where the reference tree is
present, against the live reference on fresh planes.
GPU: k_cs_step (v2e_csdvs_update) against the same digests and against the oracle on odd sizes (edges, partial blocks,
one-row and one-column planes) bit for bit; the drop-in class with cs_lambda_pixels against reference-generated fixtures --
bit for bit at DAVIS346 (float64 and float32 state: the reference's convolution sums in the order the kernel fixes), and within
1e-5 on the surround plane at a size where the reference's convolution backend sums in another order
(tests/golden/make_golden_csdvs.py measures which sizes those are)."""
import os

import numpy as np
import pytest

from fixtures import CSDVS_STEP_CASES, GOLDEN, PhiloxFixture, csdvs_step_case, sha

STOP = 1e-5


def _steps_fixture():
    return np.load(os.path.join(GOLDEN, "csdvs_steps.npz"))


@pytest.mark.parametrize("name", CSDVS_STEP_CASES)
def test_oracle_stepping_loop_matches_reference_digests(name, oracle_lib):
    z = _steps_fixture()
    p, h = csdvs_step_case(name)
    a = z[name + "_alpha"]
    steps, last = oracle_lib.csdvs_update(p, h, float(a[0]), float(a[1]), int(z[name + "_num_steps"]), STOP)
    assert steps == int(z[name + "_steps"])
    assert sha(h) == str(z[name + "_sha"])
    if name.endswith("early"):
        assert last <= STOP and steps < int(z[name + "_num_steps"])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_oracle_stepping_loop_equals_live_reference(dt, oracle_lib):
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference tree not present")
    import logging
    import torch
    logging.disable(logging.CRITICAL)
    torch.set_num_threads(1)
    EE = rh.ref_emulator_cls()
    rng = np.random.default_rng(77)
    H, W = 260, 346  # the reference's float32 conv sum is in kernel order from 200 x 200 up (make_golden_csdvs.py)
    p = (rng.standard_normal((H, W)) * 0.3 + 2).astype(dt)
    h0 = (p + rng.standard_normal((H, W)) * 0.02).astype(dt)
    lam, tau_p_ms, delta_time = 4.0, 3.0, 1 / 250
    ref = EE(seed=1, device="cpu", cs_lambda_pixels=lam, cs_tau_p_ms=tau_p_ms)
    ref.lp_log_frame, ref.cs_surround_frame = torch.from_numpy(p.copy()), torch.from_numpy(h0.copy())
    ref._update_csdvs(delta_time)
    tau_p, tau_h = tau_p_ms * 1e-3, (tau_p_ms / lam ** 2) * 1e-3
    num_steps = int(np.ceil((delta_time / min(tau_p, tau_h)) * 5))
    adt = delta_time / num_steps
    h = h0.copy()
    steps, _ = oracle_lib.csdvs_update(p, h, adt / tau_p, adt / tau_h, num_steps, STOP)
    assert steps == ref.cs_steps_taken[-1]
    assert np.array_equal(h, ref.cs_surround_frame.numpy())


def _device_update(p, h0, alpha_p, alpha_h, num_steps):
    import ctypes as C
    import torch
    from v2e_amd import _capi
    lib = _capi.lib()
    dev = torch.device("cuda")
    pd, hd = torch.from_numpy(p).to(dev), torch.from_numpy(h0).to(dev)
    scratch = torch.empty_like(hd)
    steps, last = C.c_int(0), C.c_double(0)
    H, W = p.shape
    _capi.check(lib.v2e_csdvs_update(pd.data_ptr(), hd.data_ptr(), scratch.data_ptr(), H, W, 1 if p.dtype == np.float64 else 0,
                                     float(alpha_p), float(alpha_h), int(num_steps), STOP, C.byref(steps), C.byref(last),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "v2e_csdvs_update")
    return hd.cpu().numpy(), steps.value, last.value


@pytest.mark.gpu
@pytest.mark.parametrize("name", CSDVS_STEP_CASES)
def test_device_stepping_loop_matches_reference_digests(name):
    z = _steps_fixture()
    p, h0 = csdvs_step_case(name)
    a = z[name + "_alpha"]
    h, steps, last = _device_update(p, h0, a[0], a[1], int(z[name + "_num_steps"]))
    assert steps == int(z[name + "_steps"])
    assert sha(h) == str(z[name + "_sha"])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (1, 70), (67, 1), (33, 37), (97, 131), (260, 346)])
def test_device_stepping_loop_matches_oracle(shape, dt, oracle_lib):
    """Edges (replication padding), planes smaller than a workgroup, odd step counts (the result ends in the scratch plane),
    a loop that ends inside a chunk of launches and one that ends exactly on num_steps."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for num_steps, noise in ((5, 0.05), (33, 0.05), (64, 0.05), (200, 2e-4)):
        p = (rng.standard_normal(shape) * 0.4 + 2.5).astype(dt)
        h0 = (p + rng.standard_normal(shape) * noise).astype(dt)
        h_or = h0.copy()
        osteps, olast = oracle_lib.csdvs_update(p, h_or, 0.03, 0.21, num_steps, STOP)
        h, steps, last = _device_update(p, h0, 0.03, 0.21, num_steps)
        assert steps == osteps and last == olast, (shape, num_steps)
        assert np.array_equal(h, h_or), (shape, num_steps)


def _run_emulator(fx):
    from v2e_amd import EventEmulator
    emu = EventEmulator(device="cuda", seed=fx.seed, rng_mode="philox", **fx.kw)
    evs = [emu.generate_events(f, float(t)) for f, t in zip(fx.frames, fx.times)]
    return emu, evs


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["philox_csdvs_346x260", "philox_csdvs_f32_346x260"])
def test_csdvs_emulator_matches_reference_at_davis346(name):
    fx = PhiloxFixture(name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    emu, evs = _run_emulator(fx)
    assert emu.cs_steps_taken == list(z["cs_steps"])
    assert sha(emu.cs_surround_frame.cpu().numpy()) == str(z["cs_surround_sha"])
    assert [0 if e is None else len(e) for e in evs] == list(fx.n_events)
    for k, e in enumerate(evs):
        if e is not None:
            assert sha(e) == fx.ev_sha[k], "frame %d event digest differs" % k
    assert sha(emu.lp_log_frame.cpu().numpy()) == fx.lp_sha
    assert sha(emu.base_log_frame.cpu().numpy()) == fx.base_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)
    assert emu.base_log_frame.dtype == emu.cs_surround_frame.dtype == emu.lp_log_frame.dtype


@pytest.mark.gpu
def test_csdvs_emulator_close_to_reference_where_its_conv_sums_differently():
    """97 x 131: torch's CPU convolution sums the five float32 terms in another order there (a last-bit difference per
    step in h_conv): the surround plane stays within 1e-5 (it is 1e-7 in practice) and so do the events, up to pixels
    whose brightness change sits within that of a threshold."""
    fx = PhiloxFixture("philox_csdvs_97x131")
    z = np.load(os.path.join(GOLDEN, "philox_csdvs_97x131.npz"))
    emu, evs = _run_emulator(fx)
    assert emu.cs_steps_taken == list(z["cs_steps"])
    sur = emu.cs_surround_frame.cpu().numpy()
    assert np.max(np.abs(sur - z["cs_surround_final"])) <= 1e-5
    assert np.array_equal(emu.lp_log_frame.cpu().numpy().shape, z["cs_surround_final"].shape)
    n_ref, n_got = int(fx.n_events.sum()), sum(0 if e is None else len(e) for e in evs)
    assert abs(n_got - n_ref) <= max(2, n_ref // 100)
    assert np.max(np.abs(emu.base_log_frame.cpu().numpy() - z["base_final"])) <= 0.5  # at most a threshold step apart anywhere


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [True, 0])
@pytest.mark.parametrize("name", ["philox_csdvs_346x260", "philox_csdvs_f32_346x260"])
def test_csdvs_device_resident_clip_matches_reference_at_davis346(name, use_graph):
    """generate_events_batch with the centre-surround pixel (round-3 review, missing 4): the diffuser's stepping loop is enqueued
    whole before every frame of the run, its stop rule evaluated on the device -- events, surround plane, step counts and state
    against the reference-generated DAVIS346 fixtures, bit for bit; the clip fed in two runs (the surround carries over), with
    and without the run's hipGraph."""
    from v2e_amd import EventEmulator
    fx = PhiloxFixture(name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    emu = EventEmulator(device="cuda", seed=fx.seed, rng_mode="philox", **fx.kw)
    n = len(fx.frames)
    cut = n // 2
    ev1, c1 = emu.generate_events_batch(fx.frames[:cut], fx.times[:cut], use_graph=use_graph)
    ev2, c2 = emu.generate_events_batch(fx.frames[cut:], fx.times[cut:], use_graph=use_graph)
    counts = list(c1) + list(c2)
    assert counts == list(fx.n_events)
    for ev, cs, k0 in ((ev1, c1, 0), (ev2, c2, cut)):
        row = 0
        for k, m in enumerate(cs):
            if m:
                assert sha(ev[row:row + m]) == fx.ev_sha[k0 + k], "frame %d event digest differs" % (k0 + k)
            row += m
    assert emu.cs_steps_taken == list(z["cs_steps"])
    assert sha(emu.cs_surround_frame.cpu().numpy()) == str(z["cs_surround_sha"])
    assert sha(emu.lp_log_frame.cpu().numpy()) == fx.lp_sha
    assert sha(emu.base_log_frame.cpu().numpy()) == fx.base_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.gpu
def test_csdvs_device_resident_clip_matches_oracle_and_refuses_endless_loops(oracle_lib):
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    H, W = 33, 37
    frames = int_gradient_frames(9, H, W, seed=71, noise=8, as_array=True)
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.2, shot_noise_rate_hz=2.0,
              refractory_period_s=0.0005, cs_lambda_pixels=2.5, cs_tau_p_ms=3.0)
    emu = EventEmulator(device="cuda", seed=8, rng_mode="philox", **kw)
    ora = oracle_lib.OracleEmulator(seed=8, rng_mode="philox", **kw)
    ev, counts = emu.generate_events_batch(frames, [i / 250 for i in range(9)])
    ref = [ora.generate_events(frames[i], i / 250) for i in range(9)]
    ref_all = np.concatenate([e for e in ref if e is not None])
    assert list(counts) == [0 if e is None else len(e) for e in ref] and np.array_equal(ev, ref_all)
    assert emu.cs_steps_taken == ora.cs_steps_taken
    assert np.array_equal(emu.cs_surround_frame.cpu().numpy(), ora.cs_surround_frame)
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    # cs_tau_p_ms = 0: a nanosecond time constant, i.e. millions of Euler steps per frame that only the per-frame loop's early
    # exit makes tractable -- the device-resident run says so instead of enqueueing them
    emu2 = EventEmulator(device="cuda", seed=1, rng_mode="philox", cs_lambda_pixels=3.0, cs_tau_p_ms=0)
    with pytest.raises(ValueError, match="Euler steps per frame"):
        emu2.generate_events_batch(np.full((3, 40, 48), 100, np.uint8), [0, 0.01, 0.02])


@pytest.mark.gpu
def test_csdvs_default_tape_mode_matches_reference_at_davis346():
    """The drop-in's default mode (the reference's own seeded torch generator) with the centre-surround pixel: event digests,
    surround plane, step counts and lp_log_frame against the unmodified reference."""
    import torch
    from fixtures import LiveTapeFixture, require_same_generator
    from v2e_amd import EventEmulator
    fx = LiveTapeFixture("tape_live_csdvs_346x260")
    require_same_generator(fx)
    z = np.load(os.path.join(GOLDEN, "tape_live_csdvs_346x260.npz"))
    emu = EventEmulator(device="cuda", seed=fx.seed, **fx.kw)
    assert emu.rng_mode == "tape" and emu.csdvs_enabled
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    assert emu.cs_steps_taken == list(z["cs_steps"])
    assert sha(emu.cs_surround_frame.cpu().numpy()) == str(z["cs_surround_sha"])
    assert sha(emu.lp_log_frame.cpu().numpy()) == fx.lp_sha
    if fx.host_exp_matches():
        assert sha(emu.base_log_frame.cpu().numpy()) == fx.base_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)
    emu.cleanup()


@pytest.mark.parametrize("name", ["philox_csdvs_346x260", "philox_csdvs_f32_346x260"])
def test_oracle_csdvs_emulator_matches_reference_at_davis346(name, oracle_lib):
    """The oracle's whole centre-surround path (lp preview -> stepping loop -> counts against the surround) against the
    reference-generated DAVIS346 fixtures, bit for bit (CPU)."""
    fx = PhiloxFixture(name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ora = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="philox", **fx.kw)
    evs = [ora.generate_events(f, float(t)) for f, t in zip(fx.frames, fx.times)]
    assert ora.cs_steps_taken == list(z["cs_steps"])
    assert sha(ora.cs_surround_frame) == str(z["cs_surround_sha"])
    assert [0 if e is None else len(e) for e in evs] == list(fx.n_events)
    for k, e in enumerate(evs):
        if e is not None:
            assert sha(e) == fx.ev_sha[k], "frame %d event digest differs" % k
    assert sha(ora.lp_log_frame) == fx.lp_sha and sha(ora.base_log_frame) == fx.base_sha


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff", [300, 0])
@pytest.mark.parametrize("shape", [(33, 37), (97, 131), (64, 96)])
def test_csdvs_emulator_matches_oracle_at_any_size(shape, cutoff, oracle_lib):
    """Where the reference's convolution backend sums in another order the HIP path is pinned by the oracle instead (same
    fixed order): events, surround, state planes bit for bit, float64 and float32 state, ragged sizes."""
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    H, W = shape
    frames = int_gradient_frames(9, H, W, seed=71, noise=8, as_array=True)
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=cutoff, leak_rate_hz=.2, shot_noise_rate_hz=2.0 if cutoff else 0.0,
              refractory_period_s=0.0005, cs_lambda_pixels=2.5, cs_tau_p_ms=3.0)
    emu = EventEmulator(device="cuda", seed=8, rng_mode="philox", **kw)
    ora = oracle_lib.OracleEmulator(seed=8, rng_mode="philox", **kw)
    for i in range(9):
        e, o = emu.generate_events(frames[i], i / 250), ora.generate_events(frames[i], i / 250)
        assert (e is None) == (o is None) and (e is None or np.array_equal(e, o)), "frame %d" % i
    assert emu.cs_steps_taken == ora.cs_steps_taken
    assert np.array_equal(emu.cs_surround_frame.cpu().numpy(), ora.cs_surround_frame)
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.lp_log_frame.cpu().numpy(), ora.lp_log_frame)
