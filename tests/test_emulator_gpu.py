"""GPU parity tests of the HIP emulator (through the C ABI, via v2e_amd.EventEmulator /
EmuEngine) against (a) golden vectors recorded from the reference and (b) the CPU
oracle on seeded inputs.  Everything here is bit-exact: float32 event rows (t,x,y,p),
float64/float32 state planes, integer counters."""
import numpy as np
import pytest
import torch

from fixtures import (PHILOX_FIXTURES, TAPE_FIXTURES, TAPE_LIVE_FIXTURES, TAPE_PORTABLE_FIXTURES, LiveTapeFixture, PhiloxFixture,
                      PortableTapeFixture, TapeFixture, require_same_generator,
                      events_equal, sha)

pytestmark = pytest.mark.gpu


def _mk(fx, **extra):
    from v2e_amd import EventEmulator
    emu = EventEmulator(device="cuda", **fx.kw, **extra)
    if fx.preset:
        emu.set_dvs_params(fx.preset)
    return emu


def _state(emu):
    return {k: (None if getattr(emu, k) is None else getattr(emu, k).cpu().numpy())
            for k in ("base_log_frame", "lp_log_frame", "timestamp_mem")}


@pytest.mark.parametrize("name", TAPE_FIXTURES)
def test_hip_replays_reference_tape(name, oracle_lib):
    """Same random numbers as the reference (its recorded torch draws) -> same events, same order."""
    fx = TapeFixture(name)
    emu = _mk(fx, seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items))
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        assert events_equal(ev, fx.events[k]), "frame %d differs from the reference" % k
    assert emu._tape.pos == len(fx.items)
    st = _state(emu)
    assert st["base_log_frame"].dtype == fx.base_final.dtype
    assert np.array_equal(st["base_log_frame"], fx.base_final)
    assert np.array_equal(st["lp_log_frame"], fx.lp_final)
    if fx.ts_mem_final is not None:
        assert np.array_equal(st["timestamp_mem"], fx.ts_mem_final)
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("name", TAPE_PORTABLE_FIXTURES)
def test_hip_tape_mode_at_sensor_size_with_a_portable_random_source(name):
    """The tape-mode kernels (count -> rank / scan -> host permutations -> shot -> emit -> permute) at 346x260 against the
    reference run with the SAME torch-independent random source (tests/golden/portable_tape.py): compared on every host, whatever
    its torch draws."""
    fx = PortableTapeFixture(name)
    emu = _mk(fx, seed=fx.seed, rng_mode="tape", tape=fx.tape())
    emu.noise_rate_cov_decades = 0.0  # (the 'noisy' preset sets 0.1, emulator.py:535; the fixtures keep exp() out: exp(0 * r) = 1)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    st = _state(emu)
    assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["lp_log_frame"]) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(st["timestamp_mem"]) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("name", TAPE_LIVE_FIXTURES)
def test_hip_default_mode_at_sensor_size(name, oracle_lib):
    """The drop-in exactly as v2e.py would construct it (rng_mode default = tape, the reference's seeded MT19937
    stream drawn live on the host) at 346x260 and 1280x720 -- multi-workgroup scans, k_permute and the host randperm
    gather at full size -- and BASELINE configs[0] in full (500 frames, 8 435 events): the reference's events, frame by
    frame.  Final state: bit-equal to the oracle run on this host with the same seed, and to the reference's digest
    where this host's torch.exp (noise_rate_array, emulator.py:504) has the fixture host's last bits."""
    fx = LiveTapeFixture(name)
    require_same_generator(fx)
    emu = _mk(fx, seed=fx.seed)
    assert emu.rng_mode == "tape"
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    st = _state(emu)
    ora = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="tape", **fx.kw)
    if fx.preset:
        ora.set_dvs_params(fx.preset)
    for f, t in zip(fx.frames, fx.times):
        ora.generate_events(f, float(t))
    assert np.array_equal(st["base_log_frame"].view(np.uint64), ora.base_log_frame.view(np.uint64))
    if fx.host_exp_matches():
        assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["lp_log_frame"]) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(st["timestamp_mem"]) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)
    if "moving_dot" in name:
        assert emu.num_events_total == 8435


@pytest.mark.parametrize("name", PHILOX_FIXTURES)
def test_hip_philox_frame_api_matches_reference(name):
    """Philox mode, one generate_events call per frame, vs the reference fed the same Philox numbers."""
    fx = PhiloxFixture(name)
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k], "frame %d: %d events, reference %d" % (k, n, fx.n_events[k])
        if n:
            assert sha(ev) == fx.ev_sha[k], "frame %d event digest differs" % k
    st = _state(emu)
    assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["lp_log_frame"]) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(st["timestamp_mem"]) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


# use_graph: low bits 0 plain launches / 1 hipGraph; no pipeline bit: k_chain (K frames per launch, state in registers)
# wherever it can run, |256 insists on it, |128 one frame per launch; |16 the unfused count/rank/scan/emit kernels
PIPELINES = [0, 1, 256, 257, 128, 129, 16, 17]


@pytest.mark.parametrize("use_graph", PIPELINES)
@pytest.mark.parametrize("name", [n for n in PHILOX_FIXTURES if "pnoise" not in n])  # photoreceptor noise: its own test below
def test_hip_philox_device_resident_clip_matches_reference(name, use_graph):
    """Whole clip on device (no host sync between frames; optionally one hipGraph), every pipeline."""
    fx = PhiloxFixture(name)
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=use_graph)
    assert list(counts) == list(fx.n_events)
    row = 0
    for k, n in enumerate(counts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    st = _state(emu)
    assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["lp_log_frame"]) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(st["timestamp_mem"]) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("use_graph", [True, 0, 1, 16, 17])
def test_hip_philox_device_resident_photoreceptor_noise(use_graph):
    """photoreceptor_noise=True (emulator.py:694-703) in the device-resident API: the noise plane lives in HBM across the
    run (and across runs: the clip is fed in two pieces), events and state equal the reference-generated fixture."""
    from v2e_amd._capi import V2EAmdError
    name = "philox_pnoise_97x131"
    fx = PhiloxFixture(name)
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    cut = len(fx.frames) // 2 + 1
    ev0, c0 = emu.generate_events_batch(fx.frames[:cut], fx.times[:cut], use_graph=use_graph)
    ev1, c1 = emu.generate_events_batch(fx.frames[cut:], fx.times[cut:], use_graph=use_graph)
    counts = list(c0) + list(c1)
    assert counts == list(fx.n_events)
    ev = np.concatenate([e for e in (ev0, ev1) if e is not None and len(e)])
    row = 0
    for k, n in enumerate(counts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    st = _state(emu)
    assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["lp_log_frame"]) == fx.lp_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)
    with pytest.raises(V2EAmdError):  # the chain pipelines do not carry the noise plane: refused, not silently noiseless
        emu.generate_events_batch(fx.frames[-2:], [fx.times[-1] + 0.01, fx.times[-1] + 0.02], use_graph=257)


def test_scidvs_replays_reference_tape(oracle_lib):
    """scidvs=True (emulator.py:56-80, 719-725, 747; float64 state): the reference's recorded torch draws (incl. the
    per-pixel time-constant normal and its exp) -> the reference's events, frame by frame, bit for bit; base / lp planes bit
    for bit; scidvs_highpass to 1e-12 (torch's vectorised float64 sinh and the device's differ in the last bit)."""
    import os
    from fixtures import GOLDEN
    fx = TapeFixture("tape_scidvs_40x48")
    emu = _mk(fx, seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items))
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        ev = emu.generate_events(f, float(t))
        assert events_equal(ev, fx.events[k]), "frame %d differs from the reference" % k
    assert emu._tape.pos == len(fx.items)
    st = _state(emu)
    assert np.array_equal(st["base_log_frame"], fx.base_final) and np.array_equal(st["lp_log_frame"], fx.lp_final)
    z = np.load(os.path.join(GOLDEN, "tape_scidvs_40x48.npz"))
    assert np.array_equal(emu.scidvs_tau_arr.cpu().numpy(), z["scidvs_tau"])
    hp = emu.scidvs_highpass.cpu().numpy()
    assert np.max(np.abs(hp - z["scidvs_highpass_final"])) <= 1e-12 * max(1.0, np.max(np.abs(z["scidvs_highpass_final"])))
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("api", ["frame", "clip", "clip_graph"])
def test_scidvs_philox_matches_reference(api):
    """scidvs=True in Philox mode (time constants from the portable streams), frame-at-a-time and device-resident (the count /
    rank / scan / emit kernels carry the two extra state planes), against the reference fed the same numbers."""
    import os
    from fixtures import GOLDEN
    from v2e_amd._capi import V2EAmdError
    fx = PhiloxFixture("philox_scidvs_97x131")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    if api == "frame":
        evs = [emu.generate_events(f, float(t)) for f, t in zip(fx.frames, fx.times)]
        counts = [0 if e is None else len(e) for e in evs]
        ev = np.concatenate([e for e in evs if e is not None])
    else:
        ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=(api == "clip_graph"))
        assert emu._engine.last_pipeline()[0] == "k_count/k_rank/k_scan/k_emit"
    assert list(counts) == list(fx.n_events)
    assert np.array_equal(ev, np.concatenate([e for e in fx.events if len(e)]))
    z = np.load(os.path.join(GOLDEN, "philox_scidvs_97x131.npz"))
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), z["base_final"])
    hp = emu.scidvs_highpass.cpu().numpy()
    assert np.max(np.abs(hp - z["scidvs_highpass_final"])) <= 1e-12 * max(1.0, np.max(np.abs(z["scidvs_highpass_final"])))
    if api == "clip":
        with pytest.raises(V2EAmdError):  # k_chain does not carry the SCIDVS planes: refused, not silently a plain DVS
            emu.generate_events_batch(fx.frames[-2:], [fx.times[-1] + 0.01, fx.times[-1] + 0.02], use_graph=257)


def test_scidvs_float32_state_replays_reference_tape(oracle_lib):
    """scidvs=True with FLOAT32 pixel state (cutoff_hz = 0): torch's float32 sinh (Sleef sinhf_u10) is restated bit for bit in
    include/v2e_detmath.h and shared by the kernel and the oracle.  40 x 48: the reference has no scalar tail there, so events,
    state planes AND scidvs_highpass are bit for bit."""
    import os
    from fixtures import GOLDEN
    fx = TapeFixture("tape_scidvs_f32_40x48")
    emu = _mk(fx, seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items))
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        assert events_equal(emu.generate_events(f, float(t)), fx.events[k]), "frame %d differs from the reference" % k
    st = _state(emu)
    assert st["base_log_frame"].dtype == np.float32
    assert np.array_equal(st["base_log_frame"], fx.base_final) and np.array_equal(st["lp_log_frame"], fx.lp_final)
    z = np.load(os.path.join(GOLDEN, "tape_scidvs_f32_40x48.npz"))
    assert np.array_equal(emu.scidvs_highpass.cpu().numpy(), z["scidvs_highpass_final"])
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("api", ["frame", "clip_graph"])
def test_scidvs_float32_state_philox_matches_reference_and_oracle(api, oracle_lib):
    """97 x 131 (12 707 pixels: the reference hands the plane's last three to glibc's scalar sinhf): events bit for bit against
    the reference; scidvs_highpass bit for bit against the reference on the vector body and against the oracle everywhere."""
    import os
    from fixtures import GOLDEN
    from test_oracle_golden import _tail_free
    fx = PhiloxFixture("philox_scidvs_f32_97x131")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    if api == "frame":
        evs = [emu.generate_events(f, float(t)) for f, t in zip(fx.frames, fx.times)]
        counts = [0 if e is None else len(e) for e in evs]
        ev = np.concatenate([e for e in evs if e is not None])
    else:
        ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=True)
    assert list(counts) == list(fx.n_events)
    assert np.array_equal(ev, np.concatenate([e for e in fx.events if len(e)]))
    z = np.load(os.path.join(GOLDEN, "philox_scidvs_f32_97x131.npz"))
    hp = emu.scidvs_highpass.cpu().numpy()
    assert hp.dtype == np.float32 and _tail_free(hp, z["scidvs_highpass_final"])
    ora = oracle_lib.OracleEmulator(seed=fx.seed, rng_mode="philox", **fx.kw)
    for f, t in zip(fx.frames, fx.times):
        ora.generate_events(f, float(t))
    assert np.array_equal(hp, ora.scidvs_highpass) and np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)


@pytest.mark.parametrize("chain_k", [1, 2, 3, 5, 8, 11, 16, 32, 40, 64])
@pytest.mark.parametrize("name", ["philox_refractory_346x260", "philox_noisy_346x260", "philox_defaults_346x260"])
def test_chain_launch_lengths_and_ring_wrap(name, chain_k, monkeypatch):
    """k_chain with few frames per launch: the ring of 3 K frame slots wraps several times within the fixture clip,
    launches wait on emission batches, partial last launch; with the refractory fixture every launch redoes its
    predecessor (rule on in most frames), several passes per launch once K > 1."""
    monkeypatch.setenv("V2E_AMD_CHAIN_K", str(chain_k))
    monkeypatch.setenv("V2E_AMD_CHAIN_M", "1")  # emission batches of K frames: a ring of 3 K slots
    fx = PhiloxFixture(name)
    for use_graph in (256, 257):
        emu = _mk(fx, seed=fx.seed, rng_mode="philox")
        ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=use_graph)
        assert list(counts) == list(fx.n_events)
        row = 0
        for k, n in enumerate(counts):
            if n:
                assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
            row += n
        st = _state(emu)
        assert sha(st["base_log_frame"]) == fx.base_sha
        assert sha(st["lp_log_frame"]) == fx.lp_sha
        if fx.ts_mem_sha:
            assert sha(st["timestamp_mem"]) == fx.ts_mem_sha


@pytest.mark.parametrize("variant", ["push", "pull_two_level"])
@pytest.mark.parametrize("name", ["philox_refractory_346x260", "philox_noisy_346x260", "philox_defaults_346x260"])
def test_event_writer_variants_write_the_same_rows(name, variant, monkeypatch):
    """The chain's event rows are written by k_cpull (a thread per output row: inverse bijection, prefix search, bit select; every
    other test of this file runs it with the whole prefix row in LDS).  Its two-level search (what frames beyond 260 000 pixels
    take) and the push writer k_cemit (the fallback when the pull's tables do not fit) must write the same bytes."""
    if variant == "push":
        monkeypatch.setenv("V2E_AMD_EMIT_PULL", "0")
    else:
        monkeypatch.setenv("V2E_AMD_PULL_TWO_LEVEL", "1")
    fx = PhiloxFixture(name)
    for use_graph in (256, 257):
        emu = _mk(fx, seed=fx.seed, rng_mode="philox")
        ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=use_graph)
        assert list(counts) == list(fx.n_events)
        row = 0
        for k, n in enumerate(counts):
            if n:
                assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
            row += n
        assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("variant", ["push", "pull_two_level"])
def test_event_writer_variants_many_iterations(variant, oracle_lib, monkeypatch):
    """> 64 events per pixel and frame (many keys and iterations with a handful of rows each, rule on): both writer variants."""
    from v2e_amd import EventEmulator
    if variant == "push":
        monkeypatch.setenv("V2E_AMD_EMIT_PULL", "0")
    else:
        monkeypatch.setenv("V2E_AMD_PULL_TWO_LEVEL", "1")
    kw = dict(pos_thres=0.03, neg_thres=0.04, sigma_thres=0.01, cutoff_hz=200, leak_rate_hz=0.2, shot_noise_rate_hz=30.0,
              refractory_period_s=0.0004)
    rng = np.random.Generator(np.random.PCG64(17))
    frames = [rng.integers(0, 256, size=(37, 91)).astype(np.uint8) for _ in range(7)]
    times = [0.01 * i for i in range(7)]
    ora = oracle_lib.OracleEmulator(rng_mode="philox", seed=21, **kw)
    ref = np.concatenate([e for e in (ora.generate_events(f, t) for f, t in zip(frames, times)) if e is not None])
    hip = EventEmulator(device="cuda", rng_mode="philox", seed=21, max_iters=1024, **kw)
    ev, counts = hip.generate_events_batch(np.stack(frames), times, use_graph=257, cap=6_000_000)
    assert np.array_equal(ev, ref)


@pytest.mark.parametrize("refr", [0.0005, 0.002])
def test_pipelines_agree_on_benchmark_clip(refr, oracle_lib):
    """BASELINE configs[1] at full size (346x260, 300 frames, dt = 1/300 s): every device-resident pipeline gives the
    same event stream, records and final state.  refr = 0.5 ms: the chain mis-speculates on ~1 % of the frames; 2 ms: on
    most of them (redo passes on every launch)."""
    from v2e_amd import EventEmulator
    from v2e_amd.synth import sincos_gradient_frames
    F = 300
    frames = sincos_gradient_frames(F + 1, 260, 346, seed=1)
    times = [i / 300 for i in range(F + 1)]
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01, shot_noise_rate_hz=.001,
              refractory_period_s=refr)
    ref = None
    for use_graph in (257, 129, 17):  # k_chain 32 frames per launch, one frame per launch, count/rank/scan/emit
        emu = EventEmulator(device="cuda", seed=1, rng_mode="philox", **kw)
        ev, counts = emu.generate_events_batch(frames, times, use_graph=use_graph)
        st = (sha(ev), list(counts), sha(emu.base_log_frame.cpu().numpy()), sha(emu.timestamp_mem.cpu().numpy()),
              sha(emu.lp_log_frame.cpu().numpy()))
        if ref is None:
            ref = st
            assert counts.sum() > 5_000_000
        else:
            assert st == ref, "pipeline use_graph=%d differs" % use_graph
    # ... and it is the CPU oracle's stream, event for event (about 10 M events)
    ora = oracle_lib.OracleEmulator(seed=1, rng_mode="philox", **kw)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    assert ref[1] == [0 if e is None else len(e) for e in oev]
    assert ref[0] == sha(np.concatenate([e for e in oev if e is not None]))
    assert ref[2] == sha(ora.base_log_frame)


@pytest.mark.parametrize("chunk", [1, 2, 3, 5])
def test_clip_in_small_runs_equals_whole_clip(chunk):
    """Runs of 1, 2, 3, 5 frames chained through the carried state == one run (every run ends in a tail launch that
    copies the state back when the number of launches is odd)."""
    fx = PhiloxFixture("philox_refractory_346x260")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    evs, cnts = [], []
    for lo in range(0, len(fx.frames), chunk):
        ev, c = emu.generate_events_batch(fx.frames[lo:lo + chunk], fx.times[lo:lo + chunk])
        if ev is not None:
            evs.append(ev)
        cnts += list(c)
    assert cnts == list(fx.n_events)
    ev = np.concatenate(evs)
    row = 0
    for k, n in enumerate(cnts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    st = _state(emu)
    assert sha(st["base_log_frame"]) == fx.base_sha
    assert sha(st["timestamp_mem"]) == fx.ts_mem_sha


def test_split_clip_equals_whole_clip():
    """Two consecutive device-resident runs == one run (state and frame counter carry over)."""
    fx = PhiloxFixture("philox_refractory_346x260")
    a = _mk(fx, seed=fx.seed, rng_mode="philox")
    ev_a, cnt_a = a.generate_events_batch(fx.frames, fx.times)
    b = _mk(fx, seed=fx.seed, rng_mode="philox")
    ev1, c1 = b.generate_events_batch(fx.frames[:9], fx.times[:9])
    ev2, c2 = b.generate_events_batch(fx.frames[9:], fx.times[9:])
    assert list(cnt_a) == list(c1) + list(c2)
    assert np.array_equal(ev_a, np.concatenate([ev1, ev2]))


@pytest.mark.parametrize("cfg", [
    dict(H=97, W=131, dtype="u8", kw=dict(cutoff_hz=300, leak_rate_hz=0.3, shot_noise_rate_hz=3.0, refractory_period_s=0.001)),
    dict(H=64, W=64, dtype="f32", kw=dict(cutoff_hz=0, leak_rate_hz=0.3, shot_noise_rate_hz=3.0, refractory_period_s=0.004)),
    dict(H=50, W=70, dtype="f64", kw=dict(cutoff_hz=100, leak_rate_hz=0.0, shot_noise_rate_hz=0.0, refractory_period_s=0.0, sigma_thres=0.0)),
    dict(H=1, W=1, dtype="u8", kw=dict(cutoff_hz=300, leak_rate_hz=0.1, shot_noise_rate_hz=50.0, refractory_period_s=0.0005)),
    dict(H=3, W=65, dtype="u8", kw=dict(cutoff_hz=0, leak_rate_hz=0.0, shot_noise_rate_hz=0.0, refractory_period_s=0.0)),
])
@pytest.mark.parametrize("mode", ["philox", "tape"])
def test_hip_matches_oracle_on_seeded_inputs(cfg, mode, oracle_lib):
    """Ragged sizes (not multiples of the wave), every frame dtype, float frames, big per-pixel
    counts (steps of ~100 grey levels -> many iterations, refractory active)."""
    from v2e_amd import EventEmulator
    H, W = cfg["H"], cfg["W"]
    rng = np.random.Generator(np.random.PCG64(5))
    frames = []
    base = rng.integers(0, 256, size=(H, W))
    for i in range(10):
        if i % 3 == 2:
            base = rng.integers(0, 256, size=(H, W))  # large jumps: many events per pixel
        else:
            base = np.clip(base + rng.integers(-12, 13, size=(H, W)), 0, 255)
        f = base.astype(np.float64)
        if cfg["dtype"] == "f32":
            f = (f * 0.37 + 0.123).astype(np.float32)
        elif cfg["dtype"] == "f64":
            f = f * 0.991 + 0.0625
        else:
            f = f.astype(np.uint8)
        frames.append(f)
    times = [0.002 + 0.005 * i for i in range(10)]
    kw = dict(pos_thres=0.2, neg_thres=0.15, sigma_thres=0.03)
    kw.update(cfg["kw"])
    torch.manual_seed(123)
    gen_state = torch.random.get_rng_state()
    ora = oracle_lib.OracleEmulator(seed=0 if mode == "tape" else 77, rng_mode=mode, **kw)
    if mode == "tape":
        torch.random.set_rng_state(gen_state)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    hip = EventEmulator(device="cuda", seed=0 if mode == "tape" else 77, rng_mode=mode, **kw)
    if mode == "tape":
        torch.random.set_rng_state(gen_state)
    for k, (f, t) in enumerate(zip(frames, times)):
        ev = hip.generate_events(f, t)
        assert events_equal(ev, oev[k]), "frame %d differs from the oracle" % k
    assert np.array_equal(hip.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(hip.lp_log_frame.cpu().numpy(), ora.lp_log_frame)
    if kw["refractory_period_s"] > 0:
        assert np.array_equal(hip.timestamp_mem.cpu().numpy(), ora.timestamp_mem)
    assert hip.num_events_total == ora.num_events_total and hip.num_events_on == ora.num_events_on
    assert ora.num_events_total > 0 or (H * W == 1)


def test_philox_init_planes_match_oracle(oracle_lib):
    """Device Box-Muller / exp (include/v2e_detmath.h) == host, bit for bit."""
    from v2e_amd import EventEmulator
    H, W = 260, 346
    emu = EventEmulator(device="cuda", seed=99, rng_mode="philox", cutoff_hz=300, leak_rate_hz=0.1,
                        sigma_thres=0.03, refractory_period_s=0.0005)
    f = np.full((H, W), 100, np.uint8)
    emu.generate_events(f, 0.0)
    ora = oracle_lib.OracleEmulator(seed=99, rng_mode="philox", cutoff_hz=300, leak_rate_hz=0.1, sigma_thres=0.03,
                                    refractory_period_s=0.0005)
    ora.generate_events(f, 0.0)
    assert np.array_equal(emu.pos_thres.cpu().numpy(), ora.pos_thres_arr)
    assert np.array_equal(emu.neg_thres.cpu().numpy(), ora.neg_thres_arr)
    assert np.array_equal(emu.noise_rate_array.cpu().numpy(), ora.noise_rate_array)
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    # sanity of the distribution itself
    z = (emu.pos_thres.cpu().numpy() - 0.2) / 0.03
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02


@pytest.mark.parametrize("shape", [(60, 100, 8, 3, 0.002), (260, 346, 12, 64, 0.002), (260, 346, 12, 64, 0.0005), (480, 640, 6, 2, 0.0)])
def test_multi_clip_engine_equals_single_clips(shape, oracle_lib):
    """n_clips pixel arrays advanced by one launch == independent single-clip runs (Philox clip streams): 3 small clips;
    64 clips of 346x260 with a refractory period (bench.py's `batched` workload: more workgroups than are co-resident,
    clips walked a few at a time, redo passes with rendezvous per clip); 2 clips of 640x480 without one."""
    from v2e_amd.emulator import EventEmulator
    from v2e_amd.engine import EmuEngine
    from v2e_amd.synth import int_gradient_frames
    H, W, F, NC, refr = shape
    clips = [int_gradient_frames(F, H, W, seed=20 + c, noise=8, as_array=True) for c in range(NC)]
    times = [i / 300 for i in range(F)]
    kw = dict(cutoff_hz=300, leak_rate_hz=0.2, shot_noise_rate_hz=4.0, refractory_period_s=refr)
    proto = EventEmulator(device="cuda", seed=5, rng_mode="philox", **kw)
    proto._thres_scalar = (0.2, 0.2)
    proto._thres_is_scalar = False
    P = proto._params()
    eng = EmuEngine(H, W, n_clips=NC, device="cuda")
    eng.alloc_state(True)
    frames = torch.from_numpy(np.stack(clips, axis=1)).cuda()  # [F][NC][H][W]
    eng.init_state(P, frames[0].contiguous(), times[0])
    cap = 4 * H * W * F
    ev = eng.event_buffer(cap)
    recs = eng.alloc_recs(F - 1)
    t_prev = np.array([[0.0] * NC] + [[times[f]] * NC for f in range(1, F - 1)])
    t_frame = np.array([[times[f]] * NC for f in range(1, F)])
    eng.run(P, frames[1:].contiguous(), t_prev, t_frame, 1, ev, recs, use_graph=True)
    assert eng.last_pipeline()[0].startswith("k_chain")
    r = eng.recs_to_numpy(recs)
    assert not r["flags"].any()
    for c in (range(NC) if NC <= 3 else (0, 1, 7, 31, 62, 63)):
        ora = oracle_lib.OracleEmulator(seed=5, rng_mode="philox", clip=c, **kw)
        oev = [ora.generate_events(clips[c][f], times[f]) for f in range(F)]
        ref = np.concatenate([e for e in oev if e is not None])
        n = int(r["n_events"][:, c].sum())
        assert n == len(ref), "clip %d" % c
        assert np.array_equal(ev[c, :n].cpu().numpy(), ref), "clip %d" % c
        assert np.array_equal(eng.plane(eng.base, c).cpu().numpy(), ora.base_log_frame)
        if refr > 0:
            assert np.array_equal(eng.plane(eng.ts_mem, c).cpu().numpy(), ora.timestamp_mem)


def test_run_to_run_determinism():
    fx = PhiloxFixture("philox_noisy_346x260")
    outs = []
    for _ in range(2):
        emu = _mk(fx, seed=fx.seed, rng_mode="philox")
        ev, _ = emu.generate_events_batch(fx.frames, fx.times)
        outs.append(sha(ev))
    assert outs[0] == outs[1]


def test_event_buffer_overflow_is_reported():
    from v2e_amd import V2EAmdError
    fx = PhiloxFixture("philox_defaults_346x260")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    with pytest.raises(V2EAmdError, match="capacity"):
        emu.generate_events_batch(fx.frames, fx.times, cap=1000)
    # the pixel state is past the failed run: further frames are refused until reset(), after which the clip runs again
    with pytest.raises(V2EAmdError, match="reset"):
        emu.generate_events_batch(fx.frames[-2:], [fx.times[-1] + 0.01, fx.times[-1] + 0.02])
    with pytest.raises(V2EAmdError, match="reset"):
        emu.generate_events(fx.frames[-1], float(fx.times[-1]) + 0.03)
    emu.reset()
    emu.t_previous = 0
    ev, counts = emu.generate_events_batch(fx.frames, fx.times)
    assert list(counts) == list(fx.n_events)


def test_many_iterations_grow_scratch(oracle_lib):
    """A 0->255 step with a tiny threshold gives > max_iters events per pixel; the frame API grows its scratch."""
    from v2e_amd import EventEmulator
    kw = dict(pos_thres=0.02, neg_thres=0.02, sigma_thres=0.0, cutoff_hz=0, leak_rate_hz=0, shot_noise_rate_hz=0,
              refractory_period_s=0)
    f0 = np.zeros((8, 70), np.uint8)
    f1 = np.full((8, 70), 255, np.uint8)
    hip = EventEmulator(device="cuda", rng_mode="philox", seed=3, max_iters=16, **kw)
    ora = oracle_lib.OracleEmulator(rng_mode="philox", seed=3, **kw)
    for f, t in ((f0, 0.0), (f1, 0.01), (f0, 0.02)):
        a = hip.generate_events(f, t)
        b = ora.generate_events(f, t)
        assert events_equal(a, b)
    assert ora.last["M"] > 64


@pytest.mark.parametrize("use_graph", [257, 129, 17])
@pytest.mark.parametrize("refr", [0.0, 0.0004])
def test_device_resident_clip_many_iterations(use_graph, refr, oracle_lib):
    """> 31 events per pixel per frame: several 64-key chunks in k_ctot, many iterations in k_cpull, refractory on/off."""
    from v2e_amd import EventEmulator
    kw = dict(pos_thres=0.03, neg_thres=0.04, sigma_thres=0.01, cutoff_hz=200, leak_rate_hz=0.2, shot_noise_rate_hz=30.0,
              refractory_period_s=refr)
    rng = np.random.Generator(np.random.PCG64(17))
    H, W = 37, 91
    frames = [rng.integers(0, 256, size=(H, W)).astype(np.uint8) for _ in range(7)]
    times = [0.01 * i for i in range(7)]
    hip = EventEmulator(device="cuda", rng_mode="philox", seed=21, max_iters=1024, **kw)
    ev, counts = hip.generate_events_batch(np.stack(frames), times, use_graph=use_graph, cap=6_000_000)
    ora = oracle_lib.OracleEmulator(rng_mode="philox", seed=21, **kw)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    assert ora.last["M"] > 64
    ref = np.concatenate([e for e in oev if e is not None])
    assert list(counts) == [0 if e is None else len(e) for e in oev]
    assert np.array_equal(ev, ref)
    assert np.array_equal(hip.base_log_frame.cpu().numpy(), ora.base_log_frame)
    if refr > 0:
        assert np.array_equal(hip.timestamp_mem.cpu().numpy(), ora.timestamp_mem)


def test_event_wire_format_roundtrip():
    """v2e_events_pack64 / unpack64: real event rows survive the 8-byte wire format bit for bit, and the device
    packer agrees with the CPU one used by the gloo tests."""
    from v2e_amd.dist import pack_events64, unpack_events64
    fx = PhiloxFixture("philox_moving_dot_64x64")
    ev = np.concatenate([e for e in fx.events if len(e)])
    extra = np.array([[-0.25, 1279, 719, -1], [3.5e-4, 0, 0, 1], [1e9, 16383, 16383, 1]], np.float32)  # corner values
    ev = np.concatenate([ev, extra]).astype(np.float32)
    d = torch.from_numpy(ev).cuda()
    w = pack_events64(d)
    assert torch.equal(w.cpu(), pack_events64(torch.from_numpy(ev)))
    back = unpack_events64(w)
    assert np.array_equal(back.cpu().numpy().view(np.uint32), ev.view(np.uint32))
    # the 4-byte format (payload + one time stamp per block): device kernels == the CPU implementation == the rows
    from v2e_amd.dist import pack_events32, unpack_events32
    ev32 = ev[:-1].copy()  # (the 16383 corner value does not fit 11 + 10 bits)
    d32 = torch.from_numpy(ev32).cuda()
    pl, runs, flags = pack_events32(d32, cap_runs=len(ev32))
    plc, runsc, _ = pack_events32(torch.from_numpy(ev32))
    assert int(flags[0].item()) == 0 and int(runs[0].item()) == int(runsc[0])
    R = int(runsc[0])
    assert torch.equal(pl[:len(ev32)].cpu(), plc) and torch.equal(runs[:R + 1].cpu(), runsc)
    back32 = unpack_events32(pl, runs, len(ev32))
    assert np.array_equal(back32.cpu().numpy().view(np.uint32), ev32.view(np.uint32))
    assert int(pack_events32(d, cap_runs=len(ev))[2][0].item()) & 1  # 16383 flagged
    assert int(pack_events32(d32, cap_runs=3)[2][0].item()) & 2      # run table too small flagged


def test_event_stream_gatherer_nccl_single_rank():
    """The RCCL code path of EventStreamGatherer (side stream, all_gather_into_tensor) with world_size 1."""
    import os
    import torch.distributed as dist
    from v2e_amd.dist import EventStreamGatherer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g = EventStreamGatherer(torch.device("cuda", 0), 1)
        for n in (1000, 0, 37):
            k = torch.arange(n + 5, dtype=torch.float32, device="cuda")
            ev = torch.stack([k * 1e-3 - 0.002, k % 346, (k * 7) % 260, (k % 2) * 2 - 1], dim=1).contiguous()  # (t, x, y, p) rows
            g.submit(ev, n)
            parts = g.result()
            assert len(parts) == 1 and parts[0].shape == (n, 4)
            assert torch.equal(parts[0], ev[:n])
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("api", ["frame", "clip"])
def test_all_grey_levels_static_scene_and_tensor_input(api, oracle_lib):
    """Every uint8 grey level (0..20 linear branch of lin_log, log(0) masked), a static scene (no signal
    events -> None / zero counts), torch-tensor frames (test/leak_event_test.py passes tensors), reset()."""
    from v2e_amd import EventEmulator
    kw = dict(pos_thres=0.2, neg_thres=0.2, sigma_thres=0.03, cutoff_hz=300, leak_rate_hz=0.0, shot_noise_rate_hz=0.0,
              refractory_period_s=0.0005)
    ramp = np.tile(np.arange(256, dtype=np.uint8), (16, 1))           # [16,256]: all grey levels
    frames = [ramp, ramp, ramp[:, ::-1].copy(), ramp[:, ::-1].copy(), np.zeros_like(ramp), np.full_like(ramp, 255), ramp]
    times = [0.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06]
    ora = oracle_lib.OracleEmulator(rng_mode="philox", seed=8, **kw)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    hip = EventEmulator(device="cuda", rng_mode="philox", seed=8, **kw)
    if api == "frame":
        for k, (f, t) in enumerate(zip(frames, times)):
            x = torch.from_numpy(f) if k % 2 == 0 else torch.from_numpy(f).cuda()  # CPU and CUDA tensors
            ev = hip.generate_events(x, t)
            assert events_equal(ev, oev[k]), "frame %d" % k
    else:
        ev, counts = hip.generate_events_batch(np.stack(frames), times)
        assert list(counts) == [0 if e is None else len(e) for e in oev]
        assert np.array_equal(ev, np.concatenate([e for e in oev if e is not None]))
    assert oev[1] is None or len(oev[1]) < 50                           # static scene: no signal events
    assert np.array_equal(hip.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert hip.t_previous == times[-1] and hip.num_events_total == ora.num_events_total
    # reset(): the next frame re-initialises the state like a fresh emulator
    hip.reset()
    assert hip.num_events_total == 0 and hip.generate_events(ramp, 1.0) is None


def test_frame_api_result_arrays_are_the_callers(oracle_lib):
    """The frame API hands its rows out in pinned buffers that return to a pool when the caller drops the array (no host
    copy).  What the caller keeps must stay what it was -- arrays kept across later frames, views derived from an array
    that itself was dropped -- and what it drops must be reused, not leaked."""
    import gc
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    F, H, W = 90, 64, 96
    frames = int_gradient_frames(F, H, W, seed=61, noise=8, as_array=True)
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=200, leak_rate_hz=.2, shot_noise_rate_hz=2.0)
    emu = EventEmulator(device="cuda", seed=6, rng_mode="philox", **kw)
    ora = oracle_lib.OracleEmulator(seed=6, rng_mode="philox", **kw)
    kept, kept_cols, ref = [], [], []
    for i in range(F):
        ev = emu.generate_events(frames[i], i / 300)
        oe = ora.generate_events(frames[i], i / 300)
        assert events_equal(ev, oe)
        if ev is None:
            continue
        assert ev.dtype == np.float32 and ev.flags.c_contiguous and ev.flags.writeable
        if i < 75:  # more arrays than the pool has buffers (64): the rest are ordinary copies
            kept.append(ev)
            ref.append(oe)
        elif i < 80:
            kept_cols.append((ev[:, 0], oe[:, 0].copy()))  # only a view survives this iteration
        del ev
    gc.collect()
    for a, b in zip(kept, ref):
        assert np.array_equal(a, b)
    for a, b in kept_cols:
        assert np.array_equal(a, b)
    pool = emu._engine._rows_pool
    assert len(pool.bufs) <= pool.MAX_BUFS
    busy = sum(not b.free for b in pool.bufs)
    assert busy >= min(len(kept), pool.MAX_BUFS) - 1
    # dropped arrays give their buffers back
    del kept, kept_cols, a
    gc.collect()
    assert sum(not b.free for b in pool.bufs) == 0
    n_bufs = len(pool.bufs)
    for i in range(F, F + 20):
        ev = emu.generate_events(frames[i % F], i / 300)
    assert len(pool.bufs) == n_bufs  # steady state: one buffer in use, none added


@pytest.mark.parametrize("shape", [(33, 37), (64, 96)])
def test_scidvs_matches_oracle_on_seeded_inputs(shape, oracle_lib):
    """SCIDVS (float64 state) against the oracle's restatement on ragged sizes: events bit for bit, the high-pass plane to
    1e-12 (the device's and libm's float64 sinh differ in the last bit on a few percent of the arguments)."""
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    H, W = shape
    frames = int_gradient_frames(10, H, W, seed=81, noise=8, as_array=True)
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=200, leak_rate_hz=.3, shot_noise_rate_hz=3.0,
              refractory_period_s=0.0005, scidvs=True)
    emu = EventEmulator(device="cuda", seed=12, rng_mode="philox", **kw)
    ora = oracle_lib.OracleEmulator(seed=12, rng_mode="philox", **kw)
    for i in range(10):
        assert events_equal(emu.generate_events(frames[i], i / 300), ora.generate_events(frames[i], i / 300)), "frame %d" % i
    hp = emu.scidvs_highpass.cpu().numpy()
    assert np.max(np.abs(hp - ora.scidvs_highpass)) <= 1e-12 * max(1.0, np.max(np.abs(ora.scidvs_highpass)))
    assert np.array_equal(emu.scidvs_tau_arr.cpu().numpy(), ora.scidvs_tau_arr)


@pytest.mark.parametrize("name", ["tape_defaults_40x48", "tape_refractory_float_33x37"])
def test_single_pixel_recorder_matches_reference(name, oracle_lib, tmp_path, monkeypatch):
    """record_single_pixel_states (emulator.py:279-300, 985-1009): the ten series the reference records for one pixel, from
    the reference itself (tests/golden/make_golden_single_pixel.py: same run as the tape fixture), bit for bit as float64 --
    log_new_frame and diff_frame come from the planes k_count leaves for this option, the ON / OFF counts from the frame's
    signal events.  The events of the run are unchanged by the option, and cleanup() pickles the dict as the reference does."""
    import os
    import pickle
    from fixtures import GOLDEN
    g = np.load(os.path.join(GOLDEN, "single_pixel.npz"))
    ij = tuple(int(v) for v in g[name + "__pixel"])
    fx = TapeFixture(name)
    monkeypatch.chdir(tmp_path)  # SINGLE_PIXEL_STATES_FILENAME is relative, as in the reference
    emu = _mk(fx, seed=0, rng_mode="tape", tape=oracle_lib.RecordedTape(fx.items), record_single_pixel_states=ij)
    for k, (f, t) in enumerate(zip(fx.frames, fx.times)):
        assert events_equal(emu.generate_events(f, float(t)), fx.events[k]), "frame %d differs from the reference" % k
    n = emu.single_pixel_sample_count
    assert n == len(g[name + "__time"]) == len(fx.frames) - 1
    for key, arr in emu.single_pixel_states.items():
        ref = g["%s__%s" % (name, key)]
        assert np.array_equal(arr[:n], ref), (key, arr[:n], ref)
        assert np.isnan(arr[n:]).all()
    assert float(np.sum(g[name + "__final_neg_evts_frame"]) + np.sum(g[name + "__final_pos_evts_frame"])) > 0
    emu.cleanup()
    with open(str(tmp_path / emu.SINGLE_PIXEL_STATES_FILENAME), "rb") as f:
        d = pickle.load(f)
    assert sorted(d) == sorted(emu.single_pixel_states) and np.array_equal(d["diff_frame"][:n], g[name + "__diff_frame"])
    with pytest.raises(ValueError):
        _mk(fx, record_single_pixel_states=[1, 2])  # emulator.py:284-285: a tuple


def test_model_state_planes_are_readable_without_a_display():
    """show_dvs_model_state (emulator.py:756-767): without OpenCV / a display nothing is shown, one warning, and the run is the
    same run; the named states are readable as attributes (diff_frame == lp_log_frame - base_log_frame before the update)."""
    fx = PhiloxFixture("philox_moving_dot_64x64")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox", show_dvs_model_state=["all"])
    for k, (f, t) in enumerate(zip(fx.frames[:30], fx.times[:30])):
        ev = emu.generate_events(f, float(t))
        n = 0 if ev is None else len(ev)
        assert n == fx.n_events[k] and (n == 0 or sha(ev) == fx.ev_sha[k])
    assert emu.diff_frame is not None and tuple(emu.diff_frame.shape) == tuple(fx.frames[0].shape)
    assert emu.log_new_frame.dtype == torch.float32 and emu.c_minus_s_frame is None
    emu.cleanup()


@pytest.mark.parametrize("chunk", [24, 9, 5])
@pytest.mark.parametrize("name", ["philox_refractory_346x260", "philox_noisy_346x260", "philox_defaults_346x260"])
def test_pipelined_runs_equal_whole_clip(name, chunk):
    """Pipelined runs (v2e_emu_run 0 | 1024: plain launches on four streams, the next run's upload and records beside this run's chain,
    this run's last emission batches beside the next run's chain, two scratch sets alternating): a clip fed in runs of `chunk` frames,
    every run enqueued before the result of the run before it is read, with and without the caller's word that the frames are
    resident -- the reference's events, frame by frame, and its final state."""
    fx = PhiloxFixture(name)
    for resident in (True, False):
        emu = _mk(fx, seed=fx.seed, rng_mode="philox")
        frames = torch.from_numpy(np.ascontiguousarray(fx.frames)).cuda()
        torch.cuda.synchronize()
        pend, evs, cnts = [], [], []
        for lo in range(0, len(fx.frames), chunk):
            pend.append(emu.generate_events_batch_async(frames[lo:lo + chunk], fx.times[lo:lo + chunk], use_graph=0, return_device=True,
                                                        frames_resident=resident))
            if len(pend) > 1:
                ev, c = pend.pop(0).result()
                evs.append(ev.cpu().numpy()); cnts += list(c)
        ev, c = pend.pop(0).result()
        evs.append(ev.cpu().numpy()); cnts += list(c)
        assert cnts == list(fx.n_events)
        ev = np.concatenate(evs)
        row = 0
        for k, n in enumerate(cnts):
            if n:
                assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
            row += n
        st = _state(emu)
        assert sha(st["base_log_frame"]) == fx.base_sha
        assert sha(st["lp_log_frame"]) == fx.lp_sha
        if fx.ts_mem_sha:
            assert sha(st["timestamp_mem"]) == fx.ts_mem_sha
        # ... and the frame-at-a-time API right behind a pipelined run sees its state (every other entry point joins by itself)
        emu2 = _mk(fx, seed=fx.seed, rng_mode="philox")
        half = len(fx.frames) // 2
        p = emu2.generate_events_batch_async(frames[:half], fx.times[:half], use_graph=0, frames_resident=resident)
        e1 = emu2.generate_events(fx.frames[half], float(fx.times[half]))
        assert (0 if e1 is None else len(e1)) == fx.n_events[half]
        if e1 is not None:
            assert sha(e1) == fx.ev_sha[half]
        assert list(p.result()[1]) == list(fx.n_events[:half])


def test_pipelined_runs_at_1280x720():
    """The same at 1280x720 with a refractory period (records built in the chain, K = 8, the two-level pull writer): runs of 9 frames."""
    fx = PhiloxFixture("philox_refractory_1280x720")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    emu.generate_events(fx.frames[0], float(fx.times[0]))
    frames = torch.from_numpy(np.ascontiguousarray(fx.frames)).cuda()
    torch.cuda.synchronize()
    pend, evs, cnts = [], [], [0]
    for lo in range(1, len(fx.frames), 9):
        pend.append(emu.generate_events_batch_async(frames[lo:lo + 9], fx.times[lo:lo + 9], use_graph=0, return_device=True, cap=8_000_000,
                                                    frames_resident=True))
        if len(pend) > 1:
            ev, c = pend.pop(0).result()
            evs.append(ev.cpu().numpy()); cnts += list(c)
    ev, c = pend.pop(0).result()
    evs.append(ev.cpu().numpy()); cnts += list(c)
    assert cnts == list(fx.n_events)
    ev = np.concatenate(evs)
    row = 0
    for k, n in enumerate(cnts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    assert sha(_state(emu)["base_log_frame"]) == fx.base_sha
