"""GPU parity tests of the code paths bench.py actually times (round-2 VERDICT, parity holes 1 and 2):

* 1280x720 clips long enough for full 32-frame chain launches, a validating successor and a wrap of the ring of frame
  slots (reference-generated digests, tests/golden/make_golden_hd_long.py), and 1280x720 with a refractory period;
* the chain with its records built inside the kernel (V2E_AMD_CHAIN_FUSED=1) on every 346x260 fixture -- with the
  refractory fixture that is the redo / rendezvous path of the fused instantiation -- and the 720p clip through k_ahead
  records (=0); a mid-size sensor (640x480) where the fused chain with a refractory period is what runs by default;
* benchutil.run_steps itself -- generate_events_batch_async with step n + 1 enqueued while step n executes, two event /
  record buffer sets and two pinned staging sets alternating, cached hipGraphs replayed across steps, the
  `_refr_mostly_on` switch -- and the hd_noisy enqueue loop, step by step against the CPU oracle.

Everything is bit-exact (event rows incl. order, state planes, counters)."""
import os
import numpy as np
import pytest
import torch

from fixtures import PHILOX_HD_FIXTURES, PhiloxFixture, sha

pytestmark = pytest.mark.gpu


def _mk(fx, **extra):
    from v2e_amd import EventEmulator
    emu = EventEmulator(device="cuda", **fx.kw, **extra)
    if fx.preset:
        emu.set_dvs_params(fx.preset)
    return emu


def _check_clip(fx, emu, ev, counts):
    assert list(counts) == list(fx.n_events)
    row = 0
    for k, n in enumerate(counts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    assert sha(emu.base_log_frame.cpu().numpy()) == fx.base_sha
    assert sha(emu.lp_log_frame.cpu().numpy()) == fx.lp_sha
    if fx.ts_mem_sha:
        assert sha(emu.timestamp_mem.cpu().numpy()) == fx.ts_mem_sha
    assert [emu.num_events_total, emu.num_events_on, emu.num_events_off] == list(fx.counters)


@pytest.mark.parametrize("chain_k", [None, 32])
@pytest.mark.parametrize("use_graph", [0, 1])
@pytest.mark.parametrize("name", PHILOX_HD_FIXTURES)
def test_hd_long_clips_match_reference(name, use_graph, chain_k, monkeypatch):
    """1280x720 through the default pipeline selection (and with 32 frames per launch), 40 MB of events per run handed back as one array."""
    if chain_k is not None:
        monkeypatch.setenv("V2E_AMD_CHAIN_K", str(chain_k))
    fx = PhiloxFixture(name)
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=use_graph, cap=12_000_000)
    _check_clip(fx, emu, ev, counts)
    kind, fpl, fpb = emu._engine.last_pipeline()
    assert kind.startswith("k_chain"), kind
    if "noisy" in name:  # no refractory period: the longest launch by default (a full launch and a partial one over 104 frames);
        assert fpl == (chain_k or 64)  # with 32: three full launches and a partial one


def test_hd_long_clip_in_two_async_runs():
    """The 104-frame 720p clip as the hd_noisy bench feeds it: runs enqueued back to back through the async API."""
    fx = PhiloxFixture("philox_noisy_1280x720_long")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    emu.generate_events(fx.frames[0], float(fx.times[0]))
    cuts = [1, 41, 105 - 1]
    pend = [emu.generate_events_batch_async(fx.frames[a:b], fx.times[a:b], cap=8_000_000) for a, b in zip(cuts[:-1], cuts[1:])]
    res = [p.result() for p in pend]
    ev = np.concatenate([r[0] for r in res])
    counts = [0] + [int(c) for r in res for c in r[1]]
    _check_clip(fx, emu, ev, counts)


@pytest.mark.parametrize("chain_k", [8, 32])
@pytest.mark.parametrize("name", ["philox_refractory_346x260", "philox_defaults_346x260", "philox_noisy_346x260"])
def test_chain_records_built_in_kernel_on_small_grids(name, chain_k, monkeypatch):
    """V2E_AMD_CHAIN_FUSED=1: the instantiation the 1280x720 and multi-clip workloads run, here with the refractory
    fixtures (redo passes, checkpoints, rendezvous) that a 720p grid cannot take."""
    monkeypatch.setenv("V2E_AMD_CHAIN_FUSED", "1")
    monkeypatch.setenv("V2E_AMD_CHAIN_K", str(chain_k))
    fx = PhiloxFixture(name)
    for use_graph in (256, 257):
        emu = _mk(fx, seed=fx.seed, rng_mode="philox")
        ev, counts = emu.generate_events_batch(fx.frames, fx.times, use_graph=use_graph)
        _check_clip(fx, emu, ev, counts)
        assert emu._engine.last_pipeline()[0] == "k_chain(fused records)"


def test_hd_clip_through_ahead_records(monkeypatch):
    """V2E_AMD_CHAIN_FUSED=0 at 1280x720: k_ahead's records through LDS on a grid of 3 600 workgroups."""
    monkeypatch.setenv("V2E_AMD_CHAIN_FUSED", "0")
    fx = PhiloxFixture("philox_noisy_1280x720_long")
    emu = _mk(fx, seed=fx.seed, rng_mode="philox")
    ev, counts = emu.generate_events_batch(fx.frames[:70], fx.times[:70], use_graph=257, cap=8_000_000)
    assert list(counts) == list(fx.n_events[:70])
    row = 0
    for k, n in enumerate(counts):
        if n:
            assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs" % k
        row += n
    assert emu._engine.last_pipeline()[0] == "k_chain"


@pytest.mark.parametrize("refr", [0.002, 0.0004])
def test_mid_size_sensor_fused_chain_with_refractory(refr, oracle_lib):
    """640x480 (1 200 workgroups: a large grid whose workgroups are still co-resident): records built in the chain AND
    redo passes, several launches, against the oracle event for event."""
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    F, H, W = 41, 480, 640
    frames = int_gradient_frames(F, H, W, seed=31, noise=6, as_array=True)
    times = [i / 400 for i in range(F)]
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=200, leak_rate_hz=.2, shot_noise_rate_hz=1.0,
              refractory_period_s=refr)
    emu = EventEmulator(device="cuda", seed=3, rng_mode="philox", **kw)
    ev, counts = emu.generate_events_batch(frames, times, use_graph=257)
    assert emu._engine.last_pipeline()[0] == "k_chain(fused records)"
    ora = oracle_lib.OracleEmulator(seed=3, rng_mode="philox", **kw)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    assert list(counts) == [0 if e is None else len(e) for e in oev]
    assert np.array_equal(ev, np.concatenate([e for e in oev if e is not None]))
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.lp_log_frame.cpu().numpy(), ora.lp_log_frame)
    assert np.array_equal(emu.timestamp_mem.cpu().numpy(), ora.timestamp_mem)


class _DigestSink:
    """Stands in for EventStreamGatherer in run_steps: keeps a digest and the row count of every step's stream."""

    def __init__(self):
        self.steps = []

    def submit(self, ev, n, ready_event=None, run_bound=None):
        self.steps.append((int(n), sha(ev[:n].cpu().numpy())))

    def wait(self):
        pass


@pytest.mark.parametrize("refr", [0.0005, 0.004])
def test_bench_step_loop_matches_oracle(refr, oracle_lib):
    """benchutil.run_steps exactly as bench.py calls it (1 warm-up + 4 timed steps of 300 frames at 346x260, the
    synthetic clip cycled, time running on): every step's event stream, and the final planes, against the oracle.
    refr = 4 ms makes the rule active on most frames, so `_refr_mostly_on` switches the pipeline between steps."""
    import bench as B
    from v2e_amd import EventEmulator
    from v2e_amd.benchutil import run_steps
    dev = torch.device("cuda")
    F, steps, warm = B.FRAMES_PER_STEP, 4, 1
    kw = dict(B.DEFAULT_KW)
    kw["refractory_period_s"] = refr
    frames_all = B.gen_frames_device(2 * F + 1, 1, dev)  # two seconds of video, cycled
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **kw)
    emu.generate_events(frames_all[0], 0.0)
    sink = _DigestSink()
    elapsed, n_events = run_steps(emu, frames_all, F, B.DT, steps, warm, sink, None, dev)
    assert len(sink.steps) == steps + warm and n_events == sum(n for n, _ in sink.steps[warm:])
    host = frames_all.cpu().numpy()
    ora = oracle_lib.OracleEmulator(seed=1, rng_mode="philox", **kw)
    ora.generate_events(host[0], 0.0)
    for s in range(steps + warm):
        lo = 1 + (s % 2) * F
        evs = [ora.generate_events(host[lo + i], (1 + s * F + i) * B.DT) for i in range(F)]
        ref = np.concatenate([e for e in evs if e is not None])
        assert sink.steps[s][0] == len(ref), "step %d: %d events, oracle %d" % (s, sink.steps[s][0], len(ref))
        assert sink.steps[s][1] == sha(ref), "step %d event stream differs from the oracle" % s
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.lp_log_frame.cpu().numpy(), ora.lp_log_frame)
    assert np.array_equal(emu.timestamp_mem.cpu().numpy(), ora.timestamp_mem)
    assert emu.num_events_total == ora.num_events_total and emu.num_events_on == ora.num_events_on


def test_pipelined_runs_longer_than_the_ring_match_oracle(oracle_lib):
    """Pipelined runs of a small single-clip grid take a ring of five 64-frame batches: a 300-frame run fits (one k_ahead launch, one
    wait), a 400-frame run does NOT -- k_ahead goes batch by batch again and waits for the chain where the ring wraps (batches 5, 6 reuse
    the slots of 0, 1).  Three runs of 400 frames through bench.py's loop, every run's event stream and the final planes against the
    oracle."""
    import bench as B
    from v2e_amd import EventEmulator
    from v2e_amd.benchutil import run_steps
    dev = torch.device("cuda")
    F, steps, warm = 400, 2, 1
    frames_all = B.gen_frames_device(2 * F + 1, 1, dev)
    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames_all[0], 0.0)
    sink = _DigestSink()
    run_steps(emu, frames_all, F, B.DT, steps, warm, sink, None, dev)
    assert len(sink.steps) == steps + warm
    host = frames_all.cpu().numpy()
    ora = oracle_lib.OracleEmulator(seed=1, rng_mode="philox", **B.DEFAULT_KW)
    ora.generate_events(host[0], 0.0)
    for s in range(steps + warm):
        lo = 1 + (s % 2) * F
        evs = [ora.generate_events(host[lo + i], (1 + s * F + i) * B.DT) for i in range(F)]
        ref = np.concatenate([e for e in evs if e is not None])
        assert sink.steps[s][0] == len(ref), "run %d: %d events, oracle %d" % (s, sink.steps[s][0], len(ref))
        assert sink.steps[s][1] == sha(ref), "run %d event stream differs from the oracle" % s
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.timestamp_mem.cpu().numpy(), ora.timestamp_mem)


@pytest.mark.parametrize("use_graph", [0, 1])
def test_hd_noisy_enqueue_loop_matches_oracle(use_graph, oracle_lib):
    """The loop of benchutil.hd_noisy_emulator_bench (BASELINE configs[3]: 1280x720, noisy preset, dt = 1/600 s): runs of
    64 frames over one fixed frame buffer, run n + 1 enqueued before run n's result is read -- three runs, each run's
    stream against the oracle."""
    import bench as B
    from v2e_amd import EventEmulator
    dev = torch.device("cuda")
    frames, H, W = 64, 720, 1280
    fr = B.gen_frames_device(frames + 1, 4, dev, h=H, w=W)
    emu = EventEmulator(device=dev, seed=4, rng_mode="philox", **B.DEFAULT_KW)
    emu.set_dvs_params("noisy")
    dt = 1.0 / 600.0
    emu.generate_events(fr[0], 0.0)
    buf = fr[1:].contiguous()
    cap = 400_000 * frames

    def enqueue(k):
        # (use_graph 0: pipelined runs, what the bench leg runs since round 6; 1: one hipGraph per run)
        return emu.generate_events_batch_async(buf, [(1 + k * frames + i) * dt for i in range(frames)], return_device=True, cap=cap,
                                               use_graph=use_graph, frames_resident=True)

    got, pend = [], None
    for k in range(3):
        nxt = enqueue(k)
        if pend is not None:
            ev, c = pend.result()
            got.append((int(c.sum()), sha(ev.cpu().numpy())))
        pend = nxt
    ev, c = pend.result()
    got.append((int(c.sum()), sha(ev.cpu().numpy())))
    assert emu._engine.last_pipeline()[:2] == ("k_chain(fused records)", 64)  # no refractory period: the longest launch
    host = fr.cpu().numpy()
    ora = oracle_lib.OracleEmulator(seed=4, rng_mode="philox", **B.DEFAULT_KW)
    ora.set_dvs_params("noisy")
    ora.generate_events(host[0], 0.0)
    for k in range(3):
        evs = [ora.generate_events(host[1 + i], (1 + k * frames + i) * dt) for i in range(frames)]
        ref = np.concatenate([e for e in evs if e is not None])
        assert got[k] == (len(ref), sha(ref)), "run %d differs from the oracle" % k
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.lp_log_frame.cpu().numpy(), ora.lp_log_frame)


@pytest.mark.parametrize("nframes", [9, 33, 70])
@pytest.mark.parametrize("shape", [(64, 96), (260, 346)])
def test_chain_carries_lp_without_cutoff_or_shot(shape, nframes, oracle_lib):
    """Refractory period, no photoreceptor cutoff, no shot noise (round-2 advisor finding): lp_log_frame is not part of the
    chain's arithmetic there, but the state planes ping-pong between launches and the tail launch hands them back, so it
    must be carried through whatever the number of launches (1, 2 and 3 + tail here)."""
    from v2e_amd import EventEmulator
    from v2e_amd.synth import int_gradient_frames
    H, W = shape
    frames = int_gradient_frames(nframes, H, W, seed=51, noise=6, as_array=True)
    times = [i / 300 for i in range(nframes)]
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=0, leak_rate_hz=.2, shot_noise_rate_hz=0.0,
              refractory_period_s=0.002)
    emu = EventEmulator(device="cuda", seed=5, rng_mode="philox", **kw)
    ev, counts = emu.generate_events_batch(frames, times, use_graph=257)
    assert emu._engine.last_pipeline()[0].startswith("k_chain")
    ora = oracle_lib.OracleEmulator(seed=5, rng_mode="philox", **kw)
    oev = [ora.generate_events(f, t) for f, t in zip(frames, times)]
    assert list(counts) == [0 if e is None else len(e) for e in oev]
    assert np.array_equal(ev, np.concatenate([e for e in oev if e is not None]))
    assert emu.lp_log_frame.dtype == torch.float32
    assert np.array_equal(emu.lp_log_frame.cpu().numpy(), ora.lp_log_frame)
    assert np.array_equal(emu.base_log_frame.cpu().numpy(), ora.base_log_frame)
    assert np.array_equal(emu.timestamp_mem.cpu().numpy(), ora.timestamp_mem)


def test_pipelined_loop_on_recycled_device_memory_equals_graph_runs():
    """Round 6 regression: the first pipelined run on a freshly allocated scratch set, with another run still in flight, read a
    record ring that an allocation-time fill (ordered on the default stream only) wiped AFTER k_ahead had written it -- only on
    recycled (non-zero) device memory and only where the pipeline re-allocates mid-loop (refr = 4 ms: `_refr_mostly_on` switches to
    one frame per launch after the first step).  The loop of bench.py on three emulators in a row, with other allocations on the
    device, against one hipGraph per run on one stream."""
    import bench as B
    from v2e_amd import EventEmulator
    from v2e_amd.benchutil import run_steps
    dev = torch.device("cuda")
    kw = dict(B.DEFAULT_KW)
    kw["refractory_period_s"] = 0.004
    F = B.FRAMES_PER_STEP
    frames = B.gen_frames_device(2 * F + 1, 1, dev)
    junk = [torch.full((1 << 27,), 0xA5, dtype=torch.uint8, device=dev) for _ in range(4)]  # noqa: F841 (recycled memory is not zero)
    del junk
    ref = None
    for rep in range(4):
        if rep == 0:
            os.environ["V2E_AMD_BENCH_UG"] = "1"
        else:
            os.environ.pop("V2E_AMD_BENCH_UG", None)
        try:
            emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **kw)
            emu.generate_events(frames[0], 0.0)
            sink = _DigestSink()
            run_steps(emu, frames, F, B.DT, 4, 1, sink, None, dev)
        finally:
            os.environ.pop("V2E_AMD_BENCH_UG", None)
        if ref is None:
            ref = sink.steps
        else:
            assert sink.steps == ref, "pipelined repetition %d: %s, graph runs %s" % (rep, [n for n, _ in sink.steps], [n for n, _ in ref])


@pytest.mark.parametrize("use_graph,return_device", [(0, True), (0, False), (1, True)])
def test_three_runs_enqueued_before_any_result_is_read(use_graph, return_device):
    """Two buffer sets (and, for pipelined runs, two scratch sets with their pinned records) alternate between asynchronous runs:
    the third enqueue reuses the first run's.  A handle still unread at that point is collected by the enqueue itself -- reading
    it afterwards must give ITS events, not the third run's."""
    import bench as B
    from v2e_amd import EventEmulator
    dev = torch.device("cuda")
    F = 40
    frames = B.gen_frames_device(4 * F + 1, 1, dev)
    ts = lambda s: [(1 + s * F + i) * B.DT for i in range(F)]  # noqa: E731

    ref = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    ref.generate_events(frames[0], 0.0)
    want = []
    for s in range(4):
        ev, counts = ref.generate_events_batch(frames[1 + s * F:1 + (s + 1) * F], ts(s), return_device=True)
        want.append((ev.cpu().numpy().copy(), counts.copy()))

    emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
    emu.generate_events(frames[0], 0.0)
    pend = [emu.generate_events_batch_async(frames[1 + s * F:1 + (s + 1) * F], ts(s), return_device=return_device, use_graph=use_graph)
            for s in range(4)]
    for s in (3, 0, 2, 1):  # any order
        ev, counts = pend[s].result()
        got = ev.cpu().numpy() if return_device else ev
        assert np.array_equal(counts, want[s][1]), s
        assert got.shape == want[s][0].shape and np.array_equal(got, want[s][0]), s
    assert emu.num_events_total == ref.num_events_total


def test_one_handle_mixing_pipelined_and_graph_runs():
    """A handle keeps the ring depth its first device-resident run chose (five batches when that run is pipelined on a small single-clip
    grid): later graph runs and blocking calls on the same handle run on that ring.  Pipelined, graph, blocking, pipelined again -- runs of
    300, 100, 37 and 300 frames -- against one-graph-per-run on a fresh emulator."""
    import bench as B
    from v2e_amd import EventEmulator
    dev = torch.device("cuda")
    lens = [300, 100, 37, 300]
    frames = B.gen_frames_device(sum(lens) + 1, 1, dev)
    times = [(1 + i) * B.DT for i in range(sum(lens))]

    def feed(modes):
        emu = EventEmulator(device=dev, seed=1, rng_mode="philox", **B.DEFAULT_KW)
        emu.generate_events(frames[0], 0.0)
        out, lo = [], 0
        for n, mode in zip(lens, modes):
            fr, ts = frames[1 + lo:1 + lo + n], times[lo:lo + n]
            if mode == "blocking":
                ev, counts = emu.generate_events_batch(fr, ts, return_device=True)
            else:
                ev, counts = emu.generate_events_batch_async(fr, ts, return_device=True, use_graph=0 if mode == "pipelined" else 1).result()
            out.append((sha(ev.cpu().numpy()), list(counts)))
            lo += n
        return out, sha(emu.base_log_frame.cpu().numpy()), sha(emu.timestamp_mem.cpu().numpy())

    want = feed(["graph"] * 4)
    got = feed(["pipelined", "graph", "blocking", "pipelined"])
    assert got == want
    assert feed(["graph", "pipelined", "pipelined", "blocking"]) == want
