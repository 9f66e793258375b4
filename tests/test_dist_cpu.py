"""CPU, gloo, world_size 2: the N>1 path (clip sharding + all-gather(v) of the event streams)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


from tests.bench_stub import StubEmulator as _StubEmulator, rows as _rows  # noqa: E402


def _rows_runs(n, rank, step):
    """n event rows in blocks of one time stamp (as an emulator emits them: all events of one (frame, iteration) share t),
    block lengths 1..7, x up to 1279, y up to 719, a negative and a zero time stamp among them."""
    k = torch.arange(n, dtype=torch.int64)
    blk = torch.div(k * 3 + rank, 7 + step, rounding_mode="floor")
    t = (blk.to(torch.float32) - 2.0) * 1.25e-4 * (1 + rank)
    return torch.stack([t, ((k * 37 + 13 * rank) % 1280).to(torch.float32), ((k * 7 + step) % 720).to(torch.float32),
                        ((k % 3 == 0).to(torch.float32)) * 2 - 1], dim=1).contiguous()


def _counts(world, rank, step):
    n = 5 + 7 * rank + 3 * step           # ragged, different per rank and step
    if step == 2 and rank == 1:
        n = 0                              # empty stream on one rank
    if step == 1 and rank == world - 1:
        n = 0
    return n


def _worker(rank, world, port, q, wire="pack64", algo="allgather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2e_amd.dist import EventStreamGatherer, clips_of_rank
    g = EventStreamGatherer("cpu", world, wire=wire, algo=algo, sensor=(720, 1280))
    ok = True
    rows = _rows if wire == "pack64" else _rows_runs
    for step in range(3):
        n = _counts(world, rank, step)
        ev = rows(n + 4, rank, step)
        g.submit(ev, n, run_bound=(n if wire == "pack32" else None))
        parts = g.result()
        for r in range(world):
            nr = _counts(world, r, step)
            exp = rows(nr + 4, r, step)[:nr]
            ok &= parts[r].shape == exp.shape and torch.equal(parts[r].view(torch.int32), exp.view(torch.int32))
    ok &= clips_of_rank(8, world, rank) == list(range(rank, 8, world))
    ok &= g.bytes_gathered > 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run_gloo(world, wire, algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, wire, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == {r: True for r in range(world)}


def test_event_stream_allgather_gloo_world2():
    _run_gloo(2, "pack64", "allgather")


@pytest.mark.parametrize("wire,algo", [("pack64", "allgather"), ("pack32", "allgather"), ("pack32", "p2p"), ("pack64", "p2p")])
def test_event_stream_exchange_gloo_world4_unequal_counts(wire, algo):
    """Four ranks, ragged streams incl. empty ones, both wire formats (8 bytes per event; 4 bytes per event + one time stamp
    per block), both exchange algorithms (padded all-gather; grouped point-to-point sends of exact sizes)."""
    _run_gloo(4, wire, algo)


def _worker_flags(rank, world, port, q):
    """Round-3 advisor finding: a pack32 overflow on ONE rank must be seen by EVERY rank (it used to raise on the sender only,
    after the exchange, and the peers unpacked against a truncated run table); ranks that disagree about the run bound must
    not issue different collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2e_amd.dist import EventStreamGatherer
    ok = True
    # (a) rank 1 under-states its run bound: both ranks raise, with rank 1 named
    g = EventStreamGatherer("cpu", world, wire="pack32", sensor=(720, 1280))
    n = 5 if rank == 0 else 40   # the run tables are sized by the largest bound of the step: 5 here, rank 1 has more blocks
    ev = _rows_runs(n, rank, 0)
    g.submit(ev, n, run_bound=(5 if rank == 0 else 3))
    try:
        g.result()
        ok = False
    except ValueError as e:
        ok &= "rank 1" in str(e) and "run_bound too small" in str(e) and "rank 0" not in str(e)
    # (b) a coordinate that does not fit on rank 0 (explicit pack32 without a sensor): both ranks raise
    g = EventStreamGatherer("cpu", world, wire="pack32")
    ev = _rows_runs(40, rank, 0)
    if rank == 0:
        ev[5, 1] = 2050.0
    g.submit(ev, 40, run_bound=40)
    try:
        g.result()
        ok = False
    except ValueError as e:
        ok &= "rank 0" in str(e) and "coordinate" in str(e)
    # (c) only one rank supplies a bound: every rank sends the 8-byte format for that step, and the data arrive
    g = EventStreamGatherer("cpu", world, wire="auto", sensor=(720, 1280))
    ev = _rows_runs(30 + rank, rank, 1)
    g.submit(ev, 30 + rank, run_bound=(30 if rank == 0 else None))
    ok &= g.last["wire"] == "pack64"
    parts = g.result()
    for r in range(world):
        exp = _rows_runs(30 + r, r, 1)
        ok &= torch.equal(parts[r].view(torch.int32), exp.view(torch.int32))
    # (d) no sensor given: "auto" never assumes the frame fits 2048 x 1024
    ok &= EventStreamGatherer("cpu", world).wire == "pack64"
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_pack32_overflow_and_wire_disagreement_are_seen_by_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flags, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_wire_formats_round_trip_and_agree():
    """pack32 (payload + run table) and pack64 restore the same float32 rows bit for bit: blocks of one time stamp, a
    negative time stamp, -0.0 next to +0.0 (different bits: two blocks), coordinates up to 2047 x 1023; a coordinate beyond
    that is flagged."""
    from v2e_amd.dist import pack_events32, pack_events64, unpack_events32, unpack_events64
    ev = _rows_runs(5000, 1, 2)
    ev[10:20, 0] = -0.0
    ev[20:30, 0] = 0.0
    ev[100, 1], ev[100, 2] = 2047.0, 1023.0
    pl, runs, fl = pack_events32(ev)
    assert int(fl[0]) == 0 and int(runs[0]) == runs.numel() - 1
    back32 = unpack_events32(pl, runs)
    back64 = unpack_events64(pack_events64(ev))
    assert torch.equal(back32.view(torch.int32), ev.view(torch.int32))
    assert torch.equal(back64.view(torch.int32), ev.view(torch.int32))
    # an emulator's blocks hold thousands of events: the run table is noise, the stream half of pack64's
    big = _rows_runs(60000, 0, 0)
    big[:, 0] = torch.div(torch.arange(60000), 500, rounding_mode="floor").to(torch.float32) * 1e-4
    pl2, runs2, _ = pack_events32(big)
    assert 4 * pl2.numel() + 8 * runs2.numel() < 0.51 * 8 * big.shape[0]
    assert torch.equal(unpack_events32(pl2, runs2).view(torch.int32), big.view(torch.int32))
    ev[7, 1] = 2048.0
    assert int(pack_events32(ev)[2][0]) != 0


def test_clip_sharding_covers_all_clips():
    from v2e_amd.dist import clips_of_rank
    for world in (1, 2, 4, 8):
        allc = sorted(c for r in range(world) for c in clips_of_rank(8, world, r))
        assert allc == list(range(8))


def _nccl_worker(rank, world, port, q, wire="pack64", algo="allgather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from v2e_amd.dist import EventStreamGatherer
    g = EventStreamGatherer(dev, world, wire=wire, algo=algo, sensor=(260, 346))
    ok = True
    for step in range(4):
        n = 1000 + 50000 * rank + 7 * step     # unequal per rank: the padded size must come from the gathered counts
        if step == 3 and rank == 0:
            n = 0
        ev = _rows(n + 4, rank, step).to(dev)
        # keep the main stream busy so that a count read on the wrong stream would race the collective
        _ = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
        g.submit(ev, n, run_bound=(n if wire == "pack32" else None))
        parts = g.result()
        for r in range(world):
            nr = 1000 + 50000 * r + 7 * step
            if step == 3 and r == 0:
                nr = 0
            exp = _rows(nr + 4, r, step)[:nr]
            ok &= parts[r].shape == exp.shape and torch.equal(parts[r].cpu(), exp)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("wire,algo", [("pack64", "allgather"), ("pack32", "allgather"), ("pack32", "p2p")])
def test_event_stream_allgather_nccl_world2_unequal_counts(wire, algo):
    """RCCL path with two ranks and rank-dependent row counts (needs two GPUs; the 1-GPU test box skips it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q, wire, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def _bench_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from v2e_amd.benchutil import run_steps
    from v2e_amd.dist import EventStreamGatherer
    F, K, Wm, dt = 4, 5, 2, 0.25
    clip_seed = 10 + rank                               # bench.py: one independent clip per rank, seeds 10..17
    frames_all = torch.full((3 * F + 1, 2, 3), 0, dtype=torch.uint8)
    for i in range(frames_all.shape[0]):
        frames_all[i] = (clip_seed + i) % 7
    emu = _StubEmulator(rank)
    gather = EventStreamGatherer("cpu", world)
    elapsed, n_events = run_steps(emu, frames_all, F, dt, K, Wm, gather, dist, "cpu")
    ok = elapsed > 0 and len(emu.calls) == K + Wm
    # frames: the synthetic clip is cycled through (3 steps of frames), time keeps increasing step after step
    for s, (t0, t1, px) in enumerate(emu.calls):
        ok &= abs(t0 - (1 + s * F) * dt) < 1e-12 and abs(t1 - (s * F + F) * dt) < 1e-12
        ok &= px == (clip_seed + 1 + (s % 3) * F) % 7
    # events counted = the timed steps only
    exp = 0
    for s in range(Wm, Wm + K):
        c = sum(3 + rank + (s + f) % 5 for f in range(F)) + (clip_seed + 1 + (s % 3) * F) % 7
        exp += c
    ok &= n_events == exp
    # the reduction bench.py does over the ranks
    t = torch.tensor([float(n_events)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    # the last step's streams of every rank arrived everywhere, in rank order
    parts = gather.result()
    last = Wm + K - 1
    for r in range(world):
        n_r = sum(3 + r + (last + f) % 5 for f in range(F)) + (10 + r + 1 + (last % 3) * F) % 7
        ok &= parts[r].shape == (n_r, 4) and torch.equal(parts[r], _rows(n_r, r, last))
    ok &= gather.bytes_gathered > 0
    q.put((rank, bool(ok), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_loop_gloo_world2():
    """bench.py's timed loop (v2e_amd.benchutil.run_steps: clip per rank, pipelined steps, event-stream all-gather of
    every step, barrier-bracketed timing, totals reduced over ranks) on CPU tensors, two gloo ranks, engine stubbed."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)]
    assert res[0][2] == res[1][2] > 0


# ---- SloMo pair sharding of ONE clip over ranks (v2e_amd.pipeline: pair_shard, gather_frames_in_order, VideoToEvents.run(group=))
class _StubSloMoPipeline:
    """VideoToEvents with the two device stages stubbed: upsample() = a deterministic function of each source pair, the emulator a
    recorder.  Everything else (sharding, gather, frame order, times) is v2e_amd.pipeline's code."""

    def __new__(cls, U, batch_size):
        from v2e_amd.pipeline import VideoToEvents

        class P(VideoToEvents):
            def __init__(self):
                self.U, self.batch_size, self.seen = U, batch_size, None
                outer = self

                class Emu:
                    def generate_events_batch(self, frames, times, return_device=False):
                        # (every call's frames and times, concatenated: a chunked run feeds the emulator chunk by chunk)
                        if outer.seen is None:
                            outer.seen = (frames.clone(), np.asarray(times).copy())
                        else:
                            outer.seen = (torch.cat((outer.seen[0], frames)), np.concatenate((outer.seen[1], np.asarray(times))))
                        return torch.zeros((int(frames.sum()) % 7, 4)), np.asarray([int(f.sum()) % 5 for f in frames])
                self.emu = Emu()

            def upsample(self, frames_u8):
                n = int(frames_u8.shape[0]) - 1
                out = torch.empty((n * self.U,) + tuple(frames_u8.shape[1:]), dtype=torch.uint8)
                for k in range(n):
                    for u in range(self.U):
                        out[k * self.U + u] = ((frames_u8[k].to(torch.int32) * (self.U - u) + frames_u8[k + 1].to(torch.int32) * u) // self.U).to(torch.uint8)
                return out
        return P()


def _slomo_shard_worker(rank, world, port, q, n_src, U):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11)
    frames = torch.randint(0, 256, (n_src, 5, 7), dtype=torch.uint8, generator=g)
    single = _StubSloMoPipeline(U, 2)
    ev1, c1, n1 = single.run(frames, 1 / 30)
    sharded = _StubSloMoPipeline(U, 2)
    ev2, c2, n2 = sharded.run(frames, 1 / 30, group=dist.group.WORLD, owner=world - 1)
    ok = n1 == n2 == (n_src - 1) * U
    if rank == world - 1:  # the owner saw the clip's frames in the single-rank order, with the single-rank times, and produced its events
        ok &= torch.equal(sharded.seen[0], single.seen[0]) and np.array_equal(sharded.seen[1], single.seen[1])
        ok &= torch.equal(ev1, ev2) and np.array_equal(c1, c2)
    else:
        ok &= ev2 is None and c2 is None and sharded.seen is None
    every = sharded.upsample_sharded(frames, dist.group.WORLD, None)  # owner=None: the clip on every rank
    ok &= torch.equal(every, single.seen[0])
    # in chunks of 3 and of 1 source pairs (the owner's DVS stage on chunk c while chunk c + 1 is interpolated; ragged last chunk, chunks
    # with fewer pairs than ranks): the same frames in the same order with the same times, the same per-frame counts
    for cp in (3, 1):
        chunked = _StubSloMoPipeline(U, 2)
        ev3, c3, n3 = chunked.run(frames, 1 / 30, group=dist.group.WORLD, owner=world - 1, chunk_pairs=cp)
        ok &= n3 == n1
        if rank == world - 1:
            ok &= torch.equal(chunked.seen[0], single.seen[0]) and np.array_equal(chunked.seen[1], single.seen[1]) and np.array_equal(c3, c1)
        else:
            ok &= ev3 is None and c3 is None and chunked.seen is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_src,U", [(2, 9, 3), (4, 11, 2), (4, 3, 5)])
def test_slomo_pair_sharding_reproduces_the_single_rank_clip(world, n_src, U):
    """SURVEY.md 8(e) / north_star: one clip's source pairs sharded over ranks, frames funnelled in order to the rank with the emulator
    state: the same frames, frame order, times and events as one rank (ragged shards; more ranks than pairs: (4, 3, 5))."""
    from v2e_amd.pipeline import pair_shard
    P = n_src - 1
    blocks = [pair_shard(P, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == P and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slomo_shard_worker, args=(r, world, port, q, n_src, U)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == {r: True for r in range(world)}
