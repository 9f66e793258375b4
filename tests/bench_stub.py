"""Stub engine for bench.py's step loop (test infrastructure; never a measurement).

Stands in for v2e_amd.EventEmulator where no GPU exists: `generate_events_batch_async(frames, times)` returns a handle whose
result() is (event rows, per-frame counts), a deterministic function of (rank, step, first pixel of the frames the loop
copied into its buffer).  Used by tests/test_dist_cpu.py (run_steps over two gloo ranks) and by `V2E_AMD_BENCH_STUB=1
python bench.py --gpus N` (tests/test_bench_launch.py: the launcher, process group, reductions and JSON line of bench.py).
"""
import numpy as np
import torch


def rows(n, rank, step):
    """n event rows (t, x, y, p) that differ per rank and step; t includes a negative value (sign bit on the wire)."""
    k = torch.arange(n, dtype=torch.float32)
    return torch.stack([k * 1e-3 - 0.002 + rank + 0.1 * step, (k + 13 * rank) % 346, (k * 7 + step) % 260, (k % 2) * 2 - 1], dim=1).contiguous()


class StubPending:
    def __init__(self, ev, counts):
        self.ev, self.counts = ev, counts

    def result(self):
        return self.ev, self.counts


def stub_counts(rank, step, F, first_pixel):
    counts = np.asarray([(3 + rank + (step + f) % 5) for f in range(F)], dtype=np.int64)
    counts[0] += int(first_pixel)  # depends on the frames the loop copied into its buffer
    return counts


class StubEmulator:
    def __init__(self, rank):
        self.rank, self.step, self.calls = rank, 0, []

    def generate_events_batch_async(self, frames, times, return_device=True, use_graph=True):
        counts = stub_counts(self.rank, self.step, len(times), frames[0, 0, 0])
        ev = rows(int(counts.sum()), self.rank, self.step)
        self.calls.append((float(times[0]), float(times[-1]), int(frames[0, 0, 0])))
        self.step += 1
        return StubPending(ev, counts)


class StubUpsampler:
    """Stands in for VideoToEvents' SuperSloMo stage in bench.py's `slomo_sharded` leg: upsample() is a deterministic blend of each
    source pair; the sharding and the in-order gather are v2e_amd.pipeline's own code (pair_shard, gather_frames_in_order)."""

    def __init__(self, U):
        self.U = int(U)

    def upsample(self, frames_u8):
        n, U = int(frames_u8.shape[0]) - 1, self.U
        a = frames_u8[:-1].to(torch.int32)[:, None]
        b = frames_u8[1:].to(torch.int32)[:, None]
        u = torch.arange(U, dtype=torch.int32).view(1, U, 1, 1)
        return ((a * (U - u) + b * u) // U).to(torch.uint8).reshape((n * U,) + tuple(frames_u8.shape[1:]))

    def upsample_sharded(self, frames_u8, group=None, owner=None):
        import torch.distributed as dist
        from v2e_amd.pipeline import gather_frames_in_order, pair_shard
        G, r = dist.get_world_size(group), dist.get_rank(group)
        N = int(frames_u8.shape[0])
        lo, hi = pair_shard(N - 1, G, r)
        mine = self.upsample(frames_u8[lo:hi + 1]) if hi > lo else frames_u8.new_empty((0,) + tuple(frames_u8.shape[1:]))
        return gather_frames_in_order(mine, N - 1, self.U, group, owner)
