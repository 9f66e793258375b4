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
