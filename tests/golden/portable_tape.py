"""A random source that is the same on every host and torch build (test infrastructure).

Tape mode of the drop-in replays the reference's torch MT19937 stream, so its sensor-size fixtures (tape_live_*) can only be
compared where this host's torch draws what the fixture host's torch drew.  `PortableTape` implements the tape interface
(oracle.oracle.TorchTape: normal / randn / rand / randperm / linspace / exp_*) from integer hashing and exactly representable
float conversions only, so the TAPE-MODE kernels (k_count -> k_rank / k_scan -> host permutations -> k_shot -> k_emit -> k_permute)
are pinned at 346x260 independently of torch's generator: tests/golden/make_golden_tape_portable.py runs the unmodified reference
with torch.normal / randn / rand / randperm replaced by this source, the tests feed the same source to the oracle and to the HIP path.

  call c of the run (counted over normal, randn, rand, randperm), element i:
    h(c, i, j) = splitmix64(seed + c * 0xD1B54A32D192ED03 + (i * 16 + j + 1) * 0x9E3779B97F4A7C15)    (uint64, wrapping)
    rand      float32(top 24 bits of h(c, i, 0) / 2^24)                       exact
    randn     float32(sum_{j<12} top24(h(c, i, j)) / 2^24 - 6)                Irwin-Hall(12): exact in float64, one rounding to float32
    normal    float32(mean) + float32(std) * randn                            float32 arithmetic
    randperm  stable argsort of h(c, i, 0)
  linspace is not a draw: it is torch.linspace itself (what the drop-in calls on the host in tape mode).
  exp_noise_rate / exp_scidvs are torch.exp; the fixtures use noise_rate_cov_decades = 0, where exp(0 * r) = 1 on every host.
"""
import math

import numpy as np

_M1, _M2 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)
_G, _C = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xD1B54A32D192ED03)


def _mix(x):
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


class PortableTape:
    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.calls = 0

    def _h(self, n, j=0):
        with np.errstate(over="ignore"):
            c = np.uint64(self.calls)
            i = np.arange(n, dtype=np.uint64)
            return _mix(self.seed + c * _C + (i * np.uint64(16) + np.uint64(j + 1)) * _G)

    def _u24(self, n, j=0):
        return (self._h(n, j) >> np.uint64(40)).astype(np.int64)

    def rand(self, shape):
        n = int(np.prod(shape))
        r = (self._u24(n).astype(np.float64) / 16777216.0).astype(np.float32).reshape(shape)
        self.calls += 1
        return r

    def randn(self, shape):
        n = int(np.prod(shape))
        s = np.zeros(n, np.int64)
        for j in range(12):
            s += self._u24(n, j)
        r = (s.astype(np.float64) / 16777216.0 - 6.0).astype(np.float32).reshape(shape)
        self.calls += 1
        return r

    def normal(self, mean, std, shape):
        return (np.float32(mean) + np.float32(std) * self.randn(shape)).astype(np.float32)

    def randperm(self, n, frame=None, it=None):
        p = np.argsort(self._h(int(n)), kind="stable").astype(np.int64)
        self.calls += 1
        return p

    def linspace(self, start, end, n):
        import torch
        return torch.linspace(start=start, end=end, steps=n, dtype=torch.float32).numpy()

    def exp_noise_rate(self, cov, randn):
        import torch
        return torch.exp(math.log(10) * cov * torch.from_numpy(np.ascontiguousarray(randn))).numpy()

    def exp_scidvs(self, draw):
        import torch
        return torch.exp(torch.from_numpy(np.ascontiguousarray(draw))).numpy()


class PortableSource:
    """Context manager: the reference's torch.normal / randn / rand / randperm draw from a PortableTape (same call order)."""

    def __init__(self, seed):
        self.tape = PortableTape(seed)
        self._orig = {}

    def __enter__(self):
        import torch
        for name in ("normal", "randn", "rand", "randperm"):
            self._orig[name] = getattr(torch, name)
        t = self.tape

        def _shape(a, k):
            s = k.get("size", a[0] if a else None)
            return tuple(s) if not isinstance(s, int) else (s,)

        def normal(mean, std, size=None, **k):
            return torch.from_numpy(t.normal(mean, std, tuple(size)))

        def randn(*a, **k):
            return torch.from_numpy(t.randn(_shape(a, k)))

        def rand(*a, **k):
            return torch.from_numpy(t.rand(_shape(a, k)))

        def randperm(n, *a, **k):
            return torch.from_numpy(t.randperm(n))

        torch.normal, torch.randn, torch.rand, torch.randperm = normal, randn, rand, randperm
        return self

    def __exit__(self, *exc):
        import torch
        for name, fn in self._orig.items():
            setattr(torch, name, fn)
