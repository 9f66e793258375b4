"""Synthetic input clips for tests and bench.py (no datasets are reachable offline).

`int_gradient_frames` uses integer arithmetic only (PCG64 bounded integers + integer
triangle waves), so the same frames are produced bit for bit on any host -- golden
fixtures store only its arguments.  `sincos_gradient_frames` is the SURVEY.md section 8(d)
"random-gradient video" formula used for throughput runs.
"""
import numpy as np


def _tri(v, period, amp):
    """integer triangle wave of `period` in [-amp, amp]"""
    half = period // 2
    m = np.mod(v, period)
    return (np.abs(m - half) * (2 * amp)) // half - amp


def int_gradient_frames(n, H, W, seed=1, noise=6, as_array=False):
    """n uint8 frames [H,W]: a drifting 2-D triangle-wave gradient plus integer noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    out = []
    for i in range(n):
        a = _tri(x + 3 * i, 94, 64)
        b = _tri(y - 2 * i + 1000 * 126, 126, 64)
        f = 127 + (100 * a * b) // 4096
        if noise > 0:
            nz = rng.integers(-noise, noise + 1, size=(4, H, W), dtype=np.int64).sum(axis=0) // 2
            f = f + nz
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out) if as_array else out


def sincos_gradient_frames(n, H, W, seed=1, sigma=3.0):
    """frame_i[y,x] = clip(127 + 100 sin((x+3i)/15) cos((y-2i)/20) + sigma N(0,1), 0, 255) as uint8."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    out = np.empty((n, H, W), np.uint8)
    for i in range(n):
        f = 127 + 100 * np.sin((x + 3 * i) / 15.0) * np.cos((y - 2 * i) / 20.0)
        if sigma > 0:
            f = f + sigma * rng.standard_normal((H, W))
        out[i] = np.clip(f, 0, 255).astype(np.uint8)
    return out
