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


def unet_layer_shapes(cin, cout):
    """(name, cout, cin, k) of the 23 convolutions of model.UNet (model.py:184-196)."""
    L = [("conv1", 32, cin, 7), ("conv2", 32, 32, 7)]
    ch = [32, 64, 128, 256, 512, 512]
    ks = [5, 3, 3, 3, 3]
    for d in range(1, 6):
        L.append(("down%d.conv1" % d, ch[d], ch[d - 1], ks[d - 1]))
        L.append(("down%d.conv2" % d, ch[d], ch[d], ks[d - 1]))
    up_in = [512, 512, 256, 128, 64]
    up_out = [512, 256, 128, 64, 32]
    for u in range(1, 6):
        L.append(("up%d.conv1" % u, up_out[u - 1], up_in[u - 1], 3))
        L.append(("up%d.conv2" % u, up_out[u - 1], 2 * up_out[u - 1], 3))
    L.append(("conv3", cout, 32, 3))
    return L


def portable_unet_state_dict(cin, cout, seed, gain=1.7):
    """Random-init UNet weights (numpy, torch key names) from integer draws only, so every
    host regenerates the same bits: uniform(-b, b), b = gain / sqrt(fan_in) (the pretrained
    SuperSloMo39.ckpt is not obtainable offline)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, co, ci, k in unet_layer_shapes(cin, cout):
        bound = np.float32(gain / np.sqrt(ci * k * k))
        u = rng.integers(0, 1 << 24, size=(co, ci, k, k), dtype=np.int64).astype(np.float32) / np.float32(1 << 23) - np.float32(1.0)
        sd[name + ".weight"] = (u * bound).astype(np.float32)
        ub = rng.integers(0, 1 << 24, size=(co,), dtype=np.int64).astype(np.float32) / np.float32(1 << 23) - np.float32(1.0)
        sd[name + ".bias"] = (ub * bound).astype(np.float32)
    return sd
