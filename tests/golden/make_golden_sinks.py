#!/usr/bin/env python
"""Golden bytes of the reference's AEDAT-2.0 writer (v2ecore/output/aedat2_output.py, imported from
/root/reference) and of the HDF5 conversion lines (emulator.py:955-965) for a seeded event list."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def sample_events(n, w, h, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.sort(rng.integers(0, 40_000_000, size=n)).astype(np.float32) * np.float32(1e-6)  # up to 40 s
    t = t + np.float32(1e-7) * rng.integers(0, 10, size=n).astype(np.float32)
    ev = np.stack([t, rng.integers(0, w, size=n).astype(np.float32), rng.integers(0, h, size=n).astype(np.float32),
                   rng.integers(0, 2, size=n).astype(np.float32) * 2 - 1], axis=1).astype(np.float32)
    return ev


def main():
    rh.install_stubs()
    from v2ecore.output.aedat2_output import AEDat2Output
    out = {}
    for (w, h) in ((346, 260), (640, 480)):
        ev = sample_events(5000, w, h, w)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "x.aedat")
            wr = AEDat2Output(path, output_width=w, output_height=h, label_signal_noise=True)
            hdr = os.path.getsize(path) if os.path.exists(path) else 0
            wr.file.flush()
            hdr = os.path.getsize(path)
            label = np.ones(len(ev), bool)
            label[4000:] = False
            wr.appendEvents(ev, signnoise_label=label)
            wr.close()
            data = open(path, "rb").read()[hdr:]
        out["ev_%dx%d" % (w, h)] = ev
        out["aedat2_%dx%d" % (w, h)] = np.frombuffer(data, dtype=np.uint8)
        # emulator.py:957-960
        temp = np.array(ev, dtype=np.float32)
        temp[:, 0] = temp[:, 0] * 1e6
        temp[temp[:, 3] == -1, 3] = 0
        out["h5_%dx%d" % (w, h)] = temp.astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "sinks.npz"), **out)
    print("sinks.npz", os.path.getsize(os.path.join(HERE, "sinks.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
