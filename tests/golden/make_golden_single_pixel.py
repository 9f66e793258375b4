#!/usr/bin/env python
"""The reference's single-pixel recorder (`record_single_pixel_states`, emulator.py:279-300, 985-1009) on two of the tape
fixtures' configurations: same frames, times, parameters and seed as tape_defaults_40x48 / tape_refractory_float_33x37
(make_golden.py; the recorder draws no random numbers, so those fixtures' tapes replay this run too), pixel index tuples
(7, 11) and (5, 9) -- the reference applies the tuple to its [H, W] planes as it is.

  single_pixel.npz   per case the ten recorded series (time, new_frame, base_log_frame, lp_log_frame, log_new_frame,
                     pos_thres, neg_thres, diff_frame, final_neg_evts_frame, final_pos_evts_frame), first n samples
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402

CASES = {"tape_defaults_40x48": (7, 11), "tape_refractory_float_33x37": (5, 9)}


def main():
    import json
    import logging
    logging.disable(logging.CRITICAL)
    EE = rh.ref_emulator_cls()
    torch.set_num_threads(1)
    out = {}
    for name, ij in CASES.items():
        z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
        kw = json.loads(str(z["kw"]))
        ref = EE(seed=int(z["seed"]), device="cpu", record_single_pixel_states=ij, **kw)
        evs = [ref.generate_events(f, float(t)) for f, t in zip(z["frames"], z["times"])]
        for k, e in enumerate(evs):  # the same run as the tape fixture
            g = z["ev_%04d" % k]
            assert (e is None and len(g) == 0) or np.array_equal(e, g), (name, k)
        n = ref.single_pixel_sample_count
        for key, arr in ref.single_pixel_states.items():
            out["%s__%s" % (name, key)] = np.asarray(arr[:n], np.float64)
        out["%s__pixel" % name] = np.asarray(ij)
        ref.record_single_pixel_states = None  # no pickle file from the atexit cleanup
        print("%-30s pixel %s: %d samples, events at the pixel on/off %d/%d" % (
            name, ij, n, int(np.nansum(ref.single_pixel_states["final_pos_evts_frame"][:n])),
            int(np.nansum(ref.single_pixel_states["final_neg_evts_frame"][:n]))))
    np.savez_compressed(os.path.join(HERE, "single_pixel.npz"), torch_version=torch.__version__, **out)


if __name__ == "__main__":
    main()
