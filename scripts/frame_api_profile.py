"""Where a frame of the frame API (EventEmulator.generate_events, host numpy in / out, Philox mode, 346x260) spends its wall time:
the loop bench.py's frame_api leg runs, once plain (frames/s) and once under cProfile (the C call against the Python around it).
Run on the MI355X box: python scripts/frame_api_profile.py [philox|tape] [frames]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from v2e_amd import EventEmulator  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "philox"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    frames_all = B.gen_frames_device(n + 1, 1, torch.device("cuda"))
    host = frames_all.cpu().numpy()
    emu = EventEmulator(device="cuda", seed=1, rng_mode=mode, **B.DEFAULT_KW)
    emu.generate_events(host[0], 0.0)
    for i in range(1, 21):
        emu.generate_events(host[i], i * B.DT)

    def loop(lo, hi):
        ne = 0
        for i in range(lo, hi):
            e = emu.generate_events(host[i], i * B.DT)
            ne += 0 if e is None else len(e)
        return ne

    half = 21 + (n - 20) // 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ne = loop(21, half)
    sec = time.perf_counter() - t0
    print(mode + ": %.1f frames/s, %.1f us per frame, %.0f events per frame" % ((half - 21) / sec, sec / (half - 21) * 1e6, ne / (half - 21)))
    pr = cProfile.Profile()
    pr.enable()
    loop(half, n + 1)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
