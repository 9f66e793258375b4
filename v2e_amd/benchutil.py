"""Side measurements reported by bench.py next to the headline number: the SuperSloMo
path (interpolated frames/s, f32-MFMA roofline) and the emulator with many clips per launch
(how far the same kernels go once the launch is large enough to leave the latency floor)."""
import os
import time

import numpy as np
import torch

F32_MFMA_PEAK = 157.3e12  # MI355X_MICROARCH.md: f32-input MFMA, dense
BF16_MFMA_PEAK = 2.5e15  # MI355X_MICROARCH.md: bf16 MFMA, dense (32x32x16)
HBM_PEAK = 8.0e12


def run_steps(emu, frames_all, F, dt, steps, warmup, gather, dist, device, first_step=0):
    """The timed loop of bench.py: `warmup` untimed steps, then `steps` steps of F frames bracketed by a barrier and a
    device synchronisation on both sides; returns (seconds, events of this rank).  `first_step`: index of the first step
    (time keeps running across consecutive calls on the same emulator).  Step s + 1 is prepared and enqueued
    while step s executes; the event stream of step s goes to `gather` (all-gather over the ranks) when there is one.
    Works on any object with EventEmulator's generate_events_batch_async (the CPU tests pass a stub)."""
    device = torch.device(device)
    cuda = device.type == "cuda"
    # Python's cyclic collector: a full (generation-2) collection over everything torch imported is a 50-70 ms pause that lands
    # in whichever 20-step block is running when the allocation counter trips (measured: scripts/step_times.py).  What exists
    # now is moved to the permanent generation once, so collections during the loop only look at what the loop allocated.
    import gc
    gc.collect()
    gc.freeze()
    nclip = max((int(frames_all.shape[0]) - 1) // F, 1)
    import inspect
    # The loop's mode: pipelined runs (plain launches on four streams, consecutive runs overlapping: v2e_emu_run 0 | 1024; the clip was
    # generated and synchronised before the loop, so the frames are resident).  V2E_AMD_BENCH_UG=1: one hipGraph per run, run after
    # run on one stream (rounds 2-5), for A/B.  (The CPU tests' stand-ins have no such switch.)
    ug = int(os.environ.get("V2E_AMD_BENCH_UG", "0"))
    kw = {}
    if "pipelined" in inspect.signature(emu.generate_events_batch_async).parameters:
        kw = dict(pipelined=os.environ.get("V2E_AMD_BENCH_PIPELINED", "1") != "0", frames_resident=True)

    def enqueue(s):
        lo = 1 + (s % nclip) * F  # the synthetic clip is cycled through; time keeps running
        # the step's frames are read where they lie in HBM: the run's kernels take the frames' address from a device variable its
        # upload fills (v2e_emu_run), so the captured graph is replayed over any buffer (rounds 1-4 baked the pointer into the graph
        # and began every step with a 27 MB device-to-device copy into one fixed buffer)
        src = frames_all[lo:lo + F]
        return emu.generate_events_batch_async(src, [(1 + s * F + i) * dt for i in range(F)], return_device=True,
                                               use_graph=ug, **kw)

    def finish(pend):
        ev, counts = pend.result()
        n = int(counts.sum())
        if gather is not None:
            # all-gather of this step's stream overlaps the next step: the side stream waits for THIS run's completion event
            # only; the blocks of one time stamp are bounded by the frames' iteration counts (4-byte wire format)
            rh = getattr(pend, "rec_host", None)
            rb = int(np.maximum(rh["max_events"], 1).sum()) if rh is not None else None
            gather.submit(ev, n, ready_event=getattr(pend, "done", None), run_bound=rb)
        return n

    def sync():
        if gather is not None:
            gather.wait()
        if cuda:
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize(device)

    def loop(first, count):
        n, pend = 0, None
        for s in range(first, first + count):
            nxt = enqueue(s)
            if pend is not None:
                n += finish(pend)
            pend = nxt
        if pend is not None:
            n += finish(pend)
        return n

    loop(first_step, warmup)
    sync()
    t0 = time.perf_counter()
    n_events = loop(first_step + warmup, steps)
    sync()
    return time.perf_counter() - t0, n_events


def unet_flops(cin, cout, h, w):
    """2*MACs of model.UNet on one [cin,h,w] sample."""
    from .synth import unet_layer_shapes
    res = {"conv1": 0, "conv2": 0, "conv3": 0}
    for d in range(1, 6):
        res["down%d" % d] = d
    for u in range(1, 6):
        res["up%d" % u] = 5 - u
    macs = 0
    for name, co, ci, k in unet_layer_shapes(cin, cout):
        lvl = res[name.split(".")[0]]
        macs += (h >> lvl) * (w >> lvl) * co * ci * k * k
    return 2 * macs


def slomo_bench(device, B=8, U=10, H=256, W=320, iters=5, conv_math=None):
    """Interpolated frames/s of slomo.py:338-433 at 320x256 (346x260 source), 10x slowdown.

    One iteration = one batch of B source pairs (B = 8, the v2e CLI default --batch_size) -> U*B interpolated frames: flow UNet on B
    samples, prep, interpolation UNet on U*B samples, fuse.  Inputs resident in HBM.
    """
    from .slomo import SloMoEngine
    from .synth import portable_unet_state_dict
    sd_f, sd_i = portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102)
    eng = SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()},
                      {k: torch.from_numpy(v) for k, v in sd_i.items()}, device, conv_math=conv_math)
    g = torch.Generator(device=device)
    g.manual_seed(2)
    I0 = torch.rand((B, 1, H, W), device=device, generator=g) - 0.428
    I1 = torch.rand((B, 1, H, W), device=device, generator=g) - 0.428
    ts = [(k + 0.5) / U for k in range(U)]
    # as SuperSloMo.interpolate runs its batches: the flow UNet of the NEXT batch is started (side stream) before this batch's
    # interpolation UNet is enqueued -- every iteration still executes one flow pass and one interpolation pass
    eng.interpolate(I0, I1, ts, next_pair=(I0, I1))  # warm-up (allocations)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.interpolate(I0, I1, ts, next_pair=(I0, I1))
    e1.record()
    torch.cuda.synchronize(device)
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    flops = B * unet_flops(2, 4, H, W) + U * B * unet_flops(12, 5, H, W)
    # time of the interpolation UNet alone (the dominant kernels), for the roofline object
    x12 = eng.last["x12"]
    e0.record()
    for _ in range(iters):
        eng.interp_net.forward(x12)
    e1.record()
    torch.cuda.synchronize(device)
    sec_u = e0.elapsed_time(e1) * 1e-3 / iters
    fl_u = U * B * unet_flops(12, 5, H, W)
    fallbacks = eng.flow_net.fallbacks + eng.interp_net.fallbacks
    math_run = eng.conv_math
    if math_run == "auto":  # what actually ran: the two-piece math, unless the range guard redid passes with the exact split
        math_run = "fp16x2" if (fallbacks == 0 and eng.interp_net.descs_exact is not None) else "bf16x3"
    s3 = math_run in ("bf16x3", "fp16x2")
    npr = {"bf16x3": 6, "fp16x2": 3}.get(math_run, 1)  # piece products executed per f32 multiply
    # executed matrix-core work: with split-bf16 operands every f32 multiply-add is six bf16 ones (the 12-channel conv1,
    # the 8x10 level and the 5-channel head stay on f32 instructions: ~6 % of the FLOPs)
    ach = fl_u / sec_u
    roof = {"bound": "mfma", "kernel": ("k_conv_s3 / k_conv_s3p" if s3 else "k_conv") + " (23 launches of the interpolation UNet)",
            "achieved": round(ach * npr / 1e12, 2), "peak": (BF16_MFMA_PEAK if s3 else F32_MFMA_PEAK) / 1e12,
            "unit": "TFLOP/s", "frac": round(ach * npr / (BF16_MFMA_PEAK if s3 else F32_MFMA_PEAK), 4),
            # (PMC passes are committed for the 80-sample 320x256 forward in every conv math and for 2 samples at 1280x704)
            "traffic": slomo_pmc_traffic(math_run, unet_algorithmic_bytes(U * B, 12, 5, H, W), "%dx%dx%d" % (U * B, H, W))[0],
            "traffic_detail": slomo_pmc_traffic(math_run, unet_algorithmic_bytes(U * B, 12, 5, H, W), "%dx%dx%d" % (U * B, H, W))[1],
            "f32_equivalent_TFLOPs": round(ach / 1e12, 2),
            "f32_equivalent_vs_f32_mfma_peak": round(ach / F32_MFMA_PEAK, 4),
            "whole_step_TFLOPs": round(flops / sec / 1e12, 2),
            "note": ("achieved = algorithmic f32 FLOPs of the UNet x 6 (bf16 piece products executed per f32 multiply: x = p0+p1+p2 "
                     "exactly, products with i+j<=2) / time, against the dense bf16 MFMA peak; f32_equivalent is the same time "
                     "priced in the algorithm's own f32 FLOPs (the f32 matrix-core peak is 157.3 TF/s)") if math_run == "bf16x3" else
                    ("achieved = algorithmic f32 FLOPs of the UNet x 3 (float16 piece products executed per f32 multiply: x = h0 + h1 + r, "
                     "|r| <= 2^-22 |x|; products h0 g0, h0 g1, h1 g0) / time, against the dense f16 MFMA peak (= the bf16 one)")
                    if math_run == "fp16x2" else "f32 matrix-core instructions (v_mfma_f32_32x32x2_f32)"}
    return {
        "metric": "interpolated frames/s (SuperSloMo flow UNet + per-t warps + interpolation UNet + fusion)",
        "value": round(U * B / sec, 2), "unit": "frames/s",
        "config": {"workload": ("BASELINE configs[2] SloMo stage: 320x256 (346x260 source), U=%d, batch of %d pairs, "
                                "seeded random-init weights (checkpoint not available offline)" % (U, B)) if (H, W) == (256, 320) else
                               ("SloMo stage at %dx%d (SURVEY 8(a): a 1280x720 source is interpolated at 1280x704), U=%d, batch of %d "
                                "pair(s), seeded random-init weights" % (W, H, U, B))},
        "dtype": "f32" + ({"bf16x3": " (operands split exactly into 3 bf16 pieces, 6 products on the bf16 matrix cores, f32 accumulation)",
                           "fp16x2": " (operands split into 2 float16 pieces, residual <= 2^-22; 3 products on the f16 matrix cores, f32 accumulation)"}
                          .get(math_run, "")),
        "conv_math": eng.conv_math, "conv_math_run": math_run, "range_guard_fallbacks": fallbacks,
        "ms_per_batch": round(sec * 1e3, 3),
        "gflop_per_frame": round(flops / (U * B) / 1e9, 2),
        "roofline": roof,
    }


def slomo_pmc_traffic(conv_math, algorithmic=None, shape="80x256x320"):
    """HBM bytes per interpolation-UNet forward of the conv math that ran, from the committed rocprofv3 PMC passes of THIS round's
    kernels (profiles/r06_slomo_counters.txt, lines '# unet_forward_bytes <conv_math> <fetch> <write> <samples>x<H>x<W>', made by
    scripts/gpu_r06_profiles.sh + scripts/make_profiles_r06.py; round 5's file is the fall-back).  Returns (bytes or None, detail dict)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("r06_slomo_counters.txt", "r05_slomo_counters.txt"):
        try:
            for line in open(os.path.join(root, "profiles", name)):
                if line.startswith("# unet_forward_bytes"):
                    parts = line.split()
                    if parts[2] == conv_math and (parts[5] if len(parts) > 5 else "80x256x320") == shape:
                        fetch, write = float(parts[3]), float(parts[4])
                        d = {"source": "profiles/" + name, "shape": shape, "fetch_bytes": int(fetch), "write_bytes": int(write)}
                        if algorithmic:
                            d["algorithmic_bytes"] = int(algorithmic)
                            d["traffic_over_algorithmic"] = round((fetch + write) / algorithmic, 3)
                        return int(fetch + write), d
        except Exception:
            pass
    return None, {"source": None, "note": "no PMC pass of this conv math and shape is committed"}


def unet_algorithmic_bytes(n, cin, cout, h, w):
    """Bytes one UNet forward has to move if every activation is written once and read once by its consumer(s) (skip
    connections twice), plus the weights once: the 'algorithmic' side of the traffic ratio."""
    from .synth import unet_layer_shapes
    res = {"conv1": 0, "conv2": 0, "conv3": 0}
    for d in range(1, 6):
        res["down%d" % d] = d
    for u in range(1, 6):
        res["up%d" % u] = 5 - u
    byts = 0
    for name, co, ci, k in unet_layer_shapes(cin, cout):
        lvl = res[name.split(".")[0]]
        hw = (h >> lvl) * (w >> lvl)
        byts += 4 * n * hw * (ci + co) + 4 * co * ci * k * k  # read the input, write the output, the weights
    return byts


def batched_emulator_bench(device, n_clips=64, frames=60, H=260, W=346):
    """Same emulator kernels, `n_clips` independent 346x260 clips advanced per launch."""
    from .emulator import EventEmulator
    from .engine import EmuEngine
    kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01,
              shot_noise_rate_hz=.001, refractory_period_s=.0005)
    proto = EventEmulator(device=device, seed=3, rng_mode="philox", **kw)
    proto._thres_scalar = (0.2, 0.2)
    proto._thres_is_scalar = False
    P = proto._params()
    eng = EmuEngine(H, W, n_clips=n_clips, device=device)
    eng.alloc_state(True)
    g = torch.Generator(device=device)
    g.manual_seed(5)
    y = torch.arange(H, device=device, dtype=torch.float32).view(1, 1, H, 1)
    x = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, 1, W)
    i = torch.arange(frames + 1, device=device, dtype=torch.float32).view(-1, 1, 1, 1)
    ph = torch.arange(n_clips, device=device, dtype=torch.float32).view(1, -1, 1, 1) * 7.0
    f = 127 + 100 * torch.sin((x + 3 * i + ph) / 15.0) * torch.cos((y - 2 * i) / 20.0)
    f = f + 3.0 * torch.randn(f.shape, device=device, generator=g)
    fr = f.clamp_(0, 255).to(torch.uint8).contiguous()  # [F+1][NC][H][W]
    del f
    eng.init_state(P, fr[0].contiguous(), 0.0)
    cap = 80000 * frames
    evs = [eng.event_buffer(cap, 0), eng.event_buffer(cap, 1)]   # two buffer sets alternate: run k + 1 is enqueued before run k is read,
    recs = [eng.alloc_recs(frames, 0), eng.alloc_recs(frames, 1)]  # as in the headline loop and the 1280x720 leg
    dt = 1.0 / 300.0
    frames_run = fr[1:].contiguous()
    ug = int(os.environ.get("V2E_AMD_BATCHED_UG", "1"))

    def enqueue(k):
        t_prev = np.array([[(k * frames + j) * dt] * n_clips for j in range(frames)])
        t_frame = t_prev + dt
        eng.run(P, frames_run, t_prev, t_frame, 1 + k * frames, evs[k & 1], recs[k & 1], use_graph=ug)
        ticket = eng.run_ticket() if (ug & 1024) else -1
        if ticket >= 0:  # pipelined: the library hands the records over itself (v2e_emu_run_wait / _run_recs)
            return k & 1, ticket
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(device))
        return k & 1, done

    def collect(pend):
        which, done = pend
        if isinstance(done, int):
            eng.run_wait(done)
            return int(eng.run_recs(done)["n_events"].sum())
        return int(eng.read_recs_after(recs[which], done)["n_events"].sum())

    for k in range(2):
        collect(enqueue(k))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 6
    n_ev, pend = 0, None
    for k in range(2, 2 + reps):
        nxt = enqueue(k)
        if pend is not None:
            n_ev += collect(pend)
        pend = nxt
    n_ev += collect(pend)
    torch.cuda.synchronize(device)
    sec = time.perf_counter() - t0
    bpp = 53
    byts = bpp * H * W * n_clips * frames * reps + 16 * n_ev
    return {"value": round(n_ev / sec / 1e6, 1), "unit": "Mevents/s", "clips_per_launch": n_clips,
            "frames_per_s_all_clips": round(n_clips * frames * reps / sec, 1),
            "algorithmic_GBps": round(byts / sec / 1e9, 1), "hbm_frac": round(byts / sec / HBM_PEAK, 4), "event_writer": eng.event_writer(),
            "note": "same kernels as the headline run, %d clips advanced per launch" % n_clips}


def hd_noisy_emulator_bench(device, frames=64, H=720, W=1280, reps=6):
    """BASELINE configs[3]: 1280x720, set_dvs_params('noisy'), 20x slowdown (dt = 1/600 s): compaction stress.
    Runs of `frames` frames, the host preparing run n + 1 while run n executes (generate_events_batch_async)."""
    import bench as B
    from .emulator import EventEmulator
    fr = B.gen_frames_device(frames + 1, 4, device, h=H, w=W)
    emu = EventEmulator(device=device, seed=4, rng_mode="philox", **B.DEFAULT_KW)
    emu.set_dvs_params("noisy")
    dt = 1.0 / 600.0
    emu.generate_events(fr[0], 0.0)
    buf = fr[1:].contiguous()
    cap = 400_000 * frames

    def enqueue(k):
        # (pipelined runs, the headline loop's mode: with the packets the chain's stream no longer carries -- round 6, experiments 23 and 30 --
        # 13.1-13.65 Gev/s against 11.5-12.4 with one hipGraph per run, V2E_AMD_HD_UG=1; earlier in the round it was 11.3-11.4 against 11.5-12.0)
        return emu.generate_events_batch_async(buf, [(1 + k * frames + i) * dt for i in range(frames)], return_device=True, cap=cap,
                                               use_graph=int(os.environ.get("V2E_AMD_HD_UG", "0")), frames_resident=True)

    for k in range(2):
        enqueue(k).result()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n_ev, pend = 0, None
    for k in range(2, 2 + reps):
        nxt = enqueue(k)
        if pend is not None:
            n_ev += int(pend.result()[1].sum())
        pend = nxt
    n_ev += int(pend.result()[1].sum())
    torch.cuda.synchronize(device)
    sec = time.perf_counter() - t0
    bpp = 45  # noisy preset: cutoff, leak, shot, no refractory
    byts = bpp * H * W * frames * reps + 16 * n_ev
    kind, fpl, fpb = emu._engine.last_pipeline()
    return {"value": round(n_ev / sec / 1e6, 1), "unit": "Mevents/s", "frames_per_s": round(frames * reps / sec, 1),
            "events_per_frame": round(n_ev / (frames * reps), 1), "algorithmic_GBps": round(byts / sec / 1e9, 1),
            "hbm_frac": round(byts / sec / HBM_PEAK, 4), "pipeline": "%s, %d frames per launch" % (kind, fpl),
            "event_writer": emu._engine.event_writer(),
            "config": "BASELINE configs[3]: 1280x720, dvs_params noisy, dt=1/600 s, one clip, Philox"}


def e2e_bench(device, n_src=31, U=10, H=260, W=346, batch=10, reps=3):
    """BASELINE configs[2]: 346x260 uniform-random uint8 video (seed 2), 31 source frames @30 fps, U=10
    -> 300 interpolated frames at 320x256 -> quantise/resize -> emulator, everything in HBM
    (VideoToEvents).  Seeded random-init UNet weights."""
    from .emulator import EventEmulator
    from .pipeline import VideoToEvents
    from .slomo import SloMoEngine
    from .synth import portable_unet_state_dict
    import bench as B
    sd_f, sd_i = portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102)
    eng = SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()},
                      {k: torch.from_numpy(v) for k, v in sd_i.items()}, device)
    g = torch.Generator(device=device)
    g.manual_seed(2)
    src = torch.randint(0, 256, (n_src, H, W), dtype=torch.uint8, device=device, generator=g)
    dt_src = 1.0 / 30.0

    def once(seed):
        emu = EventEmulator(device=device, seed=seed, rng_mode="philox", **B.DEFAULT_KW)
        pipe = VideoToEvents(eng, emu, U, batch_size=batch)
        ev, counts, nfr = pipe.run(src, dt_src, return_device=True)
        return int(counts.sum()), nfr

    once(1)  # warm-up: allocations, graph capture
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n_ev = n_fr = 0
    for r in range(reps):
        a, b = once(2 + r)
        n_ev += a
        n_fr += b
    torch.cuda.synchronize(device)
    sec = time.perf_counter() - t0
    return {"interpolated_frames_per_s": round(n_fr / sec, 1), "Mevents_per_s": round(n_ev / sec / 1e6, 1),
            "events_per_frame": round(n_ev / n_fr, 1), "ms_per_clip": round(sec / reps * 1e3, 2),
            "config": "BASELINE configs[2]: 346x260 random uint8 video, %d source frames, U=%d, SloMo HIP -> Pillow-exact "
                      "quantise/resize on device -> emulator HIP, batch %d pairs, frames never leave HBM" % (n_src, U, batch)}


def slomo_sharded_bench(device, dist, n_src=65, U=10, H=260, W=346, batch=8, reps=3, pipe=None):
    """ONE clip's SuperSloMo stage sharded over the ranks of the job by source pairs (north_star: "frame batches shard across the
    GPUs"): every rank holds the same 346x260 source clip (seed 2), interpolates its contiguous block of the 64 pairs
    (VideoToEvents.upsample_sharded) and the uint8 frames are sent in order to rank 0 (point to point, exact sizes).  STRONG scaling: the clip is fixed,
    the ranks split it; value = interpolated frames of the whole clip / wall time (barrier + synchronise on both sides, max over
    ranks by construction of the barrier).  At world size 1 it is the unsharded stage with a one-rank gather.
    pipe: an object with upsample_sharded(frames, group, owner) and upsample(frames) in place of the HIP pipeline (bench.py's stub
    mode on CPU: the collective sequence of this leg is then exercised by tests/test_bench_launch.py); the gathered clip is compared
    with the unsharded one on rank 0 in that case."""
    device = torch.device(device)
    on_gpu = device.type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize(device)

    stub = pipe is not None
    if pipe is None:
        from .pipeline import VideoToEvents
        from .slomo import SloMoEngine
        from .synth import portable_unet_state_dict
        sd_f, sd_i = portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102)
        eng = SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()},
                          {k: torch.from_numpy(v) for k, v in sd_i.items()}, device)
        pipe = VideoToEvents(eng, None, U, batch_size=batch)
    g = torch.Generator(device=device)
    g.manual_seed(2)
    src = torch.randint(0, 256, (n_src, H, W), dtype=torch.uint8, device=device, generator=g)
    world = dist.get_world_size()

    def agreed(fn):
        """fn() on every rank; an exception on ANY rank is agreed on (an all-reduce of the error flags) before anybody enters the next
        collective -- a rank that raised would otherwise leave its peers waiting in it."""
        err = None
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            res, err = None, repr(e)[:200]
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag)
        if int(flag.item()):
            raise RuntimeError("slomo_sharded: %d rank(s) failed%s" % (int(flag.item()), (": " + err) if err else ""))
        return res

    out = agreed(lambda: pipe.upsample_sharded(src, dist.group.WORLD, 0))  # warm-up: allocations, RCCL channels
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = agreed(lambda: pipe.upsample_sharded(src, dist.group.WORLD, 0))
    sync()
    dist.barrier()
    sync()
    sec = (time.perf_counter() - t0) / reps
    n = (n_src - 1) * U
    # rank 0: the gathered clip against the unsharded stage on the same source (the shards move the interpolation's batch boundaries:
    # with the default conv math the activation scale is per batch tensor, so the uint8 frames may differ by a grey level here and there)
    maxdiff = frac_diff = None
    if out is not None:
        ref = pipe.upsample(src)
        d = (out.to(torch.int16) - ref.to(torch.int16)).abs()
        maxdiff, frac_diff = int(d.max().item()), float((d > 0).float().mean().item())
    ok = out is None or (tuple(out.shape) == (n, H, W) and maxdiff <= (0 if stub else 1))
    return {"value": round(n / sec, 1), "unit": "interpolated frames/s", "scaling": "strong", "ranks": world, "ms_per_clip": round(sec * 1e3, 2),
            "frames_gathered_in_order": bool(ok), "max_abs_diff_vs_unsharded": maxdiff, "fraction_of_pixels_differing": frac_diff,
            "bytes_received_by_owner_per_clip": int(n * H * W * (world - 1) / max(world, 1)),
            "config": "one %dx%d clip, %d source pairs, U = %d, pairs [r P / G, (r + 1) P / G) per rank, uint8 frames sent in order "
                      "to rank 0, point to point (v2e_amd.pipeline.VideoToEvents.upsample_sharded)" % (W, H, n_src - 1, U)}
