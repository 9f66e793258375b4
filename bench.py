#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric on MI355X.

Workload (BASELINE.json configs[1]): a 346x260
"random-gradient" video at 10x slowdown (dt = 1/300 s), emulator-only: the whole
EventEmulator.generate_events path (v2ecore/emulator.py:619-1022) as HIP kernels, CLI
default DVS parameters (v2e_args.py:150-204), Philox RNG, frames resident in HBM.
One *step* = one second of source video = 300 emulator frames advanced on device with no
host synchronisation in between; the host prepares and enqueues step s+1 while step s
executes (EventEmulator.generate_events_batch_async) and then reads step s's records.
value = events emitted / wall time (Mevents/s), whole job.  The timed region of `--steps K` steps (bracketed by barrier +
device synchronisation on both sides, max over ranks) is repeated `--blocks` times (default 9) and the MEDIAN block is the
one reported (`ms_per_step`, `value`), with the spread of the blocks beside it: a 20-step region is 25 ms, and the first
region after start-up runs a few percent slow.

N > 1 (driver: python -m torch.distributed.run ...): one independent clip per GPU
(BASELINE.json configs[4]); every step ends with an RCCL all-gather of the ranks' event
streams, overlapped with the next step's kernels on a side stream.  The line then also
carries `compute_only` (the same loop without the exchange) and `bytes_gathered`.

Extra objects on the JSON line: roofline (dominant emulator kernel, HIP-event timed),
cpu_baseline (CPU oracle on this host, rank 0, N=1; + the recorded reference torch-CPU run),
frame_api / delivered_to_host (what a v2e.py caller of the drop-in class gets), slomo
(interpolated frames/s of the SuperSloMo HIP path with its own MFMA roofline), batched,
hd_noisy, end_to_end.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 260, 346
DT = 1.0 / 300.0
FRAMES_PER_STEP = 300
DEFAULT_KW = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01,
                  shot_noise_rate_hz=.001, refractory_period_s=.0005)
HBM_PEAK = 8.0e12        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_RATE = 256 * 4 * 2.4e9 / 2  # wave64 VALU instructions per second: a SIMD-32 takes one in 2 cycles (MI355X_MICROARCH.md)
SALU_RATE = 256 * 2.4e9          # one scalar unit per CU
PMC_FILES = ("r06_emulator_pmc_hbm.txt", "r05_emulator_pmc_hbm.txt")
TRACE_FILES = ("r06_emulator_chain_kernel_trace.txt",)
SQ_FILES = ("r06_emulator_sq.txt", "r05_emulator_sq.txt")
STAMP_STEPS = 20          # steps of the timed loop repeated with the chain's launches stamped on the device (roofline.frac)
F32_MFMA_PEAK = 157.3e12  # MI355X_MICROARCH.md: f32-input MFMA peak
INSTR_STEPS = 6           # steps re-run instrumented for the live per-launch kernel times of the roofline object
CLIP_STEPS = 24           # distinct seconds of synthetic video generated; longer runs cycle through them


def gen_frames_device(n, seed, device, h=H, w=W, sigma=3.0, i0=0):
    """SURVEY.md 8(d) config-2 pattern generated directly in HBM (synthetic data; torch is plumbing)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    y = torch.arange(h, device=device, dtype=torch.float32).view(1, h, 1)
    x = torch.arange(w, device=device, dtype=torch.float32).view(1, 1, w)
    out = torch.empty((n, h, w), dtype=torch.uint8, device=device)
    chunk = 100
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = torch.arange(i0 + s, i0 + e, device=device, dtype=torch.float32).view(-1, 1, 1)
        f = 127 + 100 * torch.sin((x + 3 * i) / 15.0) * torch.cos((y - 2 * i) / 20.0)
        f = f + sigma * torch.randn(f.shape, device=device, generator=g)
        out[s:e] = f.clamp_(0, 255).to(torch.uint8)
    return out


def emulator_bytes_per_pixel(kw):
    """SURVEY.md 8(d) general formula for the algorithmic HBM bytes per pixel per frame (u8 frames)."""
    s = 8 if kw["cutoff_hz"] > 0 else 4
    b = 1 + 2 * s + 8
    if kw["cutoff_hz"] > 0:
        b += 2 * s
    if kw["leak_rate_hz"] > 0:
        b += 4
    if kw["refractory_period_s"] > 0:
        b += 8
    return b


def pmc_traffic_per_launch(kernel):
    """HBM bytes per chain-kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r03_emulator_pmc_hbm.txt: FETCH_SIZE and WRITE_SIZE in separate runs).  Counters cannot
    be read from inside this process, so the value is the recorded one or null."""
    for name in PMC_FILES:
        try:
            for line in open(os.path.join(ROOT, "profiles", name)):
                if line.startswith("# " + kernel + "<") or line.startswith("# " + kernel + " "):
                    parts = line.split()
                    return int((float(parts[-2]) + float(parts[-1])) * 1024), "profiles/" + name
        except Exception:
            pass
    return None, None


def rocprof_kernel_us(kernel):
    """Average duration (us) of the chain kernel in the committed rocprofv3 kernel trace of this same command
    (profiles/r03_emulator_chain_kernel_trace.txt), or None: bench.py's own figure is measured live with HIP events."""
    for name in TRACE_FILES:
        try:
            for line in open(os.path.join(ROOT, "profiles", name)):
                parts = line.split()
                if len(parts) > 4 and parts[0].isdigit() and (kernel + "<") in line:
                    return float(parts[2]), "profiles/" + name
        except Exception:
            pass
    return None, None


def instruction_issue(ms_per_frame):
    """The resource that binds this pipeline (round-3 review): vector and scalar instructions issued per frame over ALL its
    kernels, from the committed SQ-counter pass of the headline workload (profiles/r06_emulator_sq.txt: SQ_INSTS_VALU / SQ_INSTS_SALU
    of k_ahead, k_chain, k_ctot, k_cframe1, k_cpull divided by the frames run), against the chip's issue rates
    (MI355X_MICROARCH.md: a wave64 float32 VALU instruction occupies its SIMD-32 for 2 cycles, a float64 one for 4 -> every
    kernel's VALU count is priced at 2 + 2 x its float64 share, the share being a static count over the kernel's disassembly
    ('# kernel_instr_per_frame <kernel> <VALU> <SALU> <f64 share>' lines; round-4 review: "prices every VALU instruction at 2
    cycles although a large share of k_chain's are f64"); one scalar unit per CU, one instruction per cycle).
    frac = the time the busier of the two pipes needs / the time a frame takes."""
    for name in SQ_FILES:
        try:
            valu = salu = None
            per_kernel = []
            for line in open(os.path.join(ROOT, "profiles", name)):
                if line.startswith("# headline_instr_per_frame"):
                    parts = line.split()
                    valu, salu = float(parts[2]), float(parts[3])
                elif line.startswith("# kernel_instr_per_frame"):
                    parts = line.split()
                    per_kernel.append((parts[2], float(parts[3]), float(parts[4]), float(parts[5])))
            if valu is None:
                continue
            simd_hz = 256 * 4 * 2.4e9
            if per_kernel:
                valu_cycles = sum(v * (2.0 + 2.0 * sh) for _, v, _, sh in per_kernel)
                f64_share = sum(v * sh for _, v, _, sh in per_kernel) / max(sum(v for _, v, _, _ in per_kernel), 1.0)
            else:  # an older profile without the per-kernel lines: every VALU instruction at 2 cycles (optimistic)
                valu_cycles, f64_share = valu * 2.0, None
            t_v, t_s = valu_cycles / simd_hz, salu / SALU_RATE
            return {"valu_per_frame": int(valu), "salu_per_frame": int(salu),
                    "per_64px_wave_frame": round((valu + salu) / (H * W / 64.0), 1),
                    "valu_f64_share_static": None if f64_share is None else round(f64_share, 3),
                    "per_kernel": [{"kernel": k, "valu": int(v), "salu": int(sa), "f64_share_static": sh} for k, v, sa, sh in per_kernel] or None,
                    "valu_cycles_per_frame": int(valu_cycles), "simd_cycles_per_s": simd_hz, "salu_rate_per_s": SALU_RATE,
                    "valu_us_per_frame": round(t_v * 1e6, 4), "salu_us_per_frame": round(t_s * 1e6, 4),
                    "bound_us_per_frame": round(max(t_v, t_s) * 1e6, 4), "measured_us_per_frame": round(ms_per_frame * 1e3, 4),
                    "frac": round(max(t_v, t_s) / (ms_per_frame * 1e-3), 4), "source": "profiles/" + name,
                    "note": "instructions of all kernels of the pipeline per frame (rocprofv3 SQ counters, recorded) priced at the chip's "
                            "issue costs (float32 VALU 2 cycles, float64 VALU 4 cycles by each kernel's static float64 share, scalar 1 "
                            "per CU and cycle) against this run's time per frame: the fraction of the BINDING resource; the rest is "
                            "dependency latency that one wave per SIMD cannot hide"}
        except Exception:
            pass
    return None


def recorded_reference():
    """The unmodified reference timed on the build container's CPU (scripts/cpu_reference_baseline.py): recorded,
    because /root/reference does not exist on the GPU box."""
    try:
        for name in ("r06_cpu_reference.json", "r02_cpu_reference.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                ref = json.load(open(path))
                ref["file"] = "profiles/" + name
                return ref
        return None
    except Exception:
        return None


def _oracle_run(frames_host, budget_s, out, idx):
    from oracle import oracle as orc
    o = orc.OracleEmulator(seed=1 + idx, rng_mode="philox", **DEFAULT_KW)
    o.generate_events(frames_host[0], 0.0)
    n_ev, n_fr = 0, 0
    t0 = time.perf_counter()
    for i in range(1, len(frames_host)):
        e = o.generate_events(frames_host[i], i * DT)
        n_ev += 0 if e is None else len(e)
        n_fr += 1
        if time.perf_counter() - t0 > budget_s:
            break
    out[idx] = (n_ev, n_fr, time.perf_counter() - t0)


def cpu_baseline(frames_host, budget_s=8.0):
    """CPU oracle (C restatement of the reference) on a bounded sample of the same clip, on THIS host: one thread (`value`), and
    every core (`all_cores`: one independent oracle instance per thread over the same frames -- clips are independent, this is
    how a CPU would be filled; the C calls and numpy release the GIL)."""
    import threading
    res = {}
    _oracle_run(frames_host, budget_s, res, 0)
    n_ev, n_fr, dt = res[0]
    out = {"value": round(n_ev / dt / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port", "same_host_reference": False,
           "frames_per_s": round(n_fr / dt, 1),
           "sample": "first %d frames of the same 346x260 clip, C oracle (oracle/emu_oracle.c), Philox RNG, %.1f s" % (n_fr, dt)}
    try:
        nthr = max(1, min(len(os.sched_getaffinity(0)), 64))
        res = {}
        th = [threading.Thread(target=_oracle_run, args=(frames_host, budget_s, res, i)) for i in range(nthr)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(sum(v[0] for v in res.values()) / wall / 1e6, 3), "unit": "Mevents/s", "cores": nthr,
                            "kind": "port", "frames_per_s": round(sum(v[1] for v in res.values()) / wall, 1),
                            "sample": "%d independent oracle instances (one per thread, of %d hardware threads) over the first %d "
                                      "frames of the same clip, %.1f s" % (nthr, os.cpu_count() or 0, max(v[1] for v in res.values()), wall)}
    except Exception as e:
        out["all_cores"] = {"error": repr(e)[:200]}
    ref = recorded_reference()
    if ref:
        runs = ref["emulator"]["runs"]
        out["same_host_reference_note"] = ("no same-host reference baseline exists: the reference tree cannot travel to the GPU box; `value` "
                                            "is the C port on THIS host, `reference` the unmodified reference on the build container")
        out["reference"] = {"kind": "reference, recorded on a DIFFERENT host (the build container: the reference tree does not exist on the GPU box)",
                            "host": ref["host"], "script": "scripts/cpu_reference_baseline.py -> " + ref["file"],
                            "runs": [{"cores": r["threads"], "value": r["Mevents_per_s"], "unit": "Mevents/s",
                                      "frames_per_s": r["frames_per_s"]} for r in runs]}
    return out


def frame_api_bench(frames_all, budget_frames=600):
    """What a v2e.py caller of the drop-in gets: generate_events(frame, t) one frame at a time, host numpy in, host
    numpy out (PCIe inclusive), in the default tape mode (the reference's own MT19937 stream) and in Philox mode."""
    from v2e_amd import EventEmulator
    host = frames_all[:budget_frames + 1].cpu().numpy()
    out = {}
    for mode in ("tape", "philox"):
        emu = EventEmulator(device="cuda", seed=1, rng_mode=mode, **DEFAULT_KW)
        emu.generate_events(host[0], 0.0)
        for i in range(1, 21):
            emu.generate_events(host[i], i * DT)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ev = 0
        for i in range(21, len(host)):
            e = emu.generate_events(host[i], i * DT)
            n_ev += 0 if e is None else len(e)
        sec = time.perf_counter() - t0
        out[mode] = {"frames_per_s": round((len(host) - 21) / sec, 1), "Mevents_per_s": round(n_ev / sec / 1e6, 2)}
    out["note"] = "EventEmulator.generate_events per frame, host ndarray in / host ndarray out (PCIe inclusive), 346x260"
    return out


def delivered_to_host_bench(device, frames_all, steps=10):
    """The device-resident run with its event rows delivered to pinned host memory: the D2H copy of step s on a side
    stream overlaps step s + 1 (two event buffers alternate)."""
    from v2e_amd import EventEmulator
    F = FRAMES_PER_STEP
    emu = EventEmulator(device=device, seed=1, rng_mode="philox", **DEFAULT_KW)
    emu.generate_events(frames_all[0], 0.0)
    buf = torch.empty((F, H, W), dtype=torch.uint8, device=device)
    copy_stream = torch.cuda.Stream(device)
    host = [None, None]

    def enqueue(s):
        lo = 1 + (s % max((int(frames_all.shape[0]) - 1) // F, 1)) * F   # the steps of video that were generated
        buf.copy_(frames_all[lo:lo + F])
        return emu.generate_events_batch_async(buf, [(1 + s * F + i) * DT for i in range(F)], return_device=True)

    def deliver(pend, slot):
        ev, counts = pend.result()
        n = int(counts.sum())
        if host[slot] is None or host[slot].shape[0] < n:
            host[slot] = torch.empty((int(n * 1.2) + 1024, 4), dtype=torch.float32).pin_memory()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(pend.done)
            host[slot][:n].copy_(ev, non_blocking=True)
        return n

    pend = enqueue(0)
    n0 = deliver(pend, 0)
    # (both pinned buffers exist before the clock starts: page-locking 200 MB takes tens of milliseconds, a multiple of a step -- the
    #  second one used to be allocated inside the timed loop and the figure swung between 0.6 and 2.0 Gev/s with it)
    host[1] = torch.empty((int(n0 * 1.2) + 1024, 4), dtype=torch.float32).pin_memory()
    copy_stream.synchronize()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n_ev = 0
    pend = enqueue(1)
    for s in range(2, 2 + steps):
        nxt = enqueue(s)
        n_ev += deliver(pend, s & 1)
        pend = nxt
    n_ev += deliver(pend, (2 + steps) & 1)
    copy_stream.synchronize()
    torch.cuda.synchronize(device)
    sec = time.perf_counter() - t0
    return {"value": round(n_ev / sec / 1e6, 1), "unit": "Mevents/s", "GB_per_s_over_pcie": round(n_ev * 16 / sec / 1e9, 2),
            "note": "same device-resident run, event rows (16 B each) copied to pinned host memory on a side stream while the "
                    "next step executes"}


def self_allgather_bench(device, frames_all, steps=30, warmup=3):
    """N = 1 self-test of the exchange path (round-3 review): the headline loop with the event-stream all-gather switched on
    over a one-rank RCCL group -- the pack32 kernels, the host-side count exchange and the ncclAllGather enqueue are
    executed and timed on the driver's box, beside the same loop without them.  Says nothing about links (there is no
    peer); it pins that the path runs and what it costs the producing GPU."""
    import torch.distributed as dist
    from v2e_amd import EventEmulator
    from v2e_amd.benchutil import run_steps
    from v2e_amd.dist import EventStreamGatherer
    own_pg = not dist.is_initialized()
    # RCCL and gloo print banners through C stdio on fd 1: the JSON line must stay the only thing on stdout
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        res = {}
        for tag, wire in (("compute_only", None), ("pack32", "pack32"), ("pack64", "pack64")):
            emu = EventEmulator(device=device, seed=1, rng_mode="philox", **DEFAULT_KW)
            emu.generate_events(frames_all[0], 0.0)
            g = EventStreamGatherer(device, 1, wire=wire, algo="allgather", sensor=(H, W)) if wire else None
            el, ne = run_steps(emu, frames_all, FRAMES_PER_STEP, DT, steps, warmup, g, dist, device)
            res[tag] = {"value": round(ne / el / 1e6, 1), "unit": "Mevents/s", "ms_per_step": round(el / steps * 1e3, 4)}
            if g is not None:
                res[tag]["bytes_gathered_per_step"] = int(g.bytes_gathered / (steps + warmup))
                parts = g.result()
                res[tag]["last_step_rows_back"] = int(parts[0].shape[0])
        res["note"] = ("world size 1 over RCCL: pack kernels + host count exchange + ncclAllGather enqueue on a side stream, "
                       "overlapped with the next step; no link is exercised")
        return res
    finally:
        if own_pg:
            dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved_fd, 1)
        os.close(saved_fd)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """Bare `python bench.py --gpus N` (no WORLD_SIZE in the environment): start one rank per GPU under
    torch.distributed.run on this node and relay rank 0's JSON line as the only line on stdout; the exit code is the
    launcher's.  (The driver's documented N > 1 command already runs under torch.distributed.run and never comes here.)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in p.stdout.decode("utf-8", "replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)  # banners of the ranks: never on stdout
    if line is None:
        raise SystemExit(p.returncode or 1)
    print(line, flush=True)
    raise SystemExit(p.returncode)


STUB = os.environ.get("V2E_AMD_BENCH_STUB") == "1"
"""V2E_AMD_BENCH_STUB=1: the launcher / process-group / step-loop / reduction / JSON plumbing of this file on CPU tensors with
gloo and a stub engine (tests/bench_stub.py) -- no kernel runs and the line says so (`data: "stub"`, metric prefixed STUB):
it exists so that the N > 1 start-up path is exercised every round in a container without GPUs (tests/test_bench_launch.py)."""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=9, help="timed regions of --steps steps each; the median one is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the slomo / batched / frame-API side measurements")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--force-allgather", action="store_true", help="run the RCCL gather path even with one rank (self-test)")
    ap.add_argument("--gather-algo", choices=["allgather", "p2p"], default="allgather",
                    help="event-stream exchange: ncclAllGather on padded payloads, or grouped point-to-point sends of exact sizes")
    ap.add_argument("--gather-wire", choices=["auto", "pack32", "pack64"], default="auto")
    ap.add_argument("--no-roofline-rerun", action="store_true",
                    help="skip the instrumented re-run behind the timed blocks (profiling: the trace then holds the timed configuration's launches only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    if STUB:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    ranks_seen = None
    if world > 1 or args.force_allgather:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)  # the ranks that took part in a collective of this job (RCCL on the GPU box), not the number asked for
        ranks_seen = int(one.item())

    from v2e_amd.benchutil import run_steps
    from v2e_amd.dist import EventStreamGatherer
    if STUB:
        from tests.bench_stub import StubEmulator
    else:
        from v2e_amd import EventEmulator

    K, Wm = args.steps, args.warmup
    F = FRAMES_PER_STEP if not STUB else 4
    H, W = (260, 346) if not STUB else (4, 6)
    # one independent clip per rank (configs[4]: seeds 10..17); rank 0 at N=1 uses seed 1 (configs[1])
    clip_seed = 1 if world == 1 else 10 + rank
    frames_all = gen_frames_device(min(K + Wm, CLIP_STEPS) * F + 1, clip_seed, device, h=H, w=W)

    def make_emu():
        if STUB:
            return StubEmulator(rank)
        emu = EventEmulator(device=device, seed=clip_seed, rng_mode="philox", **DEFAULT_KW)
        emu.generate_events(frames_all[0], 0.0)  # first frame: state init, no events
        return emu

    use_gather = (world > 1 or args.force_allgather) and not args.no_allgather
    nblocks = max(1, args.blocks)

    def timed_blocks(emu, gather):
        """`nblocks` timed regions of K steps each on one emulator (the clip keeps running); every region is bracketed by
        barrier + synchronize inside run_steps; (seconds, events) per region, seconds = max over ranks."""
        res = []
        for b in range(nblocks):
            el, ne = run_steps(emu, frames_all, F, DT, K, Wm if b == 0 else 0, gather, dist, device, first_step=b * K + (Wm if b else 0))
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t[0].item())
                n = torch.tensor([ne], dtype=torch.float64, device=device)
                dist.all_reduce(n, op=dist.ReduceOp.SUM)
                ne = int(n[0].item())
            res.append((el, ne))
        return res

    def median_block(res):
        order = sorted(range(len(res)), key=lambda i: res[i][0] / max(res[i][1], 1))
        return res[order[len(order) // 2]]

    emu = make_emu()
    gather = None
    if use_gather:
        gather = (EventStreamGatherer("cpu", world) if STUB else
                  EventStreamGatherer(device, world, wire=args.gather_wire, algo=args.gather_algo, sensor=(H, W)))
    blocks = timed_blocks(emu, gather)
    elapsed, tot_events = median_block(blocks)
    compute_only = None
    if use_gather:  # the same loop without the exchange: how much of the N-GPU number the interconnect decides
        emu_c = make_emu()
        compute_only = median_block(timed_blocks(emu_c, None))

    out = None
    if rank == 0:
        bpp = emulator_bytes_per_pixel(DEFAULT_KW)
        out = {
            "metric": ("STUB (no kernel ran) " if STUB else "") + "Mevents/s (EventEmulator.generate_events, 346x260, 10x slowdown)",
            "value": round(tot_events / elapsed / 1e6, 3),
            "unit": "Mevents/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "timed_blocks": {"blocks": nblocks, "steps_each": K, "reported": "median block",
                             "Mevents_per_s": [round(ne / el / 1e6, 1) for el, ne in blocks],
                             "spread_rel": round((max(ne / el for el, ne in blocks) - min(ne / el for el, ne in blocks)) /
                                                 (tot_events / elapsed), 4)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "stub" if STUB else "synthetic",
            "ranks_seen": ranks_seen if ranks_seen is not None else 1,
            "collective_backend": None if dist is None else ("gloo (stub)" if STUB else "nccl (RCCL)"),
            "config": {"workload": "BASELINE configs[1]: 346x260 random-gradient video (SURVEY 8(d) config 2), "
                                   "dt=1/300 s, emulator-only, v2e CLI default DVS params, Philox RNG, "
                                   "%d frames/step device-resident, one clip per GPU" % F,
                       "frames_per_step": F, "clips_per_gpu": 1,
                       "event_stream_allgather": bool(use_gather)},
            "emulator_frames_per_s": round(world * K * F / elapsed, 1),
            "events_per_frame": round(tot_events / (world * K * F), 1),
        }
        if compute_only:
            out["compute_only"] = {"value": round(compute_only[1] / compute_only[0] / 1e6, 3), "unit": "Mevents/s",
                                   "ms_per_step": round(compute_only[0] / K * 1e3, 4),
                                   "note": "same loop, no event-stream exchange"}
            out["with_allgather"] = {"value": out["value"], "unit": "Mevents/s",
                                     "bytes_gathered_per_rank_per_step": int(gather.bytes_gathered / max(nblocks * K + Wm, 1)),
                                     "algo": args.gather_algo,
                                     "wire_format": ("4 B per event + 8 B per block of one time stamp (v2e_events_pack32)"
                                                     if gather.wire == "pack32" else "8 B per event (v2e_events_pack64)")}

    # ---------------- roofline of the dominant emulator kernel (rank 0, HIP events, same workload)
    if rank == 0 and not STUB and args.no_roofline_rerun:
        whole = emulator_bytes_per_pixel(DEFAULT_KW) * H * W + 16 * tot_events / (world * K * F)
        out["roofline"] = {"bound": "hbm", "kernel": "k_chain", "achieved": None, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": None, "traffic": None,
                           "whole_step": {"algorithmic_bytes_per_frame": int(whole), "achieved_GBps": round(whole * K * F / elapsed / 1e9, 2),
                                          "frac": round(whole * K * F / elapsed / HBM_PEAK, 5)},
                           "note": "--no-roofline-rerun: no per-launch measurement in this run"}
    if rank == 0 and not STUB and not args.no_roofline_rerun:
        eng = emu._engine
        P = emu._params()
        buf = torch.empty((F, H, W), dtype=torch.uint8, device=device)
        # ---- the kernel AS IT RUNS IN THE TIMED CONFIGURATION: the timed loop once more (same run_steps, same pipelined runs, k_ahead and
        # the emission on their streams beside the chain), every chain launch leaving the device wall-clock time of its first workgroup's
        # start and its last workgroup's end (v2e_emu_launch_stamps: two atomics per workgroup, nothing else changes)
        eng.launch_stamps(STAMP_STEPS)
        el_st, ne_st = run_steps(emu, frames_all, F, DT, STAMP_STEPS, 0, None, None, device, first_step=Wm + nblocks * K)
        st = eng.launch_stamps(0, read=STAMP_STEPS).astype(np.int64)
        nl_st = int((st[0, :, 1] > 0).sum())
        st_us = st[:, :nl_st] / 1e3
        st_dur = st_us[:, :, 1] - st_us[:, :, 0]
        st_gap = st_us[:, 1:, 0] - st_us[:, :-1, 1]
        st_bound = st_us[1:, 0, 0] - st_us[:-1, -1, 1]
        # re-run the frames of INSTR_STEPS steps instrumented (state keeps advancing; timing only).  Several steps, because the
        # launches that first redo their predecessor are few and uneven (one step of the clip holds a 245 us launch, others
        # none above 70): the average over one step says little
        ev = eng.event_buffer(1)
        recs = eng.alloc_recs(F)
        prof, per_launch_all, evs = None, [], []
        for si in range(INSTR_STEPS):
            lo = 1 + ((Wm + K - 1 + si) % max((int(frames_all.shape[0]) - 1) // F, 1)) * F
            buf.copy_(frames_all[lo:lo + F])
            t0s = emu.t_previous + si * F * DT
            t_prev = [t0s + i * DT for i in range(F)]
            t_frame = [t0s + (i + 1) * DT for i in range(F)]
            eng.run(P, buf, t_prev, t_frame, emu.frame_counter + si * F, ev, recs, use_graph=2)
            pr = eng.last_profile()
            per_launch_all += list(pr.get("chain_launch_us", []))
            evs.append(float(eng.recs_to_numpy(recs)[:, 0]["n_events"].mean()))
            if prof is None:
                prof = dict(pr)
            else:
                for key in ("rank", "emit"):
                    prof[key] += pr[key]
                for key in ("step_launches", "emit_batches"):
                    if key in pr:
                        prof[key] = prof.get(key, 0) + pr[key]
        prof["chain_launch_us"] = per_launch_all
        kname, fpl, fpb = eng.last_pipeline()
        ev_per_frame = float(np.mean(evs))
        npx = H * W
        # The dependency chain owns the per-pixel state traffic: SURVEY 8(d) prices a frame at 53 B/pixel (frame 1 + lp 16 +
        # base 16 + thresholds 8 + noise rate 4 + ts_mem 8) + 16 B/event; the chain kernel also writes the 4-byte count word the
        # event writer reads.  A launch covers `fpl` frames.
        n_step = max(prof.get("step_launches", 0), 1)   # chain launches of the instrumented steps
        n_per_step = n_step / INSTR_STEPS
        period_us = elapsed / K / n_per_step * 1e6  # the driver-timed region: one step = n_per_step chain launches
        kernel_us = prof["rank"] / n_step * 1e3    # HIP events before and after every chain launch, on its stream: the kernels alone
        # 53 B per pixel and frame x the frames a launch advances ON AVERAGE: a step's launches are full ones (fpl
        # frames), one partial one and the tail launch that only validates, and all of them are in the average duration
        step_bytes = bpp * npx * F / n_per_step
        full_bytes = bpp * npx * fpl
        emit_bytes = 16 * ev_per_frame + 4 * npx
        ach = step_bytes / (kernel_us * 1e-6)
        per_launch = prof.get("chain_launch_us", [])
        nps = int(round(n_per_step))
        full = sorted(u for i, u in enumerate(per_launch) if i % nps < F // fpl)  # the launches that advance fpl frames (the partial and the tail one excluded)
        p50 = full[len(full) // 2] if full else None
        whole = (bpp * npx + 16 * ev_per_frame)
        traffic, prof_file = pmc_traffic_per_launch(kname.split("(")[0])
        rp_us, rp_file = rocprof_kernel_us(kname.split("(")[0])
        live = {"avg_kernel_us": round(kernel_us, 3), "achieved_GBps": round(ach / 1e9, 2), "frac": round(ach / HBM_PEAK, 5),
                "launches_timed": n_step, "launch_us": [round(u, 1) for u in per_launch],
                "note": "HIP events before and after every chain launch of an instrumented re-run of %d steps' frames with ALL kernels "
                        "of the run on ONE stream: each kernel running alone (no contention with k_ahead / the emission kernels)" % INSTR_STEPS}
        st_full = np.sort(st_dur[:, :F // fpl].ravel())   # the launches that advance fpl frames
        st_mean = float(st_dur.mean())
        timed = {
            "avg_kernel_us": round(st_mean, 3), "achieved_GBps": round(step_bytes / (st_mean * 1e-6) / 1e9, 2),
            "frac": round(step_bytes / (st_mean * 1e-6) / HBM_PEAK, 5),
            "launches_timed": int(st_dur.size), "launches_per_step": nl_st, "steps": int(st_dur.shape[0]),
            "Mevents_per_s_of_the_stamped_steps": round(ne_st / el_st / 1e6, 1),
            "full_launch_us": {"p10": round(float(st_full[len(st_full) // 10]), 2), "p50": round(float(st_full[len(st_full) // 2]), 2),
                               "p90": round(float(st_full[len(st_full) * 9 // 10]), 2)},
            "per_step_us": {"sum_of_launch_durations": round(float(st_dur.sum(1).mean()), 1),
                            "gaps_between_launches": round(float(st_gap.sum(1).mean()), 1),
                            "last_launch_end_to_next_steps_first_start": round(float(st_bound.mean()), 1)},
            "note": "LIVE, this process, this binary: %d more steps of the timed loop (pipelined runs: k_ahead and the emission kernels on "
                    "their streams beside the chain) with every chain launch stamped on the device (first workgroup's start to last "
                    "workgroup's end, s_memrealtime); the mean covers every launch of a step -- the full ones, the partial one, the tail "
                    "launch that only validates, redo passes included" % STAMP_STEPS}
        recorded = None if rp_us is None else {
            "avg_kernel_us": rp_us, "source": rp_file, "frac": round(step_bytes / (rp_us * 1e-6) / HBM_PEAK, 5),
            "note": "the kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of `bench.py --no-roofline-rerun` "
                    "(the timed configuration's launches only): the cross-check of `timed_configuration` (the profiler's duration runs from "
                    "dispatch to completion, the stamps from first workgroup to last)"}
        head = timed
        out["roofline"] = {
            "bound": "hbm", "kernel": kname.split("(")[0],
            "achieved": head["achieved_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": head["frac"],
            "frac_is": "algorithmic bytes per launch / the kernel's average launch duration in the TIMED configuration, measured live on the "
                       "device in this process (timed_configuration); alone_hip_events / median_full_launch: every kernel alone; "
                       "as_delivered / whole_step: against the driver-timed region",
            "traffic": traffic,
            "traffic_source": str(prof_file) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; counters cannot be read "
                                          "from inside the process)",
            "traffic_over_algorithmic": None if not traffic else round(traffic / step_bytes, 3),
            "algorithmic_bytes_per_launch": int(step_bytes), "frames_per_launch": fpl,
            "frames_per_launch_avg": round(F / n_per_step, 2),
            "timed_configuration": timed,
            "rocprof_recorded": recorded,
            "alone_hip_events": live,
            "median_full_launch": None if p50 is None else {
                "us": round(p50, 3), "bytes": int(full_bytes), "achieved_GBps": round(full_bytes / (p50 * 1e-6) / 1e9, 2),
                "frac": round(full_bytes / (p50 * 1e-6) / HBM_PEAK, 5),
                "note": "the median of the launches that advance %d frames, each alone: a launch WITHOUT a redo pass" % fpl},
            "launch_period_us": round(period_us, 3),
            "as_delivered": {"achieved_GBps": round(step_bytes / (period_us * 1e-6) / 1e9, 2),
                             "frac": round(step_bytes / (period_us * 1e-6) / HBM_PEAK, 5),
                             "note": "the same bytes against the timed region's time per chain launch (ms_per_step / launches per step: "
                                     "everything the step does -- k_ahead, the emission kernels, gaps, redo passes -- included)"},
            "instruction_issue": instruction_issue(elapsed / (K * F) * 1e3),
            "event_writer_us_per_batch": round(prof["emit"] / max(prof.get("emit_batches", 1), 1) * 1e3, 3),
            "emission": {"frames_per_batch": fpb, "algorithmic_bytes_per_frame": int(emit_bytes)},
            "whole_step": {"algorithmic_bytes_per_frame": int(whole),
                           "achieved_GBps": round(whole * K * F / elapsed / 1e9, 2),
                           "frac": round(whole * K * F / elapsed / HBM_PEAK, 5)},
            "note": "algorithmic bytes = 53 B x pixels x the step's frames / the step's chain launches (what a launch advances on average; "
                    "SURVEY.md 8(d)).  Three durations divide them: timed_configuration (what `frac` is: live device stamps of the timed "
                    "loop), alone_hip_events (live HIP events, each kernel alone), as_delivered (the driver-timed region per launch).  The "
                    "per-pixel state (2.9 MB at 346x260) stays in registers for a launch's 32 frames and the PMC traffic is below the "
                    "algorithmic bytes: HBM is not what binds this kernel -- instruction_issue states the fraction of the resource that "
                    "does, and whole_step prices the complete frame (state + event rows) against the driver-timed region",
        }
        if not args.no_extras and world == 1:
            try:
                from v2e_amd.benchutil import batched_emulator_bench, e2e_bench, hd_noisy_emulator_bench, slomo_bench

                def leg(fn, *a, **k):
                    # what the leg before left unreachable (an emulator and its GB of device scratch) is released HERE, not by a
                    # garbage collection in the middle of this leg's timed loop (freeing device memory synchronises the device)
                    import gc
                    gc.collect()
                    torch.cuda.synchronize(device)
                    return fn(*a, **k)

                out["frame_api"] = leg(frame_api_bench, frames_all)
                out["delivered_to_host"] = leg(delivered_to_host_bench, device, frames_all)
                out["batched"] = leg(batched_emulator_bench, device)
                out["hd_noisy"] = leg(hd_noisy_emulator_bench, device)
                out["slomo"] = leg(slomo_bench, device)
                out["slomo_f32"] = leg(slomo_bench, device, conv_math="f32")
                out["slomo_bf16x3"] = leg(slomo_bench, device, conv_math="bf16x3")  # the exact three-piece split (what the range guard falls back to)
                ref = recorded_reference()
                if ref:
                    out["slomo"]["cpu_baseline"] = {
                        "kind": "reference, recorded on a DIFFERENT host (scripts/cpu_reference_baseline.py, build container)", "host": ref["host"]["cpu"],
                        "same_host_reference": False,
                        "runs": [{"cores": q["threads"], "batch_pairs": q["batch_pairs"], "value": q["interpolated_frames_per_s"],
                                  "unit": "frames/s"} for q in ref["slomo"]["runs"]]}
                out["end_to_end"] = leg(e2e_bench, device)
            except Exception as e:  # side measurements must never hide the headline number
                out["extras_error"] = repr(e)[:300]
            try:  # SURVEY 8(a)'s other size (round-3 review): one pair of a 1280x720 source, U = 2, as the 1280x704 parity fixture runs it
                out["slomo_hd"] = slomo_bench(device, B=1, U=2, H=704, W=1280, iters=3)
            except Exception as e:
                out["slomo_hd"] = {"error": repr(e)[:300]}
            if dist is None:
                try:
                    out["self_allgather"] = self_allgather_bench(device, frames_all)
                except Exception as e:
                    out["self_allgather"] = {"error": repr(e)[:300]}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        # (behind the GPU legs: its 64 host threads leave the main thread on whatever NUMA node the scheduler likes, and pinned host
        #  buffers allocated from there cost the PCIe legs half their rate -- delivered_to_host 1.0 instead of 3.1 Gev/s, round 6)
        out["cpu_baseline"] = cpu_baseline(frames_all[:1501].cpu().numpy())
    if dist is not None and not args.no_extras:
        # the SuperSloMo stage of ONE clip sharded over the ranks by source pairs (every rank takes part; rank 0 reports); stub mode:
        # the same collective sequence over gloo with a stand-in for the interpolator, a small clip
        try:
            from v2e_amd.benchutil import slomo_sharded_bench
            if STUB:
                from tests.bench_stub import StubUpsampler
                r = slomo_sharded_bench(device, dist, n_src=11, U=3, H=6, W=8, reps=2, pipe=StubUpsampler(3))
            else:
                r = slomo_sharded_bench(device, dist)
            if rank == 0:
                out["slomo_sharded"] = r
        except Exception as e:
            if rank == 0:
                out["slomo_sharded"] = {"error": repr(e)[:300]}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its banner through C stdio; drain it so the JSON line is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stderr.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
