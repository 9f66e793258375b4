#!/usr/bin/env python
"""Assemble the committed round-6 profile artefacts (profiles/r06_*.txt) from what scripts/gpu_r06_profiles.sh left under
gpurun_out/ (rocprofv3 summaries made on the MI355X box).  bench.py / benchutil.py read the machine-readable lines:
  r06_emulator_pmc_hbm.txt   '# k_chain<...>  <launches> <FETCH KiB> <WRITE KiB>'
  r06_emulator_sq.txt        '# headline_instr_per_frame <VALU> <SALU> <frames>'
  r06_slomo_counters.txt     '# unet_forward_bytes <conv_math> <fetch bytes> <write bytes>'"""
import os
import re

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
EMU_KERNELS = ("k_ahead", "k_chain", "k_ctot", "k_cframe1", "k_cframe", "k_cpull", "k_cemit", "k_coff", "k_zero_words")


def rd(name):
    return open(os.path.join(G, name)).read()


def wr(name, text):
    open(os.path.join(R, "profiles", name), "w").write(text)
    print("profiles/%s  %d lines" % (name, text.count("\n")))


def pmc_avg(txt, kernel_prefix):
    for line in txt.splitlines():
        if kernel_prefix in line:
            m = re.search(r"launches\s+(\d+)\s+avg\s+([0-9.]+)", line)
            return int(m.group(1)), float(m.group(2))
    return None, None


def sq_totals(txt):
    """(VALU, SALU) summed over the emulator kernels of an SQ summary table: calls x waves x per-wave count."""
    v = s = 0.0
    rows = []
    for line in txt.splitlines():
        if line.startswith("#") or not line.strip():
            continue
        parts = line.split()
        # kernel name may contain spaces ("k_chain<double, unsigned char, false>"): the numeric tail has 13 fields
        tail = parts[-13:]
        name = " ".join(parts[:-13])
        if not any(name.startswith(k) for k in EMU_KERNELS):
            continue
        calls, waves, valu_w, salu_w = float(tail[5]), float(tail[7]), float(tail[8]), float(tail[9])
        v += calls * waves * valu_w
        s += calls * waves * salu_w
        rows.append((name, calls, waves, valu_w, salu_w))
    return v, s, rows


def static_f64_share():
    """Share of float64 instructions among the VALU instructions of each emulator kernel, from the disassembly of the committed source
    (static counts: the frame loops dominate the dynamic mix, so this is an estimate, labelled as such where it is used)."""
    import subprocess, tempfile, collections
    out = {}
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "emu.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-S", "--cuda-device-only", "-Wno-unused-result", "-Wno-unused-value", "-o", asm,
                        os.path.join(R, "v2e_amd", "csrc", "emu.hip")], check=True, stderr=subprocess.DEVNULL)
        cur, st = None, collections.OrderedDict()
        for ln in open(asm):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                cur = m.group(1)
                st[cur] = [0, 0]
                continue
            if cur and ln.startswith("\t") and not ln.startswith("\t."):
                op = ln.strip().split()[0]
                if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                    st[cur][0] += 1
                    st[cur][1] += "_f64" in op
    for key, tag in (("k_chainIdhLb0ELb1E", "k_chain"), ("7k_aheadIh", "k_ahead"), ("6k_ctotE", "k_ctot"), ("7k_cpullILb0E", "k_cpull"), ("7k_cemitE", "k_cemit"), ("9k_cframe1E", "k_cframe1")):
        for k, (v, f) in st.items():
            if key in k:
                out[tag] = f / max(v, 1)
    return out


def per_kernel_lines(rows, frames):
    """'# kernel_instr_per_frame <kernel> <VALU> <SALU> <static float64 share of its VALU instructions>' (read by bench.py)."""
    share = static_f64_share()
    agg = {}
    for name, calls, waves, valu_w, salu_w in rows:
        k = name.split("<")[0].strip()
        a = agg.setdefault(k, [0.0, 0.0])
        a[0] += calls * waves * valu_w / frames
        a[1] += calls * waves * salu_w / frames
    out = ["# per kernel and frame, with the share of float64 among the kernel's VALU instructions (static count over its disassembly; a",
           "# float64 VALU instruction occupies the SIMD for 4 cycles where a float32 one takes 2: bench.py prices them so):"]
    for k, (v, sa) in agg.items():
        out.append("# kernel_instr_per_frame %s %.0f %.0f %.3f" % (k, v, sa, share.get(k, 0.0)))
    return "\n".join(out) + "\n#\n"


def emulator():
    kt, tl = rd("p6_kt.txt"), rd("p6_kt_timeline.txt")
    window = rd("p6_kt_step.txt")
    wr("r06_emulator_chain_kernel_trace.txt", """# rocprofv3 kernel trace of the headline workload, round 6: PIPELINED runs (plain launches on four streams: the chain | k_ahead |
# the emission tables k_ctot + k_cframe1 | the event rows k_cpull; no graph, no join at a run's end: DESIGN.md section 3)
# command (on the MI355X box, cd /tmp; TMPDIR=/tmp):
#   rocprofv3 --kernel-trace --stats -d out -- python bench.py --steps 12 --warmup 3 --blocks 1 --no-extras --no-cpu-baseline --no-roofline-rerun
# (--no-roofline-rerun: no instrumented re-run behind the timed steps, so EVERY k_chain launch below is one of the timed configuration,
#  under contention with the other streams; the round-5 review found 66 alone-run launches blended into that round's 121.)
# summarised by profiles/summarize_rocprof_db.py (top kernels) and scripts/kernel_timeline.py (k_chain launch timeline).
# workload: BASELINE configs[1], 346x260, one clip, 300 frames per step, CLI-default DVS parameters, Philox.
# one k_chain launch = 32 frames (10 per step: nine full ones, one of 12 frames, + a tail launch validating the last speculation; a
# redo pass runs inside the launch that finds the miss); k_ahead / k_ctot / k_cframe1 / k_cpull: one launch per 64 frames.
# UNDER THE PROFILER the host's ~100 API calls per run take several times as long and the loop becomes host-bound (the steps of this
# trace take ~2x the unprofiled 0.69 ms): the kernels' DURATIONS are what this file is for -- bench.py's live figure
# (roofline.timed_configuration: device time stamps of the same launches in the unprofiled loop) is the one `frac` uses, and this
# trace's k_chain average is its cross-check (roofline.rocprof_recorded).
#
""" + kt + "\n# k_chain launch timeline (same trace)\n" + tl + "\n# every kernel of the trace's last steps: start (us), duration (us), hardware queue, stream (scripts/dump_timeline.py)\n" + window)
    f, w = rd("p6_FETCH_SIZE.txt"), rd("p6_WRITE_SIZE.txt")
    rows = []
    for k in ("k_chain<double, unsigned char, false", "k_ahead<unsigned char>", "k_cpull<false>", "k_cemit", "k_ctot", "k_cframe1"):
        n, fa = pmc_avg(f, k)
        n2, wa = pmc_avg(w, k)
        if n is None and n2 is None:
            continue
        rows.append("# %-38s %4d  %9.1f  %9.1f" % (k + (", true>" if k.startswith("k_chain") else ""), n or n2, fa or 0.0, wa or 0.0))
    wr("r06_emulator_pmc_hbm.txt", """# HBM traffic of the emulator kernels, round 6
# commands (separate passes, as the MI355X guide prescribes; summary by profiles/summarize_rocprof_pmc.py):
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py ...
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py ...
# (the same command as the kernel trace: ... bench.py --steps 12 --warmup 3 --blocks 1 --no-extras --no-cpu-baseline --no-roofline-rerun)
# workload: 346x260, one clip, 300 frames/step, CLI-default DVS parameters; a k_chain launch covers 32 frames, the others 64.
# units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB (summed over XCDs); no x2 correction applied (the guide calibrates it for
# wide coalesced streams; these kernels load 1-16 B per lane).  A kernel missing from one pass's top list shows 0.0.
#
# kernel                                launches   FETCH_SIZE avg [KiB]   WRITE_SIZE avg [KiB]
""" + "\n".join(rows) + """
#
# algorithmic, per launch: k_chain (32 frames) 53 B x 89 960 px x 32 = 153 MB priced / what it touches: records 46 MB + state
# once + count words 11.5 MB; k_cpull (64 frames) 64 x ~35 700 events x 16 B = 36.5 MB of rows, written in row order (whole lines: the
# push writer k_cemit of rounds 2-4 wrote 55.5 MB for them); k_ctot additionally writes the pull's pixel ballots (32 B per group and key).
#
# raw summaries:
""" + f + w)
    sqh, sqa = rd("p6_sqh.txt"), rd("p6_sq.txt")
    v, s, rows = sq_totals(sqh)
    m = re.search(r"headline: (\d+) events in (\d+) frames", rd("p6_sqh_frames.txt"))
    frames = int(m.group(2)) if m else 1200
    per_wave = (v + s) / frames / (346 * 260 / 64.0)
    wr("r06_emulator_sq.txt", """# SQ counters of the emulator kernels, round 6
# command: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
#          --kernel-trace -- python scripts/emu_workloads.py headline        (first table: the headline clip, the timed loop of bench.py: pipelined runs, %d frames)
#          ... -- python scripts/emu_workloads.py batched hd                  (second table: 64 clips x 346x260; 1280x720 noisy)
# summary: profiles/summarize_rocprof_sq.py; *_w = per wave and launch.
# Instructions per frame of the headline pipeline = sum over its kernels of calls x waves x (VALU_w, SALU_w) / frames:
# headline_instr_per_frame %.0f %.0f %d
#   = %.0f VALU + %.0f SALU = %.0f instructions per 64-pixel wave and frame over the WHOLE pipeline (round 5: 700, round 4: 784; the round-5 review's target: <= 600)
#
""" % (frames, v / frames, s / frames, frames, v / frames / (346 * 260 / 64.0), s / frames / (346 * 260 / 64.0), per_wave) +
       "\n".join("#   %-44s calls %4d waves %8d  VALU_w %8.1f SALU_w %8.1f" % r for r in rows) + "\n#\n" + per_kernel_lines(rows, frames) + sqh +
       "\n# ---- batched (64 clips) and 1280x720 noisy\n" + sqa)


def hd():
    wr("r06_emulator_hd_kernel_trace.txt", """# rocprofv3 kernel trace of the 1280x720 `noisy` workload (BASELINE configs[3]; bench.py's hd_noisy leg), round 6
# (pipelined runs, as the headline loop: 13.1-13.65 Gev/s against 11.5-12.4 with one hipGraph per run -- experiment 30;
#  pull event writer k_cpull<true>, emission batches of 64 frames)
# command (on the MI355X box, cd /tmp; TMPDIR=/tmp): rocprofv3 --kernel-trace --stats -d out -- python scripts/emu_workloads.py hd
#
""" + rd("p6_hd.txt"))


def slomo():
    out = ["""# HBM traffic of ONE interpolation-UNet forward (12 -> 5 channels; 80 samples at 320x256, and 2 samples at 1280x704), round 6, per conv math
# commands (separate passes): V2E_AMD_CONV_MATH=<m> rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace -- python scripts/slomo_layers.py 80
# (three forwards per process; profiles/summarize_rocprof_pmc.py lists the per-launch average per kernel and the process total).
# FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; total / 3 forwards = bytes per forward below (the weight packing and the
# input generator of the script are in the total: < 1 %).
#"""]
    for m, shape in (("fp16x2", None), ("bf16x3", None), ("f32", None), ("hd", "2x704x1280")):
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            t = rd("p6_slomo_%s_%s.txt" % (m, c))
            mm = re.search(r"total over the whole process: ([0-9.]+)", t)
            tot[c] = float(mm.group(1)) * 1024 / 3 if mm else 0.0
        out.append("# unet_forward_bytes %s %.0f %.0f %s" % ("fp16x2" if m == "hd" else m, tot["FETCH_SIZE"], tot["WRITE_SIZE"], shape or "80x256x320"))
    out.append("#\n# algorithmic (every activation written once and read once per consumer, weights once): 16.4 GB per forward "
               "(v2e_amd.benchutil.unet_algorithmic_bytes);\n# avg_pool2d / bilinear x2 run as their own passes and add their reads and writes.\n")
    for m in ("fp16x2", "bf16x3", "f32", "hd"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out.append("# ---- %s %s\n" % ("fp16x2 at 2 x 1280x704 (slomo_layers.py 2 704 1280)" if m == "hd" else m, c) + rd("p6_slomo_%s_%s.txt" % (m, c)))
    wr("r06_slomo_counters.txt", "\n".join(out))
    wr("r06_slomo_per_layer.txt", """# Interpolation UNet (12 -> 5 channels) forward at 320x256, 80 samples (B = 8 pairs x U = 10: what bench.py's slomo leg and the
# 320x256 parity test run), per conv launch of the last of 3 forwards, round 6
# command: V2E_AMD_CONV_MATH=<m> rocprofv3 --kernel-trace --stats -- python scripts/slomo_layers.py 80   (parsed by scripts/parse_layers.py)
# TF = algorithmic f32 FLOPs of the layer / its launch duration (f32-equivalent).  First table: fp16x2 (what the default "auto"
# runs: two float16 pieces, operands staged times a power of two from the producers' range slots); second: bf16x3 (exact split).
#
""" + rd("p6_slomo_fp16x2_layers.txt") + "\n# ---- conv_math bf16x3\n" + rd("p6_slomo_bf16x3_layers.txt") +
       "\n# ---- conv_math f32 (v_mfma_f32_32x32x2_f32: the reference's own arithmetic type)\n" + rd("p6_slomo_f32_layers.txt"))


if __name__ == "__main__":
    emulator()
    hd()
    slomo()
