#!/usr/bin/env python
"""Assemble the committed round-3 profile artefacts (profiles/r03_*.txt) from what scripts/gpu_r03_profiles.sh and
scripts/gpu_r03_slomo_profiles.sh left under gpurun_out/ (rocprofv3 summaries made on the MI355X box)."""
import os
import re

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")


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


def emulator():
    kt, tl = rd("p3_kt.txt"), rd("p3_kt_timeline.txt")
    window = rd("r03g_kt_window.txt") if os.path.exists(os.path.join(G, "r03g_kt_window.txt")) else ""
    wr("r03_emulator_chain_kernel_trace.txt", """# rocprofv3 kernel trace of the headline workload, round 3 (k_ahead | k_chain | k_ctot + k_cframe1 + k_cemit, one hipGraph per run)
# command (on the MI355X box, cd /tmp; TMPDIR=/tmp):
#   rocprofv3 --kernel-trace --stats -d out -- python bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline
# summarised by profiles/summarize_rocprof_db.py (top kernels) and scripts/kernel_timeline.py (k_chain launch timeline).
# workload: BASELINE configs[1], 346x260, one clip, 300 frames per step, CLI-default DVS parameters, Philox.
# one k_chain launch = 32 frames (10 launches per step + a tail launch validating the last speculation; a redo pass runs inside
# the launch that finds the miss: p90 of the duration below); k_ahead / k_ctot / k_cframe1 / k_cemit: one launch per batch of
# 64 frames.  The torch elementwise kernels are bench.py's synthetic-video generator (outside the timed region).
# Reading it: the kernels of the three streams run side by side on the same CUs, so every duration below is a duration UNDER
# CONTENTION (k_chain alone, everything else switched off: 33 us per 32 frames; here 54 us p50): the workload is bound by
# instruction issue of the three streams together, not by any one of them and not by HBM.  bench.py measures the chain
# kernel live with HIP events (all kernels on one stream for that one run) and reports this trace's figure beside it.
#
""" + kt + "\n# k_chain launch timeline (same trace)\n" + tl +
       ("\n# a window of the same workload's trace (scripts/trace_window.py; an earlier run of this round, same code path):\n"
        "# the chain launches back to back (gap p50 10 us) with k_ahead / k_ctot / k_cemit beside them\n" + window if window else ""))
    f, w = rd("p3_FETCH_SIZE.txt"), rd("p3_WRITE_SIZE.txt")
    rows = []
    for k in ("k_chain<double, unsigned char, false>", "k_ahead<unsigned char>", "k_cemit", "k_ctot", "k_cframe1"):
        n, fa = pmc_avg(f, k)
        n2, wa = pmc_avg(w, k)
        if n is None and n2 is None:
            continue
        rows.append("# %-38s %4d  %9.1f  %9.1f" % (k, n or n2, fa or 0.0, wa or 0.0))
    wr("r03_emulator_pmc_hbm.txt", """# HBM traffic of the emulator kernels, round 3
# commands (separate passes, as the MI355X guide prescribes; summary by profiles/summarize_rocprof_pmc.py):
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline
# workload: 346x260, one clip, 300 frames/step, CLI-default DVS parameters; a k_chain launch covers 32 frames, the others 64.
# units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB (summed over XCDs); no x2 correction applied (the guide calibrates
# it for wide coalesced streams; these kernels load 1-16 B per lane).  The summariser lists the top kernels of each pass: a
# kernel missing from one of them (0.0 below) moved less than the listed ones there.
#
# kernel                                launches   FETCH_SIZE avg [KiB]   WRITE_SIZE avg [KiB]
""" + "\n".join(rows) + """
#
# per launch, what the kernels touch (algorithmic):
#   k_ahead (64 frames): reads 64 u8 frames 5.8 MB, writes 64 x 89 960 x 16 B records = 92 MB (84 MB measured: WRITE_SIZE).
#   k_chain (32 frames): reads the records 46 MB + the state once (28 B/px = 2.5 MB); writes the count word (u32) per pixel and
#             frame 11.5 MB + state once + the ping-pong planes and checkpoints a redo needs.  Fetched 28 MB: most of the
#             records of the k_ahead launch before are still in L2 / MALL.  Round 2 wrote 31 MB per launch; the per-wave max /
#             total tables moved to k_ctot (group granularity, u16) this round: 21 MB.
#   k_ctot  (64 frames): reads the count words 23 MB (11 MB fetched: L2/MALL), writes u16 group tables.
#   k_cemit (64 frames): reads count words + tables, writes 64 x ~35 700 events x 16 B = 36.5 MB; 66.5 MB measured -- the
#             per-iteration shuffle scatters 16-byte rows (1.8x write amplification, as in rounds 1 and 2).
# nothing on this path is bounded by HBM (whole frame: ~5.4 MB in 3.5 us = 1.5 TB/s, most of it L2 / MALL hits).
#
# raw summaries:
""" + f + w)
    wr("r03_emulator_sq.txt", """# SQ counters of the emulator kernels on the three benchmark workloads, round 3
# command: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
#          --kernel-trace -- python scripts/emu_workloads.py      (summary: profiles/summarize_rocprof_sq.py)
# scripts/emu_workloads.py runs, in this order: the headline clip (346x260, 1 clip: k_chain<.., false> + k_ahead), the 64-clip
# batch (346x260 x 64: k_chain<.., true>, K = 8, grid 180 224 threads x 64 clips) and 1280x720 noisy (k_chain<.., true>, K = 32).
# *_w = per wave and launch.  Per wave and FRAME on the headline path: k_chain 3 222 / 32 = 101 VALU (round 2: 190);
# k_ahead 279 per wave, a wave covering 64 pixels of a PAIR of frames; emission per 256-pixel group and frame: k_ctot 332 / 2
# (a wave takes two frames of its group) + k_cemit 659 = 825 (round 2, per 64-pixel wave 289 + 9, i.e. 1 190 per 256 pixels).
#
""" + rd("p3_sq.txt"))


def scheduling():
    tl = rd("p3_kt_timeline.txt")
    wr("r03_graph_scheduling.txt", """# How the kernels of a device-resident run are scheduled, and what the runtime does with it -- round 3
# (ROCm 7.0.2 / HIP 7.0.5, gfx950; measured with scripts/chain_ab.py, scripts/chain_timeline.py, scripts/trace_window.py,
#  scripts/kernel_timeline.py over rocprofv3 --kernel-trace runs of `bench.py --steps 4 --warmup 1 --blocks 1 --no-extras`)
#
# A run of F frames is: k_ahead (per 64 frames; records for the chain), k_chain (per 32 frames; the dependency chain),
# and per 64-frame batch the emission k_ctot -> k_cframe1 -> k_cemit.  Dependencies: chain launch l needs the records of its
# frames and launch l - 1; the emission of batch b needs the chain launches covering it; k_ahead of batch b + 3 needs the
# emission of batch b (the ring of frame slots).  Three streams, joined by events, captured once as a hipGraph and replayed.
#
# Findings (each cost at least one variant that was built and measured):
#  1. hipStreamEndCapture SEGFAULTS when the capture contains a dependency between two forked streams that does not pass
#     through the origin stream (event recorded on fork A, waited on fork B).  Edges origin <-> fork are fine.  The run is
#     therefore captured with the emission tables AND rows on ONE side stream; the chain on the origin; k_ahead on a second
#     fork that only ever talks to the origin.
#  2. A graph built node by node (hipGraphAddKernelNode with explicit dependencies: no capture, any edge possible; the
#     `Sched` helper in emu.hip can emit either form, V2E_AMD_GRAPH_EXPLICIT=1) EXECUTES SERIALLY on this runtime: no two
#     nodes overlap whatever the dependency structure says; the run then takes the serial sum of all its kernels.  Not used.
#  3. A captured graph overlaps its branches only if the capture's enqueue order lets it: with the side stream forked at the
#     start of the run, or the emission enqueued before the k_ahead it does not depend on, the branches ran one after the
#     other.  Enqueue order per batch: k_ahead first, then the chain launches, then the emission; the side stream forks at
#     the first event it needs.  (V2E_AMD_ORDER_EMISSION_FIRST / V2E_AMD_INITIAL_SIDE_FORK reproduce the slow orders.)
#  4. On plain streams (no graph) a cross-stream dependency costs ~19 us between the end of the producer and the start of
#     the consumer (event record -> wait), ~3.5-4 us inside a captured graph (scripts/ubench_latency.hip).  The instrumented run bench.py uses for the live
#     kernel time therefore puts all kernels on one stream.
#  5. With the three branches overlapping, every kernel is stretched: k_chain 33 us alone -> 52-54 us p50 beside the others,
#     k_ctot 11 -> 37 us, k_ahead 31 -> 48 us.  The sum of the stand-alone durations per 64 frames (2 x 33 + 31 + 11 + 5 + 30
#     = 143 us) is about what the overlapped schedule takes (2 x 65 us chain period): the workload is bound by instruction
#     issue over all CUs, so overlap hides latencies and launch gaps but not work.  What raised throughput this round was
#     removing instructions (k_chain 190 -> 101 VALU per wave and frame; emission per 256-pixel group), not scheduling.
#  6. Emission kernels are bound by per-wave latency chains (ballot -> LDS -> scatter), not by wave dispatch: making a wave
#     handle several frames (k_ctot, k_cemit, k_ahead variants) was slower (37 vs 13 us, 45 vs 34 us), making it handle 256
#     pixels of one frame instead of 64 was faster (fewer waves, the same chain length).
#
# k_chain launch timeline of the committed trace (profiles/r03_emulator_chain_kernel_trace.txt):
""" + tl + """
# gap p50 10.6 us: the chain launches follow one another without waiting for anything else (chain-bound); the p90 gap and
# duration are the redo launches (a frame of the previous launch turned out to be rule-on: that launch's frames are redone
# from the checkpoint below it before this launch's own).
""")


def slomo():
    wr("r03_slomo_per_layer.txt", """# Interpolation UNet (12 -> 5 channels) forward at 320x256, 80 samples (B = 8 pairs x U = 10: what bench.py's slomo leg and
# the 320x256 parity test run), per conv launch of the last of 3 forwards, round 3
# command: rocprofv3 --kernel-trace --stats -- python scripts/slomo_layers.py 80   (parsed by scripts/parse_layers.py)
# TF = algorithmic f32 FLOPs of the layer / its launch duration ("f32-equivalent": the split-bf16 kernels execute 6 bf16
# multiply-adds per f32 one).  <s3p TW, DBG> = k_conv_s3p (slomo_s3p.h: 64 x 64 register tile, one software-pipelined wave per
# SIMD, new this round); <s3 KS, CT, PT, WP, TW, NB, MODE, RG> = k_conv_s3 (slomo_s3.h); <KS, CI_T, ...> = k_conv (f32 MFMA).
# First table: conv_math bf16x3 (the exact three-piece split); second: fp16x2 (what the default "auto" runs, plus its range guard).
# Box-to-box spread of these kernels is +-3 % (they run at the chip's power limit: r03_slomo_s3p_ablation.txt); on one box,
# A/B: forward 27.68 ms with k_conv_s3 everywhere, 27.28 ms with k_conv_s3p where it fits (scripts/gpu_slomo_ab.sh).
#
""" + rd("p3_slomo_layers.txt") + (
        "\n# ---- conv_math fp16x2 (two float16 pieces, three products; <s3 KS, CT, PT, WP, TW, NB, MODE, RG, NP = 2>), same command with\n"
        "# V2E_AMD_CONV_MATH=fp16x2\n" + rd("p3_slomo_h2_layers.txt") if os.path.exists(os.path.join(G, "p3_slomo_h2_layers.txt")) else ""))
    wr("r03_slomo_s3p_ablation.txt", """# k_conv_s3p against k_conv_s3 on single layers, with the pipelined kernel's pieces switched off one at a time, and the
# SHADER CLOCK the kernel actually runs at -- round 3
# command: S3P_TIMELINE=1 V2E_AMD_S3_VARIANT={12: k_conv_s3 | 11: k_conv_s3p} [V2E_AMD_S3P_DBG=d] scripts/conv_s3_check ks cin cout n h w
# (scripts/gpu_r03_slomo_profiles.sh).  conv_s3_check feeds inputs 4x the network's magnitudes; TF = f32-equivalent.
# clocks/step: s_memtime of workgroup 0, one step = one kernel row of one 16-channel chunk = 72 multiplies
# (v_mfma_f32_32x32x16_bf16, 32 clocks each at full rate: 2 304 clocks is the floor); MHz: the same interval on the constant
# 100 MHz clock (s_memrealtime), i.e. the shader clock under this kernel's load (nominal 2 400).
# dbg 1: no side work at all (multiplies + their operand reads + one barrier per step; results wrong by construction);
# dbg 4: no global reloads; dbg 8: no LDS stores (and with them no loads: dead).
#
""" + rd("p3_s3p_ablation.txt") + """#
# Reading it:
#  * The clock is 1.68-1.79 GHz under the full kernel and 1.90 GHz with the side work removed: the chip is at its POWER
#    limit in these kernels.  Across the variants built this round the clocks per step fell 3 208 -> 2 989 (micro-slot
#    interleaving, scalar-base addressing) while the wall time per step stayed 1.82-1.83 us: every issue slot saved was
#    paid back in clock.  What changes the wall time is energy per multiply: dbg 4 / dbg 8 (no global / LDS traffic) save
#    12-17 %, and the 64 x 64 register tile (half the LDS operand bytes per multiply of k_conv_s3's 32 x 64) is where
#    k_conv_s3p's 6-9 % per layer come from.
#  * The same limit is what the bare matrix-pipe microbenchmark shows (r03_mfma_bare.txt): 98 % of the 2.5 PFLOP/s dense
#    bf16 peak on zeros, 73 % on normal(0,1) operands, 67 % on random bit patterns -- 280-305 TF/s f32-equivalent is the
#    ceiling of ANY split-bf16 convolution on real data on this chip; the complete kernels reach 155-200.
""")
    wr("r03_mfma_bare.txt", """# Bare v_mfma_f32_32x32x16_bf16 stream (scripts/ubench_mfma.hip: no loads, no LDS, no stores in the loop; operands loaded
# once), by operand data, independent accumulators per wave and waves per SIMD -- round 3, one MI355X
# f32-equiv = bf16 rate / 6 (the six piece products of the split-f32 convolution).
# Two things to read off: (1) the rate depends on the DATA (power: zeros 98 %, normal(0,1) 73 %, random bits 67 % of the
# 2.5 PFLOP/s dense peak); (2) four accumulator chains x two waves per SIMD is slow whatever the data (66 % on zeros, where
# 4 x 1, 2 x 2 and 8 x 1 reach 98 %) -- the reason k_conv_s3p runs its 64 x 64 tile as ONE wave per SIMD.
#
""" + rd("r03_mfma_bare.txt"))


if __name__ == "__main__":
    emulator()
    scheduling()
    slomo()
