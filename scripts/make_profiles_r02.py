#!/usr/bin/env python
"""Assemble the committed round-2 profile artefacts (profiles/r02_*.txt) from what scripts/gpu_r02_profiles.sh and
scripts/gpu_slomo_pmc.sh left under gpurun_out/ (rocprofv3 summaries made on the MI355X box)."""
import os
import re

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")


def rd(name):
    return open(os.path.join(G, name)).read()


def pmc_avg(txt, kernel_prefix):
    for line in txt.splitlines():
        if kernel_prefix in line:
            m = re.search(r"launches\s+(\d+)\s+avg\s+([0-9.]+)", line)
            return int(m.group(1)), float(m.group(2))
    return None, None


def total(txt):
    m = re.search(r"# total over the whole process: ([0-9.]+)", txt)
    return float(m.group(1))


def emulator():
    kt, tl = rd("p_kt.txt"), rd("p_kt_timeline.txt")
    open(os.path.join(R, "profiles", "r02_emulator_chain_kernel_trace.txt"), "w").write(
        """# rocprofv3 kernel trace of the headline workload, round 2 (chain pipeline: k_ahead | k_chain | k_cframe1 + k_cemit)
# command (on the MI355X box, cd /tmp; TMPDIR=/tmp):
#   rocprofv3 --kernel-trace --stats -d out -- python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline
# summarised by profiles/summarize_rocprof_db.py (top kernels) and scripts/kernel_timeline.py (k_chain launch timeline).
# workload: BASELINE configs[1], 346x260, one clip, 300 frames per step, CLI-default DVS parameters, Philox.
# one k_chain launch = 32 frames (10 launches per step; +1 tail launch validating the last speculation; redo passes
# run inside the launch that finds the miss: p90 of the duration below); one k_ahead / k_cframe1 / k_cemit launch = 32
# frames (an event batch).  The torch elementwise kernels are bench.py's synthetic-video generator (outside the timed
# region).  The profiler lengthens launch gaps; bench.py's HIP-event figure is the one on the line.
#
""" + kt + "\n# k_chain launch timeline (same trace)\n" + tl)
    f, w = rd("p_FETCH_SIZE.txt"), rd("p_WRITE_SIZE.txt")
    rows = []
    for k in ("k_chain<double, unsigned char, false>", "k_ahead<unsigned char>", "k_cemit", "k_cframe1"):
        n, fa = pmc_avg(f, k)
        _, wa = pmc_avg(w, k)
        if n is not None and wa is not None:
            rows.append("# %-38s %4d  %9.1f  %9.1f" % (k, n, fa, wa))
    open(os.path.join(R, "profiles", "r02_emulator_pmc_hbm.txt"), "w").write(
        """# HBM traffic of the emulator kernels (chain pipeline), round 2
# commands (separate passes, as the MI355X guide prescribes; summary by profiles/summarize_rocprof_pmc.py):
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline
# workload: 346x260, one clip, 300 frames/step, CLI-default DVS parameters; a launch of each kernel covers 32 frames.
# units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB (summed over XCDs); no x2 correction applied (the guide
# calibrates it for wide coalesced streams; these kernels load 1-16 B per lane).
#
# kernel                                launches   FETCH_SIZE avg [KiB]   WRITE_SIZE avg [KiB]
""" + "\n".join(rows) + """
#
# per 32-frame launch, what the kernels touch (algorithmic):
#   k_ahead : reads 32 u8 frames 2.9 MB, writes 32 x 89 960 x 16 B records = 46.1 MB.
#   k_chain : reads the records 46.1 MB + the state once (lp 8, base 8, thresholds 8, ts_mem 4 = 28 B/px = 2.5 MB);
#             writes count word (u32) + per-wave max and key totals per frame 11.5 MB + state once 1.8 MB + the
#             ping-pong planes a redo needs.  Fetched less than that: the records of the k_ahead launch before are
#             partly still in L2 / MALL.
#   k_cemit : reads count words 11.5 MB + tables, writes 32 x ~35 700 events x 16 B = 18.3 MB; measured more: the
#             per-iteration shuffle scatters 16-byte rows (1.8x write amplification, as in round 1) and the count plane
#             is read by k_cframe1 and k_cemit.
# whole frame: ~5.5 MB per frame vs 5.34 MB algorithmic (53 B/px + 16 B/event) -- the split pipeline moves the 16-byte
#   records through memory once more than a fused one would; at 346x260 this is free (5.5 MB in 4.3 us = 1.3 TB/s, most
#   of it L2/MALL hits), at 1280x720 it is not, which is why large grids build the records inside k_chain
#   (k_chain<..., true>, no k_ahead; DESIGN.md section 3).
#
# raw summaries:
""" + f + w)
    p = os.path.join(R, "profiles", "r02_emulator_sq.txt")
    old = open(p).read()
    head = old[:old.index("# from ")] if "# from " in old else ""
    tail = old[old.index("# ---- raw summary of the BEFORE pass"):]
    open(p, "w").write(head + rd("p_sq.txt") + "\n" + tail)


def slomo():
    lay, lay32 = rd("p_slomo_layers.txt"), rd("p_slomo32_layers.txt")
    open(os.path.join(R, "profiles", "r02_slomo_per_layer.txt"), "w").write(
        """# Interpolation UNet (12 -> 5 channels) forward at 320x256, 80 samples (B = 8 pairs x U = 10: what bench.py's slomo leg and
# the 320x256 parity test run), per conv launch of the last of 3 forwards, round 2
# command: rocprofv3 --kernel-trace --stats -- python scripts/slomo_layers.py 80   (parsed by scripts/parse_layers.py)
# TF = algorithmic f32 FLOPs of the layer / its launch duration ("f32-equivalent" for the split-bf16 kernels, which
# execute 6 bf16 multiply-adds per f32 one).  <s3 KS, CT, PT, WP, TW, NB> = k_conv_s3 (slomo_s3.h); <KS, CI_T, CT, PT, WP,
# TW, PRE, STK> = k_conv (f32 matrix-core instructions).
#
# ---- default conv math (bf16x3: f32 operands split exactly into three bf16 pieces)
""" + lay + """
# ---- V2E_AMD_CONV_MATH=f32 (the round-1 kernels), same command
""" + lay32)
    mf = rd("p_slomo_mfma.txt")
    ft, wt = rd("q_slomo_bf16x3_FETCH_SIZE.txt"), rd("q_slomo_bf16x3_WRITE_SIZE.txt")
    ft32, wt32 = rd("q_slomo_f32_FETCH_SIZE.txt"), rd("q_slomo_f32_WRITE_SIZE.txt")

    def per_forward(t):
        # 3 forwards per process; everything else in the process (weight packing, input generation) subtracted
        other = 0.0
        for line in t.splitlines():
            if any(k in line for k in ("k_pack_weight", "at::native", "copyBuffer")):
                m = re.search(r"launches\s+(\d+)\s+avg\s+([0-9.]+)", line)
                other += int(m.group(1)) * float(m.group(2))
        return (total(t) - other) / 3.0 * 1024.0

    fb, wb, fb32, wb32 = per_forward(ft), per_forward(wt), per_forward(ft32), per_forward(wt32)
    open(os.path.join(R, "profiles", "r02_slomo_counters.txt"), "w").write(
        """# Interpolation UNet forward, 80 samples at 320x256: HBM traffic and matrix-core counters, round 2
# commands (separate passes): rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python scripts/slomo_layers.py 80   (and WRITE_SIZE),
#   rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
#             SQ_INSTS_MFMA --kernel-trace -- python scripts/slomo_layers.py 80
# FETCH_SIZE / WRITE_SIZE in KiB per launch, summed over XCDs; one forward = (process total - weight packing - input
# generation) / 3 forwards, in bytes:
# unet_forward_bytes %.0f %.0f
#   = %.1f GB fetched + %.1f GB written per forward (default conv math); with V2E_AMD_CONV_MATH=f32: %.1f + %.1f GB.
# algorithmic (every activation tensor written once and read once by each consumer, pooled / upsampled copies included):
#   12.1 GB read + 10.3 GB written + 0.12 GB of split weights.  Writes match; what is fetched beyond 12.1 GB is the input
#   patch of a pixel tile re-read once per 32-channel block of the output (cout/32 times, plus the halo) where those
#   blocks do not meet in one L2, and the weight slices re-read per pixel tile -- the channel blocks of a tile are
#   consecutive on one XCD for the layers with >= 4 blocks (slomo_s3.h).  At 26 ms per forward the fetch rate is
#   ~1 TB/s: the convolutions are not HBM-bound (bound: the matrix pipe, DESIGN.md section 4).
# matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) per kernel from the table below, e.g.
#   k_conv_s3<3,1,2,4,32> 0.51, k_conv_s3<7,...> 0.62, and with V2E_AMD_CONV_MATH=f32 k_conv<7,4,...> 0.81 (first pass of the round).
#
# ---- matrix-core counters (default conv math)
""" % (fb, wb, fb / 1e9, wb / 1e9, fb32 / 1e9, wb32 / 1e9) + mf + "\n# ---- HBM traffic, default conv math\n" + ft + wt +
        "\n# ---- HBM traffic, V2E_AMD_CONV_MATH=f32\n" + ft32 + wt32)
    print("slomo forward bytes: fetch %.2f GB write %.2f GB (f32 math: %.2f / %.2f)" % (fb / 1e9, wb / 1e9, fb32 / 1e9, wb32 / 1e9))


if __name__ == "__main__":
    emulator()
    slomo()
