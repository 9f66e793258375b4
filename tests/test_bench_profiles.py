"""bench.py's roofline objects quote three committed rocprofv3 summaries (counters cannot be read from inside the process): the
parsers must find THIS round's files and read sane numbers out of them -- a format drift would otherwise fall back to an older
round's file, or to null, without a sound."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_emulator_profile_parsers_read_this_rounds_files():
    import bench as B
    traffic, src = B.pmc_traffic_per_launch("k_chain")
    assert src == "profiles/" + B.PMC_FILES[0] and "r06" in src
    assert 20e6 < traffic < 153e6                      # FETCH + WRITE per 32-frame launch: below the 153 MB priced (state in registers)
    us, src = B.rocprof_kernel_us("k_chain")
    assert src == "profiles/" + B.TRACE_FILES[0] and "r06" in src and 20.0 < us < 120.0   # (the cross-check of the live figure)
    ii = B.instruction_issue(3.0e-3)                   # 3 us per frame
    assert ii["source"] == "profiles/" + B.SQ_FILES[0] and "r06" in ii["source"]
    assert 500 < ii["per_64px_wave_frame"] < 1200 and 0.05 < ii["frac"] < 1.0
    # float64 VALU instructions are priced at 4 cycles (round-4 review): the vector pipe's time is above the all-at-2-cycles figure by
    # each kernel's static float64 share, and the per-kernel lines add up to the headline totals
    pk = ii["per_kernel"]
    assert {q["kernel"] for q in pk} >= {"k_chain", "k_ahead", "k_ctot", "k_cpull"}
    assert abs(sum(q["valu"] for q in pk) - ii["valu_per_frame"]) <= 0.01 * ii["valu_per_frame"]
    assert 0.2 < next(q for q in pk if q["kernel"] == "k_chain")["f64_share_static"] < 0.6
    all_two = ii["valu_per_frame"] / B.VALU_RATE * 1e6
    assert all_two < ii["valu_us_per_frame"] < 1.5 * all_two
    assert ii["bound_us_per_frame"] == max(ii["valu_us_per_frame"], ii["salu_us_per_frame"])


def test_slomo_traffic_reads_this_rounds_counters_per_conv_math():
    from v2e_amd.benchutil import slomo_pmc_traffic, unet_algorithmic_bytes
    alg = unet_algorithmic_bytes(80, 12, 5, 256, 320)
    for math in ("fp16x2", "bf16x3", "f32"):
        total, detail = slomo_pmc_traffic(math, alg)
        assert 0.8 * alg < total < 4 * alg, (math, total, alg)   # measured 2 - 2.9x the algorithmic bytes
        assert "r06" in str(detail)
    hd = unet_algorithmic_bytes(2, 12, 5, 704, 1280)
    total, detail = slomo_pmc_traffic("fp16x2", hd, "2x704x1280")   # the HD shape bench.py's slomo_hd leg runs
    assert 0.8 * hd < total < 4 * hd and detail["shape"] == "2x704x1280"
    assert slomo_pmc_traffic("bf16x3", hd, "2x704x1280")[0] is None  # no pass of that math at that shape: null, not a borrowed number
