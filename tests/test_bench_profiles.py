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
    assert src == "profiles/" + B.PMC_FILES[0] and "r04" in src
    assert 20e6 < traffic < 153e6                      # FETCH + WRITE per 32-frame launch: below the 153 MB priced (state in registers)
    us, src = B.rocprof_kernel_us("k_chain")
    assert src == "profiles/" + B.TRACE_FILES[0] and "r04" in src and 20.0 < us < 120.0
    ii = B.instruction_issue(3.0e-3)                   # 3 us per frame
    assert ii["source"] == "profiles/" + B.SQ_FILES[0] and "r04" in ii["source"]
    assert 500 < ii["per_64px_wave_frame"] < 1200 and 0.05 < ii["frac"] < 1.0
    assert ii["bound_us_per_frame"] == round(max(ii["valu_per_frame"] / B.VALU_RATE, ii["salu_per_frame"] / B.SALU_RATE) * 1e6, 4)


def test_slomo_traffic_reads_this_rounds_counters_per_conv_math():
    from v2e_amd.benchutil import slomo_pmc_traffic, unet_algorithmic_bytes
    alg = unet_algorithmic_bytes(80, 12, 5, 256, 320)
    for math in ("fp16x2", "bf16x3"):
        total, detail = slomo_pmc_traffic(math, alg)
        assert 0.8 * alg < total < 4 * alg, (math, total, alg)   # measured ~2x the algorithmic bytes
        assert "r04" in str(detail)
    assert slomo_pmc_traffic("f32", alg)[0] is None     # no committed pass for the f32 kernels: null, not a borrowed number
