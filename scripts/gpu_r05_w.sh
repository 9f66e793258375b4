#!/bin/bash
# NOTE: the V2E_AMD_PULL_WPF / V2E_AMD_PULL_NR / V2E_AMD_CFRAME1_MAX knobs this script sweeps existed only in the experiment builds of round 5
# (profiles/r05_emulator_experiments.txt items 8-10); the committed library has the measured values as constants.
# 1280x720: the per-frame tables by k_cframe (a workgroup per key row) instead of k_cframe1 (one workgroup per frame)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_emulator_bench_paths_gpu.py -m gpu -q -x < /dev/null > $O/cframe_tests.log 2>&1; tail -1 $O/cframe_tests.log
for mx in 4096 1024 4096 1024; do
echo "--- hd cframe1_max=$mx"; V2E_AMD_CFRAME1_MAX=$mx timeout 200 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-120
done
