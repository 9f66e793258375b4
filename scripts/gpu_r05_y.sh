#!/bin/bash
# 64-frame emission batches on large grids (new default) against V2E_AMD_CHAIN_M=1: parity on the 1280x720 / multi-clip tests, then A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 400 python -m pytest tests/test_emulator_bench_paths_gpu.py tests/test_emulator_gpu.py -m gpu -q -x -k "hd or multi_clip or mid_size or bench_step or many_iter" < /dev/null > $O/m2_tests.log 2>&1; tail -1 $O/m2_tests.log
echo "--- default (64-frame batches)"; timeout 200 python scripts/emu_workloads.py hd batched < /dev/null 2>/dev/null | cut -c1-120
echo "--- V2E_AMD_CHAIN_M=1";  V2E_AMD_CHAIN_M=1 timeout 200 python scripts/emu_workloads.py hd batched < /dev/null 2>/dev/null | cut -c1-120
