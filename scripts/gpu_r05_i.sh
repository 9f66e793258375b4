#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_check_ckpt_gpu.py -x -q 2>&1 > $O/new_tests_full.log
grep -n "Error\|assert\|^E " $O/new_tests_full.log | head -30
grep -n "end to end\|layer walk" $O/new_tests_full.log | head
timeout 900 python -m pytest tests/test_concurrency_gpu.py tests/test_device_isa.py tests/test_slomo_gpu.py -x -q -k "not hd_shape" 2>&1 | tail -5
