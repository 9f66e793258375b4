#!/bin/bash
# k_chain's ring accesses through buffer resources (slot in the scalar offset): parity, then the headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 200 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -m gpu -q -x < /dev/null > $O/buf_tests.log 2>&1; tail -1 $O/buf_tests.log
for r in 1 2 3; do
timeout 100 python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_blocks']['Mevents_per_s'], d['roofline']['alone_hip_events']['avg_kernel_us'], d['roofline']['median_full_launch'])"
done
