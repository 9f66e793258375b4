#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3; do timeout 200 python -m pytest tests/test_emulator_bench_paths_gpu.py -m gpu -q -x --timeout 120 --timeout-method=thread -k "bench_step_loop" < /dev/null 2>&1 | tail -2; done
echo "== rows on the tables' stream"; for i in 1 2; do V2E_AMD_PIPE_ROWS=one timeout 200 python -m pytest tests/test_emulator_bench_paths_gpu.py -m gpu -q -x --timeout 120 --timeout-method=thread -k "bench_step_loop" < /dev/null 2>&1 | tail -2; done
echo "== not pipelined"; V2E_AMD_BENCH_PIPELINED=0 timeout 200 python -m pytest tests/test_emulator_bench_paths_gpu.py -m gpu -q -x --timeout 120 --timeout-method=thread -k "bench_step_loop" < /dev/null 2>&1 | tail -2
