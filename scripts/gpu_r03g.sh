#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -q -x > $O/r03g_pytest.log 2>&1; tail -3 $O/r03g_pytest.log
python bench.py --steps 50 --warmup 5 --blocks 3 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
timeout 600 python scripts/emu_workloads.py batched hd 2>&1 | tail -2 | cut -c1-110
