#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 100 python scripts/chain_stamps.py 40 0 2>&1 | grep -v amdgpu.ids | head -7 | grep -v "^duration"
timeout 500 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py tests/test_csdvs.py tests/test_concurrency_gpu.py -m gpu -q -x --timeout 90 --timeout-method=thread < /dev/null > $O/ov_pytest.log 2>&1
tail -3 $O/ov_pytest.log
timeout 100 python scripts/chain_stamps.py 40 0 2>&1 | grep -v amdgpu.ids | head -7 | grep -v "^duration"
timeout 100 python scripts/host_timeline.py 60 2>&1 | grep "steps 60"
