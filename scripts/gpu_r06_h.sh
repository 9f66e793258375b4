#!/bin/bash
# round 6: pipelined plain launches (three streams, no graph, no join): parity, then stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -m gpu -q -x < /dev/null > $O/ov_pytest.log 2>&1
tail -3 $O/ov_pytest.log
echo "== graph, overlap=0"; V2E_AMD_BENCH_OVERLAP=0 timeout 300 python scripts/chain_stamps.py 40 1 2>&1 | grep -v amdgpu.ids | head -9
echo "== pipelined plain (use_graph=0, overlap=1)"; V2E_AMD_BENCH_UG=0 timeout 300 python scripts/chain_stamps.py 40 2 2>&1 | grep -v amdgpu.ids | head -12
