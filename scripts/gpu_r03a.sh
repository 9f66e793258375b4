#!/bin/bash
# round 3, call A: the new parity tests of the benchmarked paths on the round-2 kernels + the chain's in-kernel timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_emulator_bench_paths_gpu.py -q > $O/r03a_pytest.log 2>&1; echo "pytest rc $?" >> $O/r03a_pytest.log
tail -30 $O/r03a_pytest.log
timeout 300 python scripts/chain_timeline.py > $O/r03a_timeline.txt 2>&1; tail -25 $O/r03a_timeline.txt
