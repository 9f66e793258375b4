#!/bin/bash
# NOTE: the V2E_AMD_PULL_WPF / V2E_AMD_PULL_NR / V2E_AMD_CFRAME1_MAX knobs this script sweeps existed only in the experiment builds of round 5
# (profiles/r05_emulator_experiments.txt items 8-10); the committed library has the measured values as constants.
# k_cpull rows per thread (V2E_AMD_PULL_NR = 1 / 2 / 4): parity, then headline / 1280x720 / 64 clips.  Every step bounded.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
for nr in 2 4; do
V2E_AMD_PULL_NR=$nr timeout 200 python -m pytest tests/test_emulator_gpu.py -m gpu -q -x < /dev/null > $O/pullnr${nr}_tests.log 2>&1; tail -1 $O/pullnr${nr}_tests.log
done
for nr in 1 2 4 1 2 4; do
  echo "--- headline NR=$nr"
  V2E_AMD_PULL_NR=$nr timeout 120 python bench.py --steps 20 --warmup 5 --blocks 3 --no-extras --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_blocks']['Mevents_per_s'])"
done
for nr in 1 2 4; do
echo "--- hd + batched NR=$nr"; V2E_AMD_PULL_NR=$nr timeout 200 python scripts/emu_workloads.py hd batched < /dev/null 2>/dev/null | cut -c1-120
done
