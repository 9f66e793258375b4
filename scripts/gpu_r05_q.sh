#!/bin/bash
# the pull event writer (k_cpull) against the push one (k_cemit): parity tests with it on, then A/B of the headline and HD legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
V2E_AMD_EMIT_PULL=1 timeout 900 python -m pytest tests/test_emulator_gpu.py tests/test_config3_vs_reference.py -m gpu -q -x > $O/pull_tests.log 2>&1
tail -4 $O/pull_tests.log
for rep in 1 2; do
for pull in 0 1; do
  echo "--- pull=$pull"
  V2E_AMD_EMIT_PULL=$pull timeout 300 python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_blocks']['Mevents_per_s'])"
done
done
