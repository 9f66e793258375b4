#!/bin/bash
# round 6, experiment 1: K = 64 frames per chain launch on the single-clip grid (A/B against K = 32 in one session) + parity with it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
V2E_AMD_CHAIN_K=64 timeout 600 python -m pytest tests/test_emulator_gpu.py -m gpu -q -x -k "pipelines_agree_on_benchmark_clip or chain_launch_lengths or clip_in_small_runs or split_clip" < /dev/null > $O/k64_pytest.log 2>&1
tail -3 $O/k64_pytest.log
B="python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline"
for rep in 1 2; do
  for K in 32 64 48; do
    V2E_AMD_CHAIN_K=$K timeout 300 $B < /dev/null > $O/k64_bench_${K}_$rep.json 2> $O/k64_bench_${K}_$rep.log
    python - <<P
import json
d=json.loads(open('gpurun_out/k64_bench_${K}_$rep.json').read().strip().splitlines()[-1])
print('K=$K rep $rep', d['value'], d['ms_per_step'], d.get('timed_blocks'))
P
  done
done
