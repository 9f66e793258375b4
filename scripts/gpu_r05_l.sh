#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
B="python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["timed_blocks"]["Mevents_per_s"])'
for rep in 1 2; do
  timeout 120 $B 2>/dev/null | python -c "$P" default
  HIP_FORCE_QUEUE_PROFILING=1 timeout 120 $B 2>/dev/null | python -c "$P" force-queue-profiling
  for q in 1 2 3 6 8; do GPU_MAX_HW_QUEUES=$q timeout 120 $B 2>/dev/null | python -c "$P" max-hw-queues=$q; done
  HSA_ENABLE_SDMA=0 timeout 120 $B 2>/dev/null | python -c "$P" no-sdma
  DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_GRAPH_... true 2>/dev/null
done > $O/ab_env.txt 2>&1
cat $O/ab_env.txt
