#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
B="python bench.py --steps 20 --warmup 5 --blocks 7 --no-extras --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["timed_blocks"]["Mevents_per_s"])'
for rep in 1 2 3; do
  V2E_AMD_ZERO_FIRST=1 V2E_AMD_BENCH_STAGE_FRAMES=1 V2E_AMD_CTL_UPLOAD_MEMCPY=1 $B 2>/dev/null | python -c "$P" r04-config
  V2E_AMD_ZERO_FIRST=1 V2E_AMD_BENCH_STAGE_FRAMES=1 $B 2>/dev/null | python -c "$P" kernel-upload
  V2E_AMD_ZERO_FIRST=1 $B 2>/dev/null | python -c "$P" kernel-upload+inplace
  $B 2>/dev/null | python -c "$P" kernel-upload+inplace+fork-first
  V2E_AMD_ZERO_FIRST=1 V2E_AMD_TAIL_TABS_MAIN=1 $B 2>/dev/null | python -c "$P" kernel-upload+inplace+tailtabs
done > $O/ab_g.txt 2>&1
cat $O/ab_g.txt
