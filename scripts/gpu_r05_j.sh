#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
B="python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["timed_blocks"]["Mevents_per_s"])'
for rep in 1 2; do
  timeout 120 $B 2>/dev/null | python -c "$P" old-ring3
  V2E_AMD_CHAIN_RING=5 timeout 120 $B 2>/dev/null | python -c "$P" old-ring5
  V2E_AMD_CHAIN_SEGS=3 V2E_AMD_CHAIN_SEGS_PLAN_ONLY=1 timeout 120 $B 2>/dev/null | python -c "$P" new-plan-1seg-launches
  V2E_AMD_CHAIN_SEGS=3 timeout 120 $B 2>/dev/null | python -c "$P" segs=3
  V2E_AMD_CHAIN_SEGS=9 timeout 120 $B 2>/dev/null | python -c "$P" segs=9
done > $O/ab_segs.txt 2>&1
cat $O/ab_segs.txt
cd /tmp
V2E_AMD_CHAIN_SEGS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p5_kt -- python $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-extras --no-cpu-baseline > $O/p5_kt.log 2>&1
cd $R
python scripts/dump_timeline.py $O/p5_kt 0.35 3000 > $O/p5_timeline_segs3.txt 2>&1
rm -rf $O/p5_kt
