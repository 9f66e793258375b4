#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
B="python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["timed_blocks"]["Mevents_per_s"])'
for rep in 1 2; do
  $B 2>/dev/null | python -c "$P" default
  V2E_AMD_BENCH_STAGE_FRAMES=1 $B 2>/dev/null | python -c "$P" staged-frames
  V2E_AMD_NO_TAIL_TABS_MAIN=1 $B 2>/dev/null | python -c "$P" tail-tabs-on-side
done > $O/ab_f.txt 2>&1
cat $O/ab_f.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p5_kt -- python $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-extras --no-cpu-baseline > $O/p5_kt.log 2>&1
cd $R
python scripts/dump_timeline.py $O/p5_kt 0.35 3000 > $O/p5_timeline_c.txt 2>&1
rm -rf $O/p5_kt
timeout 600 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -x -q 2>&1 | tail -3
