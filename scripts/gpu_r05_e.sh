#!/bin/bash
# A/B of the run's ctl upload (kernel reading pinned memory vs hipMemcpyAsync) + timeline of the default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
B="python bench.py --steps 20 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline"
for rep in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel-upload', d['value'], d['ms_per_step'], d['timed_blocks']['Mevents_per_s'])"
  V2E_AMD_CTL_UPLOAD_MEMCPY=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('memcpy-upload', d['value'], d['ms_per_step'], d['timed_blocks']['Mevents_per_s'])"
done > $O/ab_upload.txt 2>&1
cat $O/ab_upload.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p5_kt -- python $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-extras --no-cpu-baseline > $O/p5_kt.log 2>&1
cd $R
python scripts/dump_timeline.py $O/p5_kt 0.35 3000 > $O/p5_timeline_b.txt 2>&1
rm -rf $O/p5_kt
timeout 300 python -m pytest tests/test_emulator_bench_paths_gpu.py -x -q 2>&1 | tail -3
