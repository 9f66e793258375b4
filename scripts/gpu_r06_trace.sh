#!/bin/bash
# kernel-trace timeline of the headline bench command for given env settings: gpu_r06_trace.sh <tag> [ENV=VAL ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -- python $R/bench.py --steps 12 --warmup 3 --blocks 1 --no-extras --no-cpu-baseline --no-roofline-rerun > $O/${TAG}_kt.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/${TAG}_kt/*/*.db | head -1) $O/${TAG}_kt.txt > /dev/null
python scripts/kernel_timeline.py $O/${TAG}_kt k_chain > $O/${TAG}_timeline.txt 2>&1
python scripts/dump_timeline.py $O/${TAG}_kt 0.12 140 > $O/${TAG}_step.txt 2>&1
rm -rf $O/${TAG}_kt
head -3 $O/${TAG}_timeline.txt
grep -o '"value": [0-9.]*, "unit": "Mevents/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/${TAG}_kt.log
