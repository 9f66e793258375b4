#!/bin/bash
# kernel trace (+ optional SQ counters) of the emulator workloads: gpu_prof.sh <tag> <workloads...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -- python $R/scripts/emu_workloads.py "$@" > $O/${TAG}_kt.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/${TAG}_kt/*/*.db | head -1) $O/${TAG}_kt.txt | head -24
grep -E "headline|batched|hd" $O/${TAG}_kt.log | cut -c1-300
if [ -n "$SQ" ]; then
  cd /tmp
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $O/${TAG}_sq -- python $R/scripts/emu_workloads.py "$@" > $O/${TAG}_sq.log 2>&1
  cd $R
  python profiles/summarize_rocprof_sq.py $O/${TAG}_sq k_ > $O/${TAG}_sq.txt; cat $O/${TAG}_sq.txt
  rm -rf $O/${TAG}_sq
fi
rm -rf $O/${TAG}_kt
