#!/bin/bash
# kernel trace of the 1280x720 noisy workload with the pull writer
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/hd_kt -- python $R/scripts/emu_workloads.py hd < /dev/null > $O/hd_kt.log 2>&1
cd $R
db=$(ls $O/hd_kt/*/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 100 python profiles/summarize_rocprof_db.py $db $O/hd_kt.txt < /dev/null > /dev/null 2>&1; head -12 $O/hd_kt.txt | cut -c1-150; fi
rm -rf $O/hd_kt
grep "^hd:" $O/hd_kt.log | cut -c1-200
