#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
python bench.py --steps 50 --warmup 5 --blocks 3 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
V2E_AMD_CHAIN_M=1 python bench.py --steps 50 --warmup 5 --blocks 3 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench m1', d['value'], d['ms_per_step'])"
V2E_AMD_CHAIN_RING=4 python bench.py --steps 50 --warmup 5 --blocks 3 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ring4', d['value'], d['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/q_kt -- python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline > $O/q_kt.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/q_kt/*/*.db | head -1) $O/r03g_kt.txt > /dev/null
python scripts/kernel_timeline.py $O/q_kt k_chain > $O/r03g_kt_timeline.txt 2>&1
python scripts/trace_window.py $O/q_kt k_c 120 > $O/r03g_kt_window.txt 2>&1
rm -rf $O/q_kt
head -9 $O/r03g_kt.txt | cut -c1-130; head -1 $O/r03g_kt_timeline.txt; sed -n 1,34p $O/r03g_kt_window.txt
