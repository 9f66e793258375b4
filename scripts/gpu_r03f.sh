#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/q_hd -- python $R/scripts/emu_workloads.py hd > $O/q_hd.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/q_b -- python $R/scripts/emu_workloads.py batched > $O/q_b.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/q_hd/*/*.db | head -1) $O/r03f_kt_hd.txt > /dev/null
python profiles/summarize_rocprof_db.py $(ls $O/q_b/*/*.db | head -1) $O/r03f_kt_batched.txt > /dev/null
python scripts/trace_window.py $O/q_hd k_c 60 > $O/r03f_window_hd.txt 2>&1
python scripts/trace_window.py $O/q_b k_c 100 > $O/r03f_window_batched.txt 2>&1
rm -rf $O/q_hd $O/q_b
tail -1 $O/q_hd.log; head -9 $O/r03f_kt_hd.txt | cut -c1-150; sed -n 1,24p $O/r03f_window_hd.txt
tail -1 $O/q_b.log; head -9 $O/r03f_kt_batched.txt | cut -c1-150; sed -n 1,24p $O/r03f_window_batched.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'kernel_us', r['avg_kernel_us'], 'frac', r['frac'], 'period', r['launch_period_us'], d['timed_blocks']['Mevents_per_s'])"
