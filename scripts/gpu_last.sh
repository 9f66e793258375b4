#!/bin/bash
# last GPU call of the round: the default bench line and the emulator profile passes, nothing else
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 100 python bench.py > $O/full_bench.log 2>&1; grep '^{' $O/full_bench.log | tail -1 > $O/full_bench.json
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline"
timeout 40 rocprofv3 --kernel-trace --stats -d $O/p_kt -- $BENCH > $O/p_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 40 rocprofv3 --pmc $c --kernel-trace -d $O/p_$c -- $BENCH > $O/p_$c.log 2>&1; done
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p_kt/*/*.db | head -1) $O/p_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p_$c $c > $O/p_$c.txt 2>&1; done
python scripts/kernel_timeline.py $O/p_kt k_chain > $O/p_kt_timeline.txt 2>&1
rm -rf $O/p_kt $O/p_FETCH_SIZE $O/p_WRITE_SIZE
python -c "
import json; d=json.load(open('gpurun_out/full_bench.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac']); print('hd', d['hd_noisy']['value'], d['hd_noisy']['hbm_frac'], 'batched', d['batched']['value'], 'slomo', d['slomo']['value'], 'e2e', d['end_to_end']['interpolated_frames_per_s'])"
head -2 $O/p_kt_timeline.txt
