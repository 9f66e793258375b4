#!/bin/bash
# round-3 profile artefacts of the emulator: kernel trace + launch timeline of the bench command, HBM PMC passes (separate, as
# the MI355X guide prescribes), SQ counters of the three emulator workloads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p3_kt -- $BENCH > $O/p3_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p3_$c -- $BENCH > $O/p3_$c.log 2>&1; done
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $O/p3_sq -- python $R/scripts/emu_workloads.py > $O/p3_sq.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p3_kt/*/*.db | head -1) $O/p3_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p3_$c $c > $O/p3_$c.txt 2>&1; done
python profiles/summarize_rocprof_sq.py $O/p3_sq k_ > $O/p3_sq.txt 2>&1
python scripts/kernel_timeline.py $O/p3_kt k_chain > $O/p3_kt_timeline.txt 2>&1
python scripts/trace_window.py $O/p3_kt k_c 190 > $O/p3_kt_window.txt 2>&1
rm -rf $O/p3_kt $O/p3_FETCH_SIZE $O/p3_WRITE_SIZE $O/p3_sq
head -10 $O/p3_kt.txt | cut -c1-160; cat $O/p3_kt_timeline.txt | head -3; head -12 $O/p3_FETCH_SIZE.txt | cut -c1-130; head -12 $O/p3_WRITE_SIZE.txt | cut -c1-130; head -30 $O/p3_sq.txt | cut -c1-200
