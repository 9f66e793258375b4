#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every emulator kernel (24 kernels listed: the pull writer's reads and k_ctot's ballot writes are small)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do timeout 120 rocprofv3 --pmc $c --kernel-trace -d $O/p5_$c -- $BENCH < /dev/null > $O/p5_$c.log 2>&1; done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do timeout 60 python profiles/summarize_rocprof_pmc.py $O/p5_$c $c 24 < /dev/null > $O/p5_$c.txt 2>&1; done
rm -rf $O/p5_FETCH_SIZE $O/p5_WRITE_SIZE
grep "k_c\|k_ahead\|k_chain" $O/p5_FETCH_SIZE.txt $O/p5_WRITE_SIZE.txt | cut -c1-160
