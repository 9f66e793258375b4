#!/bin/bash
# round-5 profile artefacts, EMULATOR part only (after the pull event writer) (run on the MI355X box through gpurun; summaries land in gpurun_out/, scripts/make_profiles_r05.py
# assembles profiles/r05_*.txt from them):
#   emulator: kernel trace + chain launch timeline + full step timeline of the bench command, FETCH_SIZE / WRITE_SIZE passes (separate,
#             as the MI355X guide prescribes), SQ counters of the HEADLINE workload alone and of the batched / 1280x720 workloads
#   SloMo:    per-layer table of the interpolation UNet (80 samples) for all three conv maths, FETCH_SIZE / WRITE_SIZE of one forward
#             for all three, and of the HD shape (2 samples at 1280x704: what bench.py's slomo_hd leg runs) in the default math
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline"
SQC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p5_kt -- $BENCH > $O/p5_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p5_$c -- $BENCH > $O/p5_$c.log 2>&1; done
timeout 300 rocprofv3 --pmc $SQC --kernel-trace -d $O/p5_sqh -- python $R/scripts/emu_workloads.py headline > $O/p5_sqh.log 2>&1
timeout 400 rocprofv3 --pmc $SQC --kernel-trace -d $O/p5_sq -- python $R/scripts/emu_workloads.py batched hd > $O/p5_sq.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p5_kt/*/*.db | head -1) $O/p5_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p5_$c $c 24 > $O/p5_$c.txt 2>&1; done
python profiles/summarize_rocprof_sq.py $O/p5_sqh k_ > $O/p5_sqh.txt 2>&1
python profiles/summarize_rocprof_sq.py $O/p5_sq k_ > $O/p5_sq.txt 2>&1
python scripts/kernel_timeline.py $O/p5_kt k_chain > $O/p5_kt_timeline.txt 2>&1
python scripts/dump_timeline.py $O/p5_kt 0.3 120 > $O/p5_kt_step.txt 2>&1
grep -h "headline:" $O/p5_sqh.log > $O/p5_sqh_frames.txt
rm -rf $O/p5_kt $O/p5_FETCH_SIZE $O/p5_WRITE_SIZE $O/p5_sq $O/p5_sqh
