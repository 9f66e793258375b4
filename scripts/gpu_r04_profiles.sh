#!/bin/bash
# round-4 profile artefacts (run on the MI355X box through gpurun; summaries land in gpurun_out/, scripts/make_profiles_r04.py
# assembles profiles/r04_*.txt from them):
#   emulator: kernel trace + chain launch timeline of the bench command, FETCH_SIZE / WRITE_SIZE passes (separate, as the MI355X
#             guide prescribes), SQ counters of the HEADLINE workload alone (instructions per frame over all its kernels) and of
#             the three benchmark workloads
#   SloMo:    per-layer table of the interpolation UNet (80 samples) in the default two-float16-piece math and the exact bf16
#             split, FETCH_SIZE / WRITE_SIZE of one forward in both (THIS round's kernels: operand scaling, range slots)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline"
SQC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p4_kt -- $BENCH > $O/p4_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p4_$c -- $BENCH > $O/p4_$c.log 2>&1; done
timeout 300 rocprofv3 --pmc $SQC --kernel-trace -d $O/p4_sqh -- python $R/scripts/emu_workloads.py headline > $O/p4_sqh.log 2>&1
timeout 400 rocprofv3 --pmc $SQC --kernel-trace -d $O/p4_sq -- python $R/scripts/emu_workloads.py batched hd > $O/p4_sq.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p4_kt/*/*.db | head -1) $O/p4_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p4_$c $c > $O/p4_$c.txt 2>&1; done
python profiles/summarize_rocprof_sq.py $O/p4_sqh k_ > $O/p4_sqh.txt 2>&1
python profiles/summarize_rocprof_sq.py $O/p4_sq k_ > $O/p4_sq.txt 2>&1
python scripts/kernel_timeline.py $O/p4_kt k_chain > $O/p4_kt_timeline.txt 2>&1
python scripts/trace_window.py $O/p4_kt k_c 190 > $O/p4_kt_window.txt 2>&1
grep -h "headline:" $O/p4_sqh.log > $O/p4_sqh_frames.txt
rm -rf $O/p4_kt $O/p4_FETCH_SIZE $O/p4_WRITE_SIZE $O/p4_sq $O/p4_sqh
# ---- SloMo
cd /tmp
for m in fp16x2 bf16x3; do
  V2E_AMD_CONV_MATH=$m timeout 300 rocprofv3 --kernel-trace --stats -d $O/p4_slomo_$m -- python $R/scripts/slomo_layers.py 80 > $O/p4_slomo_$m.log 2>&1
  (cd $R; python scripts/parse_layers.py $O/p4_slomo_$m 80) > $O/p4_slomo_${m}_layers.txt 2>&1
  rm -rf $O/p4_slomo_$m
  for c in FETCH_SIZE WRITE_SIZE; do
    V2E_AMD_CONV_MATH=$m timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p4q_$m$c -- python $R/scripts/slomo_layers.py 80 > $O/p4q_$m$c.log 2>&1
    (cd $R; python profiles/summarize_rocprof_pmc.py $O/p4q_$m$c $c 40) > $O/p4_slomo_${m}_$c.txt 2>&1
    rm -rf $O/p4q_$m$c
  done
done
cd $R
head -8 $O/p4_kt.txt | cut -c1-150; head -3 $O/p4_kt_timeline.txt; tail -3 $O/p4_slomo_fp16x2_layers.txt; tail -2 $O/p4_slomo_fp16x2_FETCH_SIZE.txt
