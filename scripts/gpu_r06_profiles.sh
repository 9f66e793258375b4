#!/bin/bash
# round-6 profile artefacts (run on the MI355X box through gpurun; summaries land in gpurun_out/, scripts/make_profiles_r06.py
# assembles profiles/r06_*.txt from them):
#   emulator: kernel trace + chain launch timeline + full step timeline of the bench command, FETCH_SIZE / WRITE_SIZE passes (separate,
#             as the MI355X guide prescribes), SQ counters of the HEADLINE workload alone and of the batched / 1280x720 workloads
#   SloMo:    per-layer table of the interpolation UNet (80 samples) for all three conv maths, FETCH_SIZE / WRITE_SIZE of one forward
#             for all three, and of the HD shape (2 samples at 1280x704: what bench.py's slomo_hd leg runs) in the default math
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 12 --warmup 3 --blocks 1 --no-extras --no-cpu-baseline --no-roofline-rerun"
SQC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p6_kt -- $BENCH > $O/p6_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p6_$c -- $BENCH > $O/p6_$c.log 2>&1; done
timeout 300 rocprofv3 --pmc $SQC --kernel-trace -d $O/p6_sqh -- python $R/scripts/emu_workloads.py headline > $O/p6_sqh.log 2>&1
timeout 400 rocprofv3 --pmc $SQC --kernel-trace -d $O/p6_sq -- python $R/scripts/emu_workloads.py batched hd > $O/p6_sq.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p6_kt/*/*.db | head -1) $O/p6_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p6_$c $c > $O/p6_$c.txt 2>&1; done
python profiles/summarize_rocprof_sq.py $O/p6_sqh k_ > $O/p6_sqh.txt 2>&1
python profiles/summarize_rocprof_sq.py $O/p6_sq k_ > $O/p6_sq.txt 2>&1
python scripts/kernel_timeline.py $O/p6_kt k_chain > $O/p6_kt_timeline.txt 2>&1
python scripts/dump_timeline.py $O/p6_kt 0.12 130 > $O/p6_kt_step.txt 2>&1
grep -h "headline:" $O/p6_sqh.log > $O/p6_sqh_frames.txt
rm -rf $O/p6_kt $O/p6_FETCH_SIZE $O/p6_WRITE_SIZE $O/p6_sq $O/p6_sqh
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p6_hd -- python $R/scripts/emu_workloads.py hd > $O/p6_hd.log 2>&1
(cd $R; python profiles/summarize_rocprof_db.py $(ls $O/p6_hd/*/*.db | head -1) $O/p6_hd.txt > /dev/null); rm -rf $O/p6_hd
# ---- SloMo
cd /tmp
for m in fp16x2 bf16x3 f32; do
  V2E_AMD_CONV_MATH=$m timeout 300 rocprofv3 --kernel-trace --stats -d $O/p6_slomo_$m -- python $R/scripts/slomo_layers.py 80 > $O/p6_slomo_$m.log 2>&1
  (cd $R; python scripts/parse_layers.py $O/p6_slomo_$m 80) > $O/p6_slomo_${m}_layers.txt 2>&1
  rm -rf $O/p6_slomo_$m
  for c in FETCH_SIZE WRITE_SIZE; do
    V2E_AMD_CONV_MATH=$m timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p6q_$m$c -- python $R/scripts/slomo_layers.py 80 > $O/p6q_$m$c.log 2>&1
    (cd $R; python profiles/summarize_rocprof_pmc.py $O/p6q_$m$c $c 40) > $O/p6_slomo_${m}_$c.txt 2>&1
    rm -rf $O/p6q_$m$c
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  V2E_AMD_CONV_MATH=fp16x2 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p6q_hd$c -- python $R/scripts/slomo_layers.py 2 704 1280 > $O/p6q_hd$c.log 2>&1
  (cd $R; python profiles/summarize_rocprof_pmc.py $O/p6q_hd$c $c 40) > $O/p6_slomo_hd_$c.txt 2>&1
  rm -rf $O/p6q_hd$c
done
cd $R
head -8 $O/p6_kt.txt | cut -c1-150; head -3 $O/p6_kt_timeline.txt; tail -3 $O/p6_slomo_fp16x2_layers.txt; tail -2 $O/p6_slomo_f32_FETCH_SIZE.txt; tail -2 $O/p6_slomo_hd_FETCH_SIZE.txt
