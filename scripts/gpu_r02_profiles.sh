#!/bin/bash
# round-2 profile artefacts: kernel traces, HBM PMC passes (separate, as the MI355X guide prescribes), SQ counters, SloMo layers + counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_kt -- $BENCH > $O/p_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p_$c -- $BENCH > $O/p_$c.log 2>&1; done
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $O/p_sq -- python $R/scripts/emu_workloads.py > $O/p_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_slomo -- python $R/scripts/slomo_layers.py 80 > $O/p_slomo.log 2>&1
V2E_AMD_CONV_MATH=f32 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_slomo32 -- python $R/scripts/slomo_layers.py 80 > $O/p_slomo32.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p_slomo_$c -- python $R/scripts/slomo_layers.py 80 > $O/p_slomo_$c.log 2>&1; done
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|MOPS" | head -20 > $O/p_counters_mfma.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --kernel-trace -d $O/p_slomo_mfma -- python $R/scripts/slomo_layers.py 80 > $O/p_slomo_mfma.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/p_kt/*/*.db | head -1) $O/p_kt.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python profiles/summarize_rocprof_pmc.py $O/p_$c $c > $O/p_$c.txt 2>&1; python profiles/summarize_rocprof_pmc.py $O/p_slomo_$c $c > $O/p_slomo_$c.txt 2>&1; done
python profiles/summarize_rocprof_sq.py $O/p_sq k_ > $O/p_sq.txt 2>&1
python scripts/parse_layers.py $O/p_slomo 80 > $O/p_slomo_layers.txt 2>&1
python scripts/parse_layers.py $O/p_slomo32 80 > $O/p_slomo32_layers.txt 2>&1
for c in SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA; do python profiles/summarize_rocprof_pmc.py $O/p_slomo_mfma $c; done > $O/p_slomo_mfma.txt 2>&1
python scripts/kernel_timeline.py $O/p_kt k_chain > $O/p_kt_timeline.txt 2>&1
rm -rf $O/p_slomo32 $O/p_kt $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_sq $O/p_slomo $O/p_slomo_FETCH_SIZE $O/p_slomo_WRITE_SIZE $O/p_slomo_mfma
head -8 $O/p_kt.txt; tail -8 $O/p_slomo_layers.txt
