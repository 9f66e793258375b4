#!/bin/bash
# round-2 first GPU call: parity tests, SQ counters of the emulator kernels as they stand, SloMo per-layer trace at 80 samples
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02a_pytest.log 2>&1; echo "pytest rc $?" >> $O/r02a_pytest.log
tail -5 $O/r02a_pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $O/r02a_sq -- python $R/scripts/emu_workloads.py > $O/r02a_sq.log 2>&1
tail -3 $O/r02a_sq.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02a_slomo80 -- python $R/scripts/slomo_layers.py 80 > $O/r02a_slomo80.log 2>&1
cd $R
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU; do python profiles/summarize_rocprof_pmc.py $O/r02a_sq $c; done > $O/r02a_sq.txt 2>&1
python scripts/parse_layers.py $O/r02a_slomo80 80 > $O/r02a_slomo80_layers.txt 2>&1
tail -30 $O/r02a_slomo80_layers.txt
rm -rf $O/r02a_slomo80 # raw db is large
ls -la $O/r02a_sq/*/ | head
