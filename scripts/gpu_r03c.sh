#!/bin/bash
# round 3, call C: emission tables on the k_ahead stream, k_ctot 4 frames per workgroup, chain record fetch reordered; A/B switches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py tests/test_sinks.py -q -x > $O/r03c_pytest.log 2>&1; echo "pytest rc $?" >> $O/r03c_pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc " $O/r03c_pytest.log | tail -20
TAG=graph_prio3 UG=1 python scripts/chain_ab.py 2>&1 | tail -1
TAG=plain_prio3 UG=0 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_prio0 UG=1 V2E_AMD_CHAIN_PRIO=0 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_noemit UG=1 V2E_AMD_CHAIN_NO_EMIT=1 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_tabs_on_side UG=1 V2E_AMD_TABLES_ON_SIDE=1 python scripts/chain_ab.py 2>&1 | tail -1
timeout 300 python scripts/chain_timeline.py > $O/r03c_timeline.txt 2>&1; tail -18 $O/r03c_timeline.txt
timeout 600 python scripts/emu_workloads.py batched hd 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/q_kt -- python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/q_kt.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/q_kt/*/*.db | head -1) $O/r03c_kt_headline.txt > /dev/null
python scripts/kernel_timeline.py $O/q_kt k_chain > $O/r03c_kt_timeline.txt 2>&1
python scripts/trace_window.py $O/q_kt k_c 190 > $O/r03c_kt_window.txt 2>&1
rm -rf $O/q_kt
head -9 $O/r03c_kt_headline.txt; head -1 $O/r03c_kt_timeline.txt; sed -n 1,45p $O/r03c_kt_window.txt
