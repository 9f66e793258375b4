#!/bin/bash
# round 3, call D: how the runtime schedules the run: plain launches on three streams vs the hipGraph, graph queue switches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=graph UG=1 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_q8 UG=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_nopktcap UG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python scripts/chain_ab.py 2>&1 | tail -1
TAG=graph_q8_nopktcap UG=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python scripts/chain_ab.py 2>&1 | tail -1
TAG=plain UG=0 python scripts/chain_ab.py 2>&1 | tail -1
TAG=plain_hwq8 UG=0 GPU_MAX_HW_QUEUES=8 python scripts/chain_ab.py 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
V2E_AMD_BENCH_UG=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/q_kt -- python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/q_kt.log 2>&1
cd $R
python scripts/kernel_timeline.py $O/q_kt k_chain > $O/r03d_kt_timeline_plain.txt 2>&1
python scripts/trace_window.py $O/q_kt k_c 190 > $O/r03d_kt_window_plain.txt 2>&1
rm -rf $O/q_kt
head -1 $O/r03d_kt_timeline_plain.txt; sed -n 1,50p $O/r03d_kt_window_plain.txt
