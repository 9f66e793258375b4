#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
TAG=capture_m1 UG=1 python scripts/chain_ab.py 2>&1 | tail -1
TAG=capture_m2 UG=1 V2E_AMD_CHAIN_M=2 python scripts/chain_ab.py 2>&1 | tail -1
TAG=capture_m2_ring4 UG=1 V2E_AMD_CHAIN_M=2 V2E_AMD_CHAIN_RING=4 python scripts/chain_ab.py 2>&1 | tail -1
TAG=explicit_m2 UG=1 V2E_AMD_CHAIN_M=2 V2E_AMD_GRAPH_EXPLICIT=1 python scripts/chain_ab.py 2>&1 | tail -1
TAG=plain_m2 UG=0 V2E_AMD_CHAIN_M=2 python scripts/chain_ab.py 2>&1 | tail -1
TAG=plain_m2_ring4 UG=0 V2E_AMD_CHAIN_M=2 V2E_AMD_CHAIN_RING=4 python scripts/chain_ab.py 2>&1 | tail -1
for cfg in "V2E_AMD_CHAIN_M=1" "V2E_AMD_CHAIN_M=2" "V2E_AMD_CHAIN_M=2 V2E_AMD_CHAIN_RING=4" "V2E_AMD_CHAIN_M=2 V2E_AMD_BENCH_UG=0" "V2E_AMD_CHAIN_M=2 V2E_AMD_GRAPH_EXPLICIT=1"; do
env $cfg python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench $cfg', d['value'], d['ms_per_step'])"
done
