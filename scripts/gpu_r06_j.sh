#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for SP in 1 0 1 0; do
echo "== split first k_ahead batch: $SP"; V2E_AMD_PIPE_SPLIT0=$SP timeout 100 python scripts/chain_stamps.py 40 0 2>&1 | grep -v amdgpu.ids | head -7 | grep "plain\|per run\|gap before"
done
