#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for FB in 0 1 0 1 0 1; do
echo "== fast boundary $FB"; V2E_AMD_CHAIN_FASTB=$FB timeout 100 python scripts/chain_stamps.py 40 0 2>&1 | grep -v amdgpu.ids | head -7 | grep "plain\|per run\|full launches"
done
