#!/bin/bash
# 1280x720: emission batches of 64 frames (V2E_AMD_CHAIN_M=2) against the default 32
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
for m in 1 2 1 2; do
echo "--- hd chain_m=$m"; V2E_AMD_CHAIN_M=$m timeout 200 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-120
done
