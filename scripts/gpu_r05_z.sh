#!/bin/bash
# the emulator profile passes again with the final configuration (64-frame batches on the large grid) + a kernel trace of 1280x720 noisy
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 bash scripts/gpu_r05_profiles_emu.sh < /dev/null > $O/profiles_emu.log 2>&1
timeout 300 bash scripts/gpu_r05_v.sh < /dev/null 2>&1 | tail -8
head -8 $O/p5_kt.txt | cut -c1-120
