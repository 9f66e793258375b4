#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 200 python scripts/host_profile_steps.py > $O/host_profile.txt 2>&1
head -60 $O/host_profile.txt
