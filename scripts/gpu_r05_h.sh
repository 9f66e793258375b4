#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
export V2E_AMD_ZERO_FIRST=1
( python scripts/step_stamps.py; V2E_AMD_BENCH_STAGE_FRAMES=1 python scripts/step_stamps.py; F=608 python scripts/step_stamps.py; F=1216 python scripts/step_stamps.py ) > $O/step_stamps.txt 2>&1
cat $O/step_stamps.txt
