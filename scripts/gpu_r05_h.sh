#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
( python scripts/step_stamps.py; F=1216 python scripts/step_stamps.py ) > $O/step_stamps.txt 2>&1
cat $O/step_stamps.txt
