#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 400 python scripts/slp_repro/repro3.py ) > gpurun_out/slp_repro3.log 2>&1
echo "repro3 rc=$?" >> gpurun_out/slp_repro3.log
tail -40 gpurun_out/slp_repro3.log
