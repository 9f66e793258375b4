#!/bin/bash
# round 5, call A: root-cause experiment for the SLP / MFMA concurrency finding + the two-stream standing tests
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 240 python scripts/slp_repro/repro.py ) > gpurun_out/slp_repro.log 2>&1
echo "repro rc=$?" >> gpurun_out/slp_repro.log
( timeout 300 python -m pytest tests/test_concurrency_gpu.py -x -q ) > gpurun_out/concurrency_test.log 2>&1
tail -30 gpurun_out/slp_repro.log
tail -15 gpurun_out/concurrency_test.log
