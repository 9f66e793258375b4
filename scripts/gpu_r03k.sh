#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
timeout 120 ./scripts/ubench_mfma > $O/r03_mfma_bare.txt 2>&1; cat $O/r03_mfma_bare.txt
for v in 0 8 9 10 6; do TAG=variant$v V2E_AMD_S3_VARIANT=$v python scripts/slomo_time.py 80 2>&1 | grep "n=80"; done
V2E_AMD_S3_VARIANT=8 timeout 600 python -m pytest tests/test_slomo_gpu.py -x -q -m gpu 2>&1 | tail -3
