#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in 0 11; do TAG=variant$v V2E_AMD_S3_VARIANT=$v timeout 300 python scripts/slomo_time.py 80 2>&1 | grep "n=80"; done; done
V2E_AMD_S3_VARIANT=11 timeout 600 python -m pytest tests/test_slomo_gpu.py -x -q -m gpu 2>&1 | tail -2
