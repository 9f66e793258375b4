#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
( for rep in 1 2; do
  for fu in 0 1 3 7 15 31 16 24; do TAG="fuse_up=$fu" V2E_AMD_FUSE_UP=$fu ITERS=6 python scripts/slomo_time.py 80 fp16x2 bf16x3 2>/dev/null; done
  TAG="fuse_pool=1" V2E_AMD_FUSE_POOL=1 ITERS=6 python scripts/slomo_time.py 80 fp16x2 2>/dev/null
done ) > $O/slomo_fuse_ab.txt 2>&1
cat $O/slomo_fuse_ab.txt
