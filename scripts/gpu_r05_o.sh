#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/p5_sb -- python $R/scripts/slomo_batch_trace.py > $O/p5_sb.log 2>&1
cd $R
python scripts/dump_timeline.py $O/p5_sb 0.3 400 > $O/p5_slomo_batch_timeline.txt 2>&1
rm -rf $O/p5_sb
grep "ms per batch" $O/p5_sb.log
for i in 1 2 3; do python scripts/slomo_batch_trace.py 2>/dev/null | tail -1; done; V2E_AMD_CONV_MATH=bf16x3 python scripts/slomo_batch_trace.py 2>/dev/null | tail -1; timeout 300 python -m pytest tests/test_slomo_gpu.py -q -k "ahead or benchmark_shape or class_writes or auto_upsample" 2>&1 | tail -2
