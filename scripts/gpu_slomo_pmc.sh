#!/bin/bash
# HBM traffic of one interpolation-UNet forward (80 samples), all kernels: separate FETCH_SIZE / WRITE_SIZE passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in bf16x3 f32; do for c in FETCH_SIZE WRITE_SIZE; do
  V2E_AMD_CONV_MATH=$m timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/q_$m$c -- python $R/scripts/slomo_layers.py 80 > $O/q_$m$c.log 2>&1
  (cd $R; python profiles/summarize_rocprof_pmc.py $O/q_$m$c $c 40) > $O/q_slomo_${m}_$c.txt 2>&1
  rm -rf $O/q_$m$c
done; done
cd $R; tail -3 $O/q_slomo_bf16x3_FETCH_SIZE.txt
timeout 600 python -m pytest tests/test_emulator_gpu.py -q -x -k "overflow or photoreceptor" 2>&1 | tail -3
