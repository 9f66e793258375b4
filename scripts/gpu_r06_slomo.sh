#!/bin/bash
# round 6, SloMo per-layer: gpu_r06_slomo.sh <variant> [math]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
V=${1:-0}; M=${2:-fp16x2}
V2E_AMD_S3_VARIANT=$V V2E_AMD_CONV_MATH=$M timeout 300 rocprofv3 --kernel-trace --stats -d $O/sl_$V -- python $R/scripts/slomo_layers.py 80 > $O/sl_$V.log 2>&1
(cd $R; python scripts/parse_layers.py $O/sl_$V 80) > $O/sl_${V}_${M}_layers.txt 2>&1
rm -rf $O/sl_$V
grep -E "conv1 |conv2 |up5|conv3|conv total|other|forward wall" $O/sl_${V}_${M}_layers.txt
