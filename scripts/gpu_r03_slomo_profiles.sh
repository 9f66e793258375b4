#!/bin/bash
# round-3 profile artefacts of the SloMo kernels: per-layer table of the interpolation UNet (rocprofv3 kernel trace), the
# ablation + step timeline of the pipelined 3x3 kernel, the bare matrix-pipe microbenchmark
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p3_slomo -- python $R/scripts/slomo_layers.py 80 > $O/p3_slomo.log 2>&1
cd $R
python scripts/parse_layers.py $O/p3_slomo 80 > $O/p3_slomo_layers.txt 2>&1
rm -rf $O/p3_slomo
cd /tmp
V2E_AMD_CONV_MATH=fp16x2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p3_slomo_h2 -- python $R/scripts/slomo_layers.py 80 > $O/p3_slomo_h2.log 2>&1
cd $R
python scripts/parse_layers.py $O/p3_slomo_h2 80 > $O/p3_slomo_h2_layers.txt 2>&1
rm -rf $O/p3_slomo_h2
python scripts/slomo_precision.py > $O/p3_slomo_precision.txt 2>&1
timeout 120 ./scripts/ubench_mfma > $O/r03_mfma_bare.txt 2>&1
{
export S3P_TIMELINE=1
for shape in "3 256 128 80 64 80" "3 128 64 80 128 160" "3 512 256 80 32 40"; do
  echo "== layer $shape (ks cin cout n h w)"
  V2E_AMD_S3_VARIANT=12 ./scripts/conv_s3_check $shape 2>&1 | python3 -c "
import sys,re
t=sys.stdin.read(); m=re.search(r's3\s+([\d.]+) us\s+([\d.]+) TF',t)
print('   k_conv_s3  (32 x 64 tiles, two workgroups per CU): %s us %s TF' % (m.group(1), m.group(2)))"
  DBGS="0"; [ "$shape" = "3 256 128 80 64 80" ] && DBGS="0 1 4 8"   # the ablation modes are built for the 16-wide tile only
  for d in $DBGS; do
  V2E_AMD_S3P_DBG=$d V2E_AMD_S3_VARIANT=11 ./scripts/conv_s3_check $shape 2>&1 | python3 -c "
import sys,re
t=sys.stdin.read()
m=re.search(r's3\s+([\d.]+) us\s+([\d.]+) TF',t); n=re.search(r'(\d+) shader clocks per step, shader clock (\d+) MHz',t)
print('   k_conv_s3p dbg %s: %s us %s TF | %s clocks/step @ %s MHz' % ('$d', m.group(1), m.group(2), n.group(1) if n else '-', n.group(2) if n else '-'))"
  done
done
} > $O/p3_s3p_ablation.txt 2>&1
tail -12 $O/p3_slomo_layers.txt | cut -c1-150; cat $O/p3_s3p_ablation.txt
