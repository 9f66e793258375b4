#!/bin/bash
# pull vs push: per-kernel times (rocprofv3) and the HD / batched legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
for pull in 0 1; do
  echo "--- pull=$pull"
  V2E_AMD_EMIT_PULL=$pull timeout 400 python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'hd', d['hd_noisy']['value'], d['hd_noisy'].get('hbm_frac'), 'batched', d['batched']['value'])"
done
cd /tmp
for pull in 0 1; do
  rm -rf /tmp/prof$pull
  V2E_AMD_EMIT_PULL=$pull timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$pull -o p -- python $R/bench.py --steps 10 --warmup 3 --blocks 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/prof$pull -name '*kernel_stats.csv' | head -1)
  echo "--- stats pull=$pull"; head -12 $f | cut -c1-150
  cp $f $O/pull${pull}_kernel_stats.csv
done
