#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export S3P_TIMELINE=1
V2E_AMD_S3_VARIANT=11 python scripts/slomo_variant_out.py 2>&1 | grep -v amdgpu.ids | head -1
for d in 0 1 4 8; do
V2E_AMD_S3P_DBG=$d V2E_AMD_S3_VARIANT=11 ./scripts/conv_s3_check 3 256 128 80 64 80 2>&1 | python3 -c "
import sys,re
t=sys.stdin.read()
m=re.search(r's3\s+([\d.]+) us\s+([\d.]+) TF',t); n=re.search(r'(\d+) shader clocks per step, shader clock (\d+) MHz',t)
print('dbg %3s: %s us %s TF | %s clocks/step @ %s MHz' % ('$d', m.group(1), m.group(2), n.group(1) if n else '-', n.group(2) if n else '-'))"
done
