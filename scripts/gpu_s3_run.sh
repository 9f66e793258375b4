#!/bin/bash
# dev: split-bf16 conv: layer harness (tile variants), SloMo parity tests, SloMo bench in both conv maths
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in ${VARIANTS:-0 7}; do echo "== variant $v"; S3_N=16 V2E_AMD_S3_VARIANT=$v timeout 100 scripts/conv_s3_check 2>&1 | cut -c1-95,150-200 | tail -9; done
timeout 900 python -m pytest tests/test_slomo_gpu.py -x -q > $O/s3_pytest.log 2>&1; tail -5 $O/s3_pytest.log
for m in bf16x3 f32; do V2E_AMD_CONV_MATH=$m python - <<'PY'
import os, json, torch
from v2e_amd.benchutil import slomo_bench
r = slomo_bench(torch.device("cuda"))
print(os.environ["V2E_AMD_CONV_MATH"], json.dumps({k: r[k] for k in r if k != "config"})[:600])
PY
done
