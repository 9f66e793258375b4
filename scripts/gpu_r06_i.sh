#!/bin/bash
# round 6: pipelined runs without a refractory period (1280x720 noisy) + parity of the no-refractory fixtures through it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -m gpu -q -x < /dev/null > $O/ov_pytest.log 2>&1
tail -3 $O/ov_pytest.log
python - <<'P'
import os, sys
sys.path.insert(0, '.')
import torch
from v2e_amd.benchutil import hd_noisy_emulator_bench
dev = torch.device("cuda", 0)
for ug in ("1", "0", "1", "0"):
    os.environ["V2E_AMD_BENCH_UG"] = ug
    r = hd_noisy_emulator_bench(dev)
    print("hd_noisy use_graph=%s:" % ug, r["value"], r["hbm_frac"], r["pipeline"])
P
