#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
timeout 120 ./scripts/ubench_mfma > $O/r03_mfma_bare.txt 2>&1; cat $O/r03_mfma_bare.txt
timeout 600 python -m pytest tests/test_emulator_gpu.py -x -q -m gpu -k "frame or api or single or drop or fast or growth or cap" 2>&1 | tail -4
python scripts/frame_api_rate.py
python scripts/frame_api_profile.py philox 2>&1 | head -40
