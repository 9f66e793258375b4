#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_csdvs.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_emulator_gpu.py -x -q -m gpu -k "frame or api or single or drop or fast or growth or cap or scidvs" 2>&1 | tail -4
python scripts/frame_api_rate.py
python scripts/frame_api_profile.py philox 2>&1 | head -16
