#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
python scripts/frame_api_profile.py tape 2>&1 | grep -v amdgpu.ids | head -34
