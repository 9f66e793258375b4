#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
V2E_AMD_FRAME_TIMING=1 python scripts/frame_api_rate.py 2>&1 | grep -v amdgpu.ids | head -8
V2E_AMD_FRAME_GRAPH=0 V2E_AMD_FRAME_TIMING=1 python scripts/frame_api_rate.py 2>&1 | grep -v amdgpu.ids | head -5
