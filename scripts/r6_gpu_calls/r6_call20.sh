#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for d in 0 4 0 4; do DH_F32X_DIAG=$d timeout 300 python tools/_diag_conv.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6/call20_diag.txt
cat gpurun_out/r6/call20_diag.txt
timeout 600 python -m pytest tests/test_f32x_gpu.py -q -x -k "conv or linear_f32x_against" 2>&1 | tail -3
