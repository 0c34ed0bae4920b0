#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for d in 0 1 2 3; do DH_F32X_DIAG=$d timeout 300 python tools/_diag_conv.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6/call19_diag.txt
cat gpurun_out/r6/call19_diag.txt
