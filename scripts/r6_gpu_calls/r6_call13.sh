#!/usr/bin/env bash
# round 6, call 13: the split-operand kernels on planes (gemm_f32xp.hip) against the fp32-activation kernels -- equality + time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python tools/f32xp_kbench.py > gpurun_out/r6/call13_f32xp_kbench.txt 2>&1
echo rc=$?
cat gpurun_out/r6/call13_f32xp_kbench.txt | tail -40
