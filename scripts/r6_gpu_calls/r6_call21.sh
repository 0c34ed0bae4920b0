#!/usr/bin/env bash
# round 6, call 21: the streaming 1 x 1 kernel of the fp32 split-operand trunk -- equality tests, per-layer times, whole steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_f32x_gpu.py -q -x -k "streaming" 2>&1 | tail -8
timeout 600 python tools/f32x_conv1x1_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/call21_conv1x1_bench.txt
