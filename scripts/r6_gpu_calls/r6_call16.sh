#!/usr/bin/env bash
# round 6, call 16: conflict-free fragment swizzle in the f32x tile kernels (kbench old / planes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python tools/f32xp_kbench.py > gpurun_out/r6/call16_f32xp_kbench.txt 2>&1
echo rc=$?
grep -v amdgpu.ids gpurun_out/r6/call16_f32xp_kbench.txt | tail -24
