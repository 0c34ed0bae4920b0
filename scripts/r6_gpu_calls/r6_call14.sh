#!/usr/bin/env bash
# round 6, call 14: gemm_f32xp variants (0: 8 waves of 64 x 64, MFMA order changed; 1: 4 waves of 128 x 64)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for v in 0 1; do
  DH_XP_VARIANT=$v timeout 600 python tools/f32xp_kbench.py > gpurun_out/r6/call14_f32xp_kbench_v$v.txt 2>&1
  echo variant $v rc=$?
  grep -v amdgpu.ids gpurun_out/r6/call14_f32xp_kbench_v$v.txt | tail -24
done
