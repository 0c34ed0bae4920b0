#!/usr/bin/env bash
# round 6, call 17: gemm_f32xp with two wave groups a phase apart (variant 2) against the single-phase loop (variant 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for v in 2 0 2; do
  DH_XP_VARIANT=$v timeout 600 python tools/f32xp_kbench.py > gpurun_out/r6/call17_f32xp_kbench_v$v.txt 2>&1
  echo variant $v rc=$?
  grep -v amdgpu.ids gpurun_out/r6/call17_f32xp_kbench_v$v.txt | tail -24
done
