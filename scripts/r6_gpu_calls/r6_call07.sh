#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
DH_DL_DEBUG=2 timeout 300 python scratch/dl_stamps.py > gpurun_out/r6/call07_stamps.txt 2>&1; tail -20 gpurun_out/r6/call07_stamps.txt
