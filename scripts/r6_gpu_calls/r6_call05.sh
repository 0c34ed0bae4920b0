#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 300 python scratch/dl_debug.py > gpurun_out/r6/call05_dbg.txt 2>&1; tail -12 gpurun_out/r6/call05_dbg.txt
echo "--- forced NIT=5 from t=0"
DH_DL_DEBUG=1 timeout 300 python scratch/dl_debug.py > gpurun_out/r6/call05_dbg_nit5.txt 2>&1; tail -12 gpurun_out/r6/call05_dbg_nit5.txt
