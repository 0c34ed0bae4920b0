#!/usr/bin/env bash
# round 6, call 8: the full -m gpu suite (wall time + the slowest tests), the degraded-library check of the 16-bit gates
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -q --durations=40 ) > gpurun_out/r6/call08_pytest.txt 2>&1
tail -60 gpurun_out/r6/call08_pytest.txt
DEEPHUMOR_HIP_LIB=$PWD/scratch/degraded/libdegraded.so python -m pytest tests/test_fullsize_gpu.py -q -k "16bit and LSTM" > gpurun_out/r6/call08_degraded_gates.txt 2>&1
tail -6 gpurun_out/r6/call08_degraded_gates.txt
