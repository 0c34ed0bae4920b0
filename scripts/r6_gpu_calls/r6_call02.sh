#!/usr/bin/env bash
# round 6, call 2: full -m gpu suite on the pruned tree with the new gates in RECORD mode (their observed values set G18_OBSERVED),
# wall time of the suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time DH_GATE_RECORD=1 python -m pytest tests -m gpu -q -s -k "16bit or f32x or dist_gpu or range_guard or exact_path" ) > gpurun_out/r6/call02_new_tests.txt 2>&1
tail -5 gpurun_out/r6/call02_new_tests.txt
( time DH_GATE_RECORD=1 python -m pytest tests -m gpu -q ) > gpurun_out/r6/call02_pytest.txt 2>&1
tail -8 gpurun_out/r6/call02_pytest.txt
cp gpurun_out/oracle_gate_*.json gpurun_out/r6/ 2>/dev/null
