#!/usr/bin/env bash
# round 6, call 33: where the -m gpu suite's wall time goes (every test's duration)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -q --durations=0 --durations-min=0.05 ) > gpurun_out/r6/call33_pytest_durations_all.txt 2>&1
tail -3 gpurun_out/r6/call33_pytest_durations_all.txt
