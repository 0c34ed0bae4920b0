#!/usr/bin/env bash
# round 6, call 34: the changed full-size tests + suite wall time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r6/call34_pytest.txt 2>&1
tail -22 gpurun_out/r6/call34_pytest.txt
