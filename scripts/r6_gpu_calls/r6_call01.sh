#!/usr/bin/env bash
# round 6, call 1: the pruned tree -- full -m gpu suite (wall time), then a quick bench of both workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r6/call01_pytest.txt 2>&1
tail -5 gpurun_out/r6/call01_pytest.txt
python bench.py --steps 10 --warmup 3 --quick > gpurun_out/r6/call01_bench_quick.json 2> gpurun_out/r6/call01_bench_quick.err
tail -c 600 gpurun_out/r6/call01_bench_quick.json
