#!/usr/bin/env bash
# round 6, call 15: the planes chain of the f32x path -- its tests, the f32x suite, f32x bench legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time timeout 1200 python -m pytest tests/test_f32x_gpu.py -q -x -k "planes" ) > gpurun_out/r6/call15_pytest_planes.txt 2>&1
tail -30 gpurun_out/r6/call15_pytest_planes.txt
( time timeout 1200 python -m pytest tests/test_f32x_gpu.py -q -k "not planes" ) > gpurun_out/r6/call15_pytest_f32x.txt 2>&1
tail -15 gpurun_out/r6/call15_pytest_f32x.txt
timeout 900 python tools/f32x_bench.py c2 c3 > gpurun_out/r6/call15_f32x_bench.txt 2>&1
tail -12 gpurun_out/r6/call15_f32x_bench.txt
