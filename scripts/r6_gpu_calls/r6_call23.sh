#!/usr/bin/env bash
# round 6, call 23: the LSTM step of the f32x path on planes -- f32x suite, LSTM model tests, whole steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time timeout 1500 python -m pytest tests/test_f32x_gpu.py -q -x ) > gpurun_out/r6/call23_pytest_f32x.txt 2>&1
tail -8 gpurun_out/r6/call23_pytest_f32x.txt
timeout 900 python tools/f32x_bench.py c2 > gpurun_out/r6/call23_f32x_bench.txt 2>&1
python - <<'PY'
import json
t = open("gpurun_out/r6/call23_f32x_bench.txt").read()
d = json.loads(t[t.index("{"):])
for k, v in d.items():
    print(k, v["ms_per_step"])
for kk, vv in d["c2_split1_planes"]["event_timed_ms"].items(): print("    ", kk, vv)
PY
