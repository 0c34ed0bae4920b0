#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_decode_layers_gpu.py -q > gpurun_out/r6/call06_tests.txt 2>&1; tail -5 gpurun_out/r6/call06_tests.txt
for v in 0 1 0 1; do
  DH_DECODE_LAYERS=$v timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --quick --schedule sequential 2> gpurun_out/r6/call06_c3_$v.err | tail -1 > gpurun_out/r6/call06_c3_$v.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r6/call06_c3_$v.json"))
    print("decode_layers=$v  C3 ms/step", round(d["ms_per_step"], 3), {k: v for k, v in list(d["kernel_breakdown_ms_per_step"].items())[:8]})
except Exception as e:
    print("decode_layers=$v failed", e); print(open("gpurun_out/r6/call06_c3_$v.err").read()[-1500:])
PY
done
