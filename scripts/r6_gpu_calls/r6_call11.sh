#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python bench.py > gpurun_out/r6/call11_bench_default.json 2> gpurun_out/r6/call11_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call11_bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_pipelined", "ms_per_step_pipelined", "value_f16", "value_hipgraph_replay", "value_host_inclusive", "value_parity_grade")})
c3 = d["c3"]
print("c3", c3["ms_per_step"], c3["pipelined"]["ms_per_step"], c3.get("f16_path"), c3["parity_grade_path"]["ms_per_step"])
print(d.get("f16_path"))
PY
