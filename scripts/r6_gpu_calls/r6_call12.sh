#!/usr/bin/env bash
# round 6, call 12: the tree after the hip.py split -- full -m gpu suite with wall time, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -q --durations=30 ) > gpurun_out/r6/call12_pytest.txt 2>&1
tail -45 gpurun_out/r6/call12_pytest.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -8
( time python bench.py > gpurun_out/r6/call12_bench_default.json 2> gpurun_out/r6/call12_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call12_bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_pipelined", "ms_per_step_pipelined", "value_f16", "whole_step_mfma_frac")})
print(d["roofline"])
c3 = d["c3"]
print("c3", c3["ms_per_step"], c3["pipelined"]["ms_per_step"], c3["parity_grade_path"]["ms_per_step"])
PY
