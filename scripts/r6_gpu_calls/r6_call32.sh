#!/usr/bin/env bash
# round 6, call 32: last check of the final tree -- the driver's three commands (pytest -m gpu, smoke(), bench.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r6/call32_pytest.txt 2>&1
tail -6 gpurun_out/r6/call32_pytest.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6
( time python bench.py > gpurun_out/r6/call32_bench_default.json 2> gpurun_out/r6/call32_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call32_bench_default.json"))
print({k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")})
print(d["config"]); print(d["roofline"]["frac"], d["cpu_baseline"]["value"], d["parity_grade_path"]["ms_per_step"], d["c3"]["ms_per_step"], d["c3"]["parity_grade_path"]["ms_per_step"], d.get("value_f16"))
PY
