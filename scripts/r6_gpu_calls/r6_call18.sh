#!/usr/bin/env bash
# round 6, call 18: full -m gpu suite on the planes tree, f32x bench legs, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r6/call18_pytest.txt 2>&1
tail -40 gpurun_out/r6/call18_pytest.txt
timeout 900 python tools/f32x_bench.py c2 c3 > gpurun_out/r6/call18_f32x_bench.txt 2>&1
python - <<'PY'
import json
t = open("gpurun_out/r6/call18_f32x_bench.txt").read()
d = json.loads(t[t.index("{"):])
for k, v in d.items():
    print(k, v["ms_per_step"])
PY
( time python bench.py > gpurun_out/r6/call18_bench_default.json 2> gpurun_out/r6/call18_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call18_bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_pipelined", "value_f16", "value_parity_grade")})
print(d.get("parity_grade_path", {}).get("ms_per_step"), d["c3"]["ms_per_step"], d["c3"]["parity_grade_path"]["ms_per_step"])
PY
