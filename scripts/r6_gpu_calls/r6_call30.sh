#!/usr/bin/env bash
# round 6, call 30: self-attention with 3 / 4 unrolled iterations for positions 17 - 32 -- kernel tests, model tests, C3 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_bf16_gpu.py -q -x 2>&1 | tail -4
python bench.py --workload c3 --steps 20 --warmup 5 --quick --schedule sequential 2>/dev/null | tail -1 > gpurun_out/r6/call30_c3.json
python bench.py --workload c3 --steps 20 --warmup 5 --quick --schedule sequential 2>/dev/null | tail -1 > gpurun_out/r6/call30_c3_b.json
python - <<'PY'
import json
for f in ("call30_c3.json", "call30_c3_b.json"):
    d = json.load(open("gpurun_out/r6/" + f))
    print(d["value"], d["ms_per_step"], {k: (round(v["avg_launch_us"], 2), round(v["frac"], 3)) for k, v in d.items() if k.startswith("roofline") and isinstance(v, dict) and "avg_launch_us" in v})
PY
