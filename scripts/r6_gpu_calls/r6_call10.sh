#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for dt in bf16 f16 bf16 f16; do
  python bench.py --workload c2 --dtype $dt --steps 10 --warmup 3 --quick 2>/dev/null | tail -1 > gpurun_out/r6/call10_c2_$dt.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/call10_c2_$dt.json"))
print("$dt", "seq", round(d["ms_per_step"], 3), "pipelined", round(d.get("ms_per_step_pipelined", 0), 3), "graph", round(d["hipgraph_replay"]["ms_per_step"], 3), {k: v for k, v in list(d["kernel_breakdown_ms_per_step"].items())[:7]})
PY
done
