#!/usr/bin/env bash
# Round-5 GPU call 4: f32x with the deep-ring small-tile kernel (tests + timing); fused vs unfused q-projection by row count.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call4
mkdir -p "$OUT"
cd "$R"
timeout 900 python3 -m pytest tests/test_f32x_gpu.py -q -m gpu > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 600 python3 tools/f32x_bench.py c2 c3 > $OUT/f32x_bench.json 2> $OUT/f32x_bench.err; tail -2 $OUT/f32x_bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/f32x_bench.json"))
for k, v in d.items():
    print(k, v["ms_per_step"], {a: b for a, b in list(v["event_timed_ms"].items())[:9]})
PY
B="python3 $R/bench.py"
for b in 32 64 128 256; do
  for q in 1 0; do
    DH_CROSS_QPROJ=$q $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_qproj${q}.json
  done
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c3_b*.json")):
    d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3))
PY
