#!/usr/bin/env bash
# Round-5 GPU call 3: f32x with tile variants / n_fast (tests + timing), classifier row split at 3,000 rows (C5 sweep A/B).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call3
mkdir -p "$OUT"
cd "$R"
timeout 900 python3 -m pytest tests/test_f32x_gpu.py tests/test_fullsize_gpu.py -q -m gpu > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 600 python3 tools/f32x_bench.py c2 c3 > $OUT/f32x_bench.json 2> $OUT/f32x_bench.err; tail -2 $OUT/f32x_bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/f32x_bench.json"))
for k, v in d.items():
    print(k, v["ms_per_step"], {a: b for a, b in list(v["event_timed_ms"].items())[:8]})
PY
B="python3 $R/bench.py"
for i in 0 1; do
  DH_VOCAB_SPLIT_ROWS=0 $B --workload c5 --steps 5 2>/dev/null | tail -1 > $OUT/c5_nosplit_$i.json
  $B --workload c5 --steps 5 2>/dev/null | tail -1 > $OUT/c5_split_$i.json
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c5_*.json")):
    d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_sweep"], 2))
PY
