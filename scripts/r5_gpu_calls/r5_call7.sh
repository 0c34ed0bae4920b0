#!/usr/bin/env bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call7
mkdir -p "$OUT"
cd "$R"
timeout 1800 python3 -m pytest tests -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
tail -8 $OUT/gputest.log
B="python3 $R/bench.py"
for b in 1 2 4; do
  $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_minrows1.json
  DH_DECODE_WREG_MIN_ROWS=16 $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_minrows16.json
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c*.json")):
    d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3))
PY
