#!/usr/bin/env bash
# Round-5 GPU call 5: the whole -m gpu suite on the current tree; row thresholds of the register-stationary kernels at small batches.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call5
mkdir -p "$OUT"
cd "$R"
timeout 1800 python3 -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
tail -5 $OUT/gputest.log
B="python3 $R/bench.py"
for b in 16 32 48; do
  $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_default.json
  DH_DECODE_WREG_MIN_ROWS=40 $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_wreg40.json
done
for b in 16 32 48; do
  $B --workload c2 --batch $b --quick --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/c2_b${b}_default.json
  DH_LSTM_WREG_MIN_ROWS=80 $B --workload c2 --batch $b --quick --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/c2_b${b}_lstmwreg80.json
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c*_b*.json")):
    d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3))
PY
