#!/usr/bin/env bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call9
mkdir -p "$OUT"
cd "$R"
B="python3 $R/bench.py"
timeout 600 python3 -m pytest tests/test_dist_gpu.py -q -m gpu > $OUT/dist_tests.log 2>&1; tail -3 $OUT/dist_tests.log
for i in 0 1; do
  $B --workload c2 --quick --steps 20 --warmup 5 --schedule sequential 2>$OUT/err_seq_$i.txt | tail -1 > $OUT/c2_sequential_$i.json
  $B --workload c2 --quick --steps 20 --warmup 5 2>$OUT/err_pipe_$i.txt | tail -1 > $OUT/c2_pipelined_$i.json
done
$B --workload c3 --quick --steps 10 --warmup 3 --schedule sequential 2>/dev/null | tail -1 > $OUT/c3_sequential.json
$B --workload c3 --quick --steps 10 --warmup 3 2>$OUT/err_c3.txt | tail -1 > $OUT/c3_pipelined.json
$B 2>$OUT/err_default.txt > $OUT/bench_default_bf16.json
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c*.json")) + ["$OUT/bench_default_bf16.json"]:
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3), d.get("value_sequential"), d["config"].get("schedule", "")[:12], d["roofline"]["frac"])
PY
tail -3 $OUT/err_pipe_0.txt $OUT/err_default.txt
