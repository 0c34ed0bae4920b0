#!/usr/bin/env bash
# Round-5 GPU call 10: the default bench line on the final schedule (roofline events from the sequential pass), C5, and the full
# randomised sweeps (summary -> profiles/r5/fuzz_summary.txt).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call10
mkdir -p "$OUT"
cd "$R"
python3 $R/bench.py 2>$OUT/err_default.txt > $OUT/bench_default_bf16.json
python3 $R/bench.py --workload c5 --steps 5 2>/dev/null | tail -1 > $OUT/bench_c5_f16.json
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 --quick --rccl-single 2>/dev/null | tail -1 > $OUT/bench_c2_rccl_single_rank.json
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_default_bf16.json").read().strip().splitlines()[-1])
print("C2", round(d["value"], 1), d["ms_per_step"], "seq", d.get("value_sequential"), "roofline", d["roofline"]["frac"], d["roofline"]["measured"][:60])
print("C3", round(d["c3"]["value"], 1), d["c3"]["ms_per_step"], d["c3"].get("sequential"), d["c3"]["roofline"]["kernel"], d["c3"]["roofline"]["frac"])
print("parity grade", d.get("value_parity_grade"), d["c3"].get("parity_grade_path", {}).get("value"))
PY
SEED=11 SCALE=1 bash tools/fuzz_all.sh > $OUT/fuzz_summary.txt 2>&1
cat $OUT/fuzz_summary.txt
