#!/usr/bin/env bash
# Round-5 GPU call 6: full suite on the tree with the small-row dispatch changes (decode_wreg_min_rows 16, register-streamed classifier
# at <= 640 rows for every decoder), then the small-shard regime again.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call6
mkdir -p "$OUT"
cd "$R"
timeout 1800 python3 -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
tail -5 $OUT/gputest.log
B="python3 $R/bench.py"
$B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_default.json
DH_VOCAB_WREG_TRANSFORMER_MAX_ROWS=0 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_areg_classifier.json
DH_CROSS_QPROJ=0 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_unfused_qproj.json
for b in 1 8 32; do
  $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_new.json
  DH_VOCAB_WREG_TRANSFORMER_MAX_ROWS=0 DH_DECODE_WREG_MIN_ROWS=320 $B --workload c3 --batch $b --quick --steps 6 --warmup 2 2>/dev/null | tail -1 > $OUT/c3_b${b}_r4dispatch.json
  $B --workload c2 --batch $b --quick --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/c2_b${b}_new.json
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(os.path.basename(f), "unreadable"); continue
    if "shard" in d:
        print(os.path.basename(f), "shard_ms %.2f" % d["shard"]["shard_ms"], "one_gpu_ms %.2f" % d["one_gpu"]["ms"], {k: round(v, 2) if isinstance(v, float) else v for k, v in d["projection"].items() if k.startswith("projected")})
    else:
        print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3))
PY
