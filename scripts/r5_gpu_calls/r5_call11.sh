#!/usr/bin/env bash
# Round-5 GPU call 11: the one-launch GEMM chain (dh_decode_gemm_chain): bit-equality tests, then A/B timing.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call11
mkdir -p "$OUT"
cd "$R"
timeout 900 python3 -m pytest tests/test_bf16_gpu.py -q -m gpu -k "chain" > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
tail -15 $OUT/tests.log
B="timeout 300 python3 $R/bench.py"
for i in 0 1; do
  for f in 0 1; do
    DH_DECODE_CHAIN_FUSION=$f $B --workload c3 --quick --steps 8 --warmup 2 --schedule sequential 2>/dev/null | tail -1 > $OUT/c3_b256_chain${f}_$i.json
    DH_DECODE_CHAIN_FUSION=$f $B --workload c3 --batch 32 --quick --steps 8 --warmup 2 --schedule sequential 2>/dev/null | tail -1 > $OUT/c3_b32_chain${f}_$i.json
    DH_DECODE_CHAIN_FUSION=$f $B --workload c5 --shard-of 8 --shard-only --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/c5_shard_chain${f}_$i.json
  done
done
DH_DECODE_CHAIN_FUSION=1 $B --workload c5 --steps 4 2>/dev/null | tail -1 > $OUT/c5_full_chain1.json
DH_DECODE_CHAIN_FUSION=0 $B --workload c5 --steps 4 2>/dev/null | tail -1 > $OUT/c5_full_chain0.json
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/c*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(os.path.basename(f), "unreadable"); continue
    if "shard" in d: print(os.path.basename(f), "shard_ms %.2f" % d["shard"]["shard_ms"], "launches", d["shard"]["breakdown"]["launches_per_step"])
    else: print(os.path.basename(f), round(d["value"], 1), round(d.get("ms_per_step", d.get("ms_per_sweep", 0)), 3))
PY
