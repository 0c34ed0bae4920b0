#!/usr/bin/env bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call12
mkdir -p "$OUT"
cd "$R"
timeout 300 python3 -m pytest tests/test_bf16_gpu.py -q -m gpu -s -x -k "test_decode_gemm_chain_equals_separate_launches" > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
grep -v "^  File\|^Extension" $OUT/tests.log | tail -25
DH_DECODE_CHAIN_FUSION=1 timeout 300 python3 $R/bench.py --workload c3 --batch 32 --quick --steps 3 --warmup 1 --schedule sequential > $OUT/c3_b32.json 2> $OUT/c3_b32.err; tail -12 $OUT/c3_b32.err
