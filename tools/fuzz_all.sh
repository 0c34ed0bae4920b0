#!/usr/bin/env bash
# Runs every randomised sweep of tools/ ON THE GPU BOX (through gpurun) with one seed and prints one summary line per tool:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'SEED=11 SCALE=1 bash tools/fuzz_all.sh'
# SCALE multiplies the trial counts (1 = about 12 minutes).  Failing trials stay in gpurun_out/fuzz_all/<tool>.jsonl.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${SEED:-11}
K=${SCALE:-1}
OUT=$R/gpurun_out/fuzz_all
mkdir -p "$OUT"
run() {   # name, trials, extra args...
  local name=$1 n=$(( $2 * K )); shift 2
  local tag=${name}$(echo "$*" | tr -d ' -')
  timeout 1500 python3 $R/tools/$name.py --trials $n --seed $S "$@" > $OUT/$tag.jsonl 2> $OUT/$tag.err
  echo "$tag: $(tail -1 $OUT/$tag.jsonl)"
  grep -v '"ok": true' $OUT/$tag.jsonl | grep -v 'fewer positive' | grep -v '^{"trials"' | head -3 | cut -c1-300
}
run fuzz_generate 250 --half
run fuzz_generate 25 --long
run fuzz_generate 120 --r4 --half
run fuzz_generate 150 --split
run fuzz_forward 250
run fuzz_forward 40 --big
run fuzz_encoder 80
run fuzz_beam_methods 300
run fuzz_sampler 500
run fuzz_scoring 300
run fuzz_gemm 300
run fuzz_conv 150
run fuzz_f32x 300
run fuzz_pipeline 120
run fuzz_variants 120
run poison_check 40
