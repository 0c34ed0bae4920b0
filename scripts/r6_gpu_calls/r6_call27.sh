#!/usr/bin/env bash
# round 6, call 27: every randomised sweep on the final tree (one seed), the f32x sweep (now with the planes kernels) with two more seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
SEED=83 SCALE=1 bash tools/fuzz_all.sh > gpurun_out/r6/call27_fuzz_all_seed83.txt 2>&1
cat gpurun_out/r6/call27_fuzz_all_seed83.txt
for s in 5 6; do timeout 900 python3 tools/fuzz_f32x.py --trials 400 --seed $s > gpurun_out/r6/call27_fuzz_f32x_seed$s.jsonl 2> gpurun_out/r6/call27_fuzz_f32x_seed$s.err; echo "fuzz_f32x seed $s: $(tail -1 gpurun_out/r6/call27_fuzz_f32x_seed$s.jsonl)"; grep -v '"ok": true' gpurun_out/r6/call27_fuzz_f32x_seed$s.jsonl | grep -v '^{"trials"' | head -5 | cut -c1-300; done
