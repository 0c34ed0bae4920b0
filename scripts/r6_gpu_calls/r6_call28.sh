#!/usr/bin/env bash
# round 6, call 28: the f32x sweep again with the tool's plane-reshape fixed (cout % 32 != 0), three seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for s in 83 5 6; do timeout 900 python3 tools/fuzz_f32x.py --trials 400 --seed $s > gpurun_out/r6/call28_fuzz_f32x_seed$s.jsonl 2> gpurun_out/r6/call28_fuzz_f32x_seed$s.err; echo "fuzz_f32x seed $s: $(tail -1 gpurun_out/r6/call28_fuzz_f32x_seed$s.jsonl)"; grep -v '"ok": true' gpurun_out/r6/call28_fuzz_f32x_seed$s.jsonl | grep -v '^{"trials"' | head -5 | cut -c1-300; done
