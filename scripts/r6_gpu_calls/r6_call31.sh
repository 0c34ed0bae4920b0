#!/usr/bin/env bash
# round 6, call 31: the streaming 1x1 kernel with Cin = 512, stride 2 and the planes output -- tests, per-layer times, whole steps, sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_f32x_gpu.py -q -x 2>&1 | tail -5
timeout 600 python tools/f32x_conv1x1_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/call31_conv1x1_bench.txt
timeout 900 python tools/f32x_bench.py c2 c3 > gpurun_out/r6/call31_f32x_bench.txt 2>/dev/null
python - <<'PY'
import json
t = open("gpurun_out/r6/call31_f32x_bench.txt").read()
d = json.loads(t[t.index("{"):])
for k, v in d.items():
    print(k, v["ms_per_step"])
PY
timeout 900 python3 tools/fuzz_f32x.py --trials 300 --seed 9 > gpurun_out/r6/call31_fuzz_f32x.jsonl 2>/dev/null; tail -1 gpurun_out/r6/call31_fuzz_f32x.jsonl; grep -v '"ok": true' gpurun_out/r6/call31_fuzz_f32x.jsonl | grep -v '^{"trials"' | head -5 | cut -c1-300
