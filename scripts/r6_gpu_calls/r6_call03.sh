#!/usr/bin/env bash
# round 6, call 3: the register-stationary split-operand decode layers (tests, per-shape timing, whole steps), the degraded-kernel
# check of the tightened gates, the encoder multi-stream probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_f32x_gpu.py -q -x -k "wreg or range_guard or exact_path or greedy_ids or forward_logits" > gpurun_out/r6/call03_f32x_tests.txt 2>&1
tail -3 gpurun_out/r6/call03_f32x_tests.txt
python tools/f32x_kbench.py > gpurun_out/r6/call03_f32x_kbench.txt 2>&1
cat gpurun_out/r6/call03_f32x_kbench.txt
python tools/f32x_bench.py c2 c3 > gpurun_out/r6/call03_f32x_bench.json 2> gpurun_out/r6/call03_f32x_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call03_f32x_bench.json"))
for k, v in d.items():
    print(k, v["ms_per_step"], {kk: vv for kk, vv in list(v["event_timed_ms"].items())[:6]})
PY
# the gates must turn red on a degraded kernel (bf16-rounded LSTM gate pre-activations)
DEEPHUMOR_HIP_LIB=$PWD/scratch/degraded/libdegraded.so python -m pytest tests/test_fullsize_gpu.py -q -k "16bit and LSTM" > gpurun_out/r6/call03_degraded_gates.txt 2>&1
tail -8 gpurun_out/r6/call03_degraded_gates.txt
python tools/enc_streams_probe.py > gpurun_out/r6/call03_enc_streams.txt 2>&1
cat gpurun_out/r6/call03_enc_streams.txt
