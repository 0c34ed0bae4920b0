#!/usr/bin/env bash
# round 6, call 29: the contract line alone on a fresh box (final tree), then the --split generate sweep with f32_planes alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( time python bench.py > gpurun_out/r6/call29_bench_default.json 2> gpurun_out/r6/call29_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/call29_bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_pipelined", "ms_per_step_pipelined", "value_f16", "value_parity_grade", "whole_step_mfma_frac")})
print(d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("kernel_only_us"), d["roofline"].get("frac_kernel_only"))
c3 = d["c3"]
print("c3", c3["value"], c3["ms_per_step"], c3["pipelined"]["ms_per_step"], c3.get("f16_path", {}).get("ms_per_step"), c3["parity_grade_path"]["ms_per_step"], c3["fp32_parity_path"]["ms_per_step"])
print("c2 f32x", d["parity_grade_path"]["ms_per_step"], d["fp32_parity_path"]["ms_per_step"], d.get("f16_path"))
print(c3.get("roofline_decoder_attention_combined"))
print(d["cpu_baseline"]["value"], c3["cpu_baseline"]["value"], d.get("speedup_vs_cpu"), c3.get("speedup_vs_cpu"))
PY
timeout 900 python3 tools/fuzz_generate.py --trials 150 --seed 7 --split > gpurun_out/r6/call29_fuzz_generate_split.jsonl 2>/dev/null; tail -1 gpurun_out/r6/call29_fuzz_generate_split.jsonl
