#!/usr/bin/env bash
# round 6, call 26: the tree with the f32x planes work -- full -m gpu suite, the f32 split path under rocprofv3, f32x steps, default bench line
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6/final_f32x
mkdir -p $OUT
( time python -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r6/call26_pytest.txt 2>&1
tail -34 gpurun_out/r6/call26_pytest.txt
( cd /tmp && export TMPDIR=/tmp
for wl in c2 c3; do
  DH_F32_SPLIT=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32_$wl -o t -- python3 $R/bench.py --workload $wl --dtype f32 --steps 2 --warmup 1 --quick --schedule sequential > $OUT/bench_${wl}_f32_split1_under_rocprof.json 2>/dev/null
  python3 $R/tools/rocpd_stats.py /tmp/prof_f32_$wl/t_results.db --by-grid --top 0 --csv $OUT/${wl}_f32_split1_kernel_stats.csv 2> $OUT/${wl}_f32_split1_kernel_stats.txt
done )
timeout 900 python tools/f32x_bench.py c2 c3 > $OUT/f32x_bench.json 2>/dev/null
timeout 300 python tools/f32xp_kbench.py > $OUT/f32xp_kbench.txt 2>&1
( time python bench.py > $OUT/bench_default_bf16.json 2> $OUT/bench_default_bf16.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/final_f32x/bench_default_bf16.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_pipelined", "value_f16", "value_parity_grade")})
print(d["parity_grade_path"]["ms_per_step"], d["c3"]["ms_per_step"], d["c3"]["parity_grade_path"]["ms_per_step"])
t = open("gpurun_out/r6/final_f32x/f32x_bench.json").read()
x = json.loads(t[t.index("{"):])
for k, v in x.items(): print(k, v["ms_per_step"])
PY
