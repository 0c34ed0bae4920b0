#!/usr/bin/env bash
# fused q-projection + cross-attention against fc_q GEMM + packed cross-attention, in ONE call: whole C3 steps (sequential and
# pipelined), the instrumented attention fractions, and rocprofv3 kernel-only times of both
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call13
mkdir -p "$OUT"
cd "$R"
timeout 300 python3 tools/kbench.py > $OUT/kbench.txt 2>&1; grep "cross\|self\|proj:WREG(a_ln)" $OUT/kbench.txt
for q in 1 0 1 0; do
  DH_CROSS_QPROJ=$q timeout 300 python3 bench.py --workload c3 --quick --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/c3_q${q}_$RANDOM.json
done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r5_call13/c3_q*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    d = d.get("c3", d)
    g = lambda k: (d.get(k) or {}).get("frac")
    print(os.path.basename(f), "ms", round(d.get("ms_per_step", 0), 3), "seq", (d.get("sequential") or {}).get("ms_per_step"),
          "self", g("roofline_self_attention"), "cross", g("roofline_cross_attention"), "pair", g("roofline_decoder_attention_combined"),
          "cross_us", (d.get("roofline_cross_attention") or {}).get("avg_launch_us"))
PY
cd /tmp && export TMPDIR=/tmp
for q in 1 0; do
  DH_CROSS_QPROJ=$q rocprofv3 --kernel-trace --stats -d /tmp/prof_q$q -o t -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_q$q/t_results.db --by-grid --top 14 --csv $OUT/c3_q${q}_kernel_stats.csv 2> $OUT/c3_q${q}_kernel_stats.txt
  grep -i "cross\|linear_wreg_kernelIDF16bLi4ELi40ELi1ELi0" $OUT/c3_q${q}_kernel_stats.csv | cut -c1-200
done
