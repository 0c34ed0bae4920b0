#!/usr/bin/env bash
# K / V prefetch workgroups on the fc_q launch (option cross_kv_prefetch) in front of the packed cross-attention: tests, then A/B of
# whole C3 steps and rocprofv3 kernel-only times, one call
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call14
mkdir -p "$OUT"
cd "$R"
timeout 600 python3 -m pytest tests/test_bf16_gpu.py -q -m gpu -x -k "prefetch or unfused_qproj" > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
grep -v "^  File\|^Extension" $OUT/tests.log | tail -8
for cfg in "1 0" "0 0" "0 256" "0 128" "1 0" "0 0" "0 256" "0 64"; do
  set -- $cfg
  DH_CROSS_QPROJ=$1 DH_CROSS_KV_PREFETCH=$2 timeout 300 python3 bench.py --workload c3 --quick --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/c3_q$1_pf$2_$RANDOM.json
done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r5_call14/c3_q*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    g = lambda k: round((d.get(k) or {}).get("frac") or 0, 3)
    print(os.path.basename(f), "ms", round(d.get("ms_per_step", 0), 3), "seq", round((d.get("sequential") or {}).get("ms_per_step") or 0, 3),
          "self", g("roofline_self_attention"), "cross", g("roofline_cross_attention"), "pair", g("roofline_decoder_attention_combined"),
          "cross_us", round((d.get("roofline_cross_attention") or {}).get("avg_launch_us") or 0, 2))
PY
cd /tmp && export TMPDIR=/tmp
for pf in 0 256; do
  DH_CROSS_QPROJ=0 DH_CROSS_KV_PREFETCH=$pf rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$pf -o t -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_pf$pf/t_results.db --by-grid --top 0 --csv $OUT/c3_pf${pf}_kernel_stats.csv 2> $OUT/c3_pf${pf}_kernel_stats.txt
  head -1 $OUT/c3_pf${pf}_kernel_stats.txt
  grep -i "cross_mfma\|linear_wreg_kernelIDF16bLi4ELi40ELi1ELi0\|linear_wreg_kernelIDF16bLi4ELi40ELi1ELi1" $OUT/c3_pf${pf}_kernel_stats.csv | cut -c1-200 | head -6
done
