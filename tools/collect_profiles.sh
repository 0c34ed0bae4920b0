#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel traces and PMC passes of the bench workloads of THIS tree, reduced to
# CSV summaries under gpurun_out/profiles_${ROUND:-r4}/ (copied to profiles/<round>/ afterwards).  Counters in their own passes
# (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_${ROUND:-r4}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in c2 c3; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o t -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --quick > $OUT/bench_${wl}_bf16_under_rocprof.json 2>/dev/null
  python3 $R/tools/rocpd_stats.py /tmp/prof_$wl/t_results.db --by-grid --top 0 --csv $OUT/${wl}_bf16_kernel_stats.csv 2> $OUT/${wl}_bf16_kernel_stats.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${wl}_$ctr -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 --quick > /dev/null 2>&1
    python3 $R/tools/rocpd_stats.py /tmp/pmc_${wl}_$ctr/p_results.db --pmc --top 0 --csv $OUT/pmc_${wl}_$ctr.csv
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_${wl}_mfma -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 --quick > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/pmc_${wl}_mfma/p_results.db --pmc --top 0 --csv $OUT/pmc_${wl}_mfma.csv
done
# SQ-level PMC (wait / issue / LDS / MFMA counters) of the stand-alone probes: classifier (both kernels), stage-3 bottleneck tail, the round-4
# encoder kernels, LSTM step, decode GEMMs
for pr in vocab_probe s3_probe enc_probe lstm_probe kbench; do
  bash $R/tools/pmc_sq.sh $pr $R/tools/$pr.py > /dev/null 2>&1
done
cp $R/gpurun_out/pmc_sq/*.csv $OUT/ 2>/dev/null
python3 $R/tools/make_pmc_json.py $OUT "$(cat $R/deephumor_amd/lib/BUILD_COMMIT 2>/dev/null || echo unknown)" > /dev/null 2>&1
mkdir -p $R/profiles/${ROUND:-r4} && cp $OUT/pmc_hbm_traffic.json $R/profiles/${ROUND:-r4}/pmc_hbm_traffic.json   # bench.py reads it for roofline.traffic
python3 $R/bench.py > $OUT/bench_default_bf16.json 2> /dev/null
python3 $R/bench.py --workload c5 --steps 5 > $OUT/bench_c5_f16.json 2> /dev/null
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 --quick --rccl-single 2> /dev/null | tail -1 > $OUT/bench_c2_rccl_single_rank.json
python3 $R/bench.py --workload score-c2 --steps 3 > $OUT/bench_score_c2.json 2> /dev/null
python3 $R/bench.py --workload score-c3 --steps 3 > $OUT/bench_score_c3.json 2> /dev/null
ls -la $OUT
