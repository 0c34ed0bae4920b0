#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel traces and PMC passes of the bench workloads of THIS tree, reduced to
# CSV summaries under gpurun_out/profiles_${ROUND:-r6}/ (copied to profiles/<round>/ afterwards).  Counters in their own passes
# (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_${ROUND:-r6}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in c2 c3; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o t -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --quick --schedule sequential > $OUT/bench_${wl}_bf16_under_rocprof.json 2>/dev/null
  python3 $R/tools/rocpd_stats.py /tmp/prof_$wl/t_results.db --by-grid --top 0 --csv $OUT/${wl}_bf16_kernel_stats.csv 2> $OUT/${wl}_bf16_kernel_stats.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${wl}_$ctr -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
    python3 $R/tools/rocpd_stats.py /tmp/pmc_${wl}_$ctr/p_results.db --pmc --top 0 --csv $OUT/pmc_${wl}_$ctr.csv
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_${wl}_mfma -o p -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/pmc_${wl}_mfma/p_results.db --pmc --top 0 --csv $OUT/pmc_${wl}_mfma.csv
done
# SQ-level PMC (wait / issue / LDS / MFMA counters) of the stand-alone probes: classifier (both kernels), stage-3 bottleneck tail, the round-4
# encoder kernels, LSTM step, decode GEMMs
for pr in vocab_probe s3_probe enc_probe lstm_probe kbench; do
  bash $R/tools/pmc_sq.sh $pr $R/tools/$pr.py > /dev/null 2>&1
done
cp $R/gpurun_out/pmc_sq/*.csv $OUT/ 2>/dev/null
python3 $R/tools/make_pmc_json.py $OUT "$(cat $R/deephumor_amd/lib/BUILD_COMMIT 2>/dev/null || echo unknown)" > /dev/null 2>&1
mkdir -p $R/profiles/${ROUND:-r6} && cp $OUT/pmc_hbm_traffic.json $R/profiles/${ROUND:-r6}/pmc_hbm_traffic.json   # bench.py reads it for roofline.traffic
python3 $R/bench.py > $OUT/bench_default_bf16.json 2> /dev/null
python3 $R/bench.py --workload c5 --steps 5 > $OUT/bench_c5_f16.json 2> /dev/null
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 --quick --rccl-single 2> /dev/null | tail -1 > $OUT/bench_c2_rccl_single_rank.json
python3 $R/bench.py --workload score-c2 --steps 3 > $OUT/bench_score_c2.json 2> /dev/null
python3 $R/bench.py --workload score-c3 --steps 3 > $OUT/bench_score_c3.json 2> /dev/null
# round 5: the fp32 paths under rocprofv3 (exact fp32, and option f32_split = the split-operand matrix-core path), their timing,
# the shard regime as one rank of 8 would see it, the launch-floor probe, the exchange's cost
for mode in 0 1; do
  DH_F32_SPLIT=$mode rocprofv3 --kernel-trace --stats -d /tmp/prof_f32_$mode -o t -- python3 $R/bench.py --workload c2 --dtype f32 --steps 2 --warmup 1 --quick --schedule sequential > $OUT/bench_c2_f32_split${mode}_under_rocprof.json 2>/dev/null
  python3 $R/tools/rocpd_stats.py /tmp/prof_f32_$mode/t_results.db --by-grid --top 0 --csv $OUT/c2_f32_split${mode}_kernel_stats.csv 2> $OUT/c2_f32_split${mode}_kernel_stats.txt
  DH_F32_SPLIT=$mode rocprofv3 --kernel-trace --stats -d /tmp/prof_f32c3_$mode -o t -- python3 $R/bench.py --workload c3 --dtype f32 --steps 2 --warmup 1 --quick --schedule sequential > $OUT/bench_c3_f32_split${mode}_under_rocprof.json 2>/dev/null
  python3 $R/tools/rocpd_stats.py /tmp/prof_f32c3_$mode/t_results.db --by-grid --top 0 --csv $OUT/c3_f32_split${mode}_kernel_stats.csv 2> $OUT/c3_f32_split${mode}_kernel_stats.txt
done
cd $R
python3 $R/tools/f32x_bench.py c2 c3 > $OUT/f32x_bench.json 2>/dev/null
python3 $R/bench.py --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_c5_shard_of_8_rank0.json
python3 $R/bench.py --workload c5 --shard-of 8 --shard-rank 7 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_c5_shard_of_8_rank7.json
python3 $R/bench.py --workload c3 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_c3_shard_of_8.json
python3 $R/bench.py --workload c3 --batch 32 --shard-of 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_c3_batch32_strong_shard.json
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5s -o t -- python3 $R/bench.py --workload c5 --shard-of 8 --shard-only --steps 2 --warmup 1 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_c5s/t_results.db --by-grid --top 0 --sequence 60 --csv $OUT/c5_shard_kernel_stats.csv > $OUT/c5_shard_kernel_stats.txt 2>&1 )
python3 $R/tools/gather_cost.py > $OUT/gather_cost.json 2>/dev/null
# round 6: the split-operand decode layers per shape, the persistent decoder-layer kernel (opt-in: C3 step with it, per-phase stamps of
# layer 0, rocprofv3 duration of the launch), the encoder as sub-batches on several streams
python3 $R/tools/f32x_kbench.py > $OUT/f32x_kbench.txt 2>/dev/null
DH_DECODE_LAYERS=1 python3 $R/bench.py --workload c3 --steps 5 --warmup 2 --quick --schedule sequential 2>/dev/null | tail -1 > $OUT/bench_c3_decode_layers.json
DH_DECODE_LAYERS=1 DH_DL_DEBUG=2 python3 $R/tools/decode_layers_stamps.py > $OUT/decode_layers_phase_stamps.txt 2>/dev/null
( cd /tmp && DH_DECODE_LAYERS=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3dl -o t -- python3 $R/bench.py --workload c3 --steps 2 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_c3dl/t_results.db --by-grid --top 12 --csv $OUT/c3_decode_layers_kernel_stats.csv 2> $OUT/c3_decode_layers_kernel_stats.txt )
python3 $R/tools/enc_streams_probe.py > $OUT/encoder_streams_probe.txt 2>/dev/null
ls -la $OUT
