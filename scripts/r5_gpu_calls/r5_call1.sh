#!/usr/bin/env bash
# Round-5 GPU call 1 (runs ON THE GPU BOX through gpurun): the -m gpu suite, then the small-shard regime:
# C5's 38-template x beam-10 fp16 shard and C3's 256-image shard timed as one rank of an 8-rank run would see them, with the
# row-count-selectable fusions toggled through their options' environment defaults, a rocprofv3 kernel trace of the C5 shard, and
# the --rccl-single A/B in the same call.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call1
mkdir -p "$OUT"
cd "$R"
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
tail -5 $OUT/gputest.log
B="python3 $R/bench.py"
$B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_default.json
DH_DECODE_WREG_MIN_ROWS=100000 $B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_tile_gemms.json
DH_FUSED_BEAM_STEP=1 $B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_fused_beam.json
DH_CROSS_QPROJ=0 $B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_unfused_qproj.json
DH_DECODE_STREAMS=2 $B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_streams2.json
$B --workload c5 --shard-of 8 --shard-rank 7 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_rank7.json
$B --workload c3 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c3_default.json
$B --workload c2 --shard-of 8 --rccl-single --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c2_default.json
# C3 at small batches (what an 8-rank strong-scaled C3 would see): 32 images x beam 5 = 160 rows, with the qkv fusion on / off
$B --workload c3 --batch 32 --shard-of 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c3_b32_default.json
DH_QKV_FUSION_MAX_ROWS=400 $B --workload c3 --batch 32 --shard-of 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c3_b32_qkvfusion.json
# rocprofv3 kernel trace of the C5 shard
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5s -o t -- python3 $R/bench.py --workload c5 --shard-of 8 --steps 2 --warmup 1 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_c5s/t_results.db --by-grid --top 0 --sequence 130 --csv $OUT/c5_shard_kernel_stats.csv > $OUT/c5_shard_kernel_stats.txt 2>&1 )
# --rccl-single A/B in one call, alternating
for i in 0 1; do
  $B --workload c2 --quick --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/c2_plain_$i.json
  $B --workload c2 --quick --steps 20 --warmup 5 --rccl-single 2>/dev/null | tail -1 > $OUT/c2_rccl_$i.json
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    if "shard" in d:
        print(os.path.basename(f), "shard_ms %.2f" % d["shard"]["shard_ms"], "launches", d["shard"]["breakdown"]["launches_per_step"], "one_gpu_ms %.2f" % d["one_gpu"]["ms"], {k: round(v, 1) if isinstance(v, float) else v for k, v in d["projection"].items() if k.startswith("projected")})
    else:
        print(os.path.basename(f), "value %.0f ms %.3f" % (d["value"], d["ms_per_step"]))
PY
