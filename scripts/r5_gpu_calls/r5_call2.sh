#!/usr/bin/env bash
# Round-5 GPU call 2: the f32x path's first run (tests + timing), the C5 shard regime (fixed bench), the launch-floor probe, the
# exchange's cost.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call2
mkdir -p "$OUT"
cd "$R"
timeout 900 python3 -m pytest tests/test_f32x_gpu.py -q -m gpu > $OUT/f32x_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/f32x_tests.log
tail -25 $OUT/f32x_tests.log
timeout 600 python3 tools/f32x_bench.py c2 c3 > $OUT/f32x_bench.json 2> $OUT/f32x_bench.err; tail -3 $OUT/f32x_bench.err
B="python3 $R/bench.py"
$B --workload c5 --shard-of 8 --rccl-single --steps 5 --warmup 2 2>$OUT/shard_c5.err | tail -1 > $OUT/shard_c5_default.json
DH_DECODE_WREG_MIN_ROWS=100000 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_tile_gemms.json
DH_FUSED_BEAM_STEP=1 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_fused_beam.json
DH_CROSS_QPROJ=0 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_unfused_qproj.json
DH_DECODE_STREAMS=2 $B --workload c5 --shard-of 8 --shard-only --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_streams2.json
$B --workload c5 --shard-of 8 --shard-only --shard-rank 7 --rccl-single --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/shard_c5_rank7.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5s -o t -- python3 $R/bench.py --workload c5 --shard-of 8 --shard-only --steps 2 --warmup 1 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/prof_c5s/t_results.db --by-grid --top 0 --sequence 140 --csv $OUT/c5_shard_kernel_stats.csv > $OUT/c5_shard_kernel_stats.txt 2>&1 )
# launch floor probe, runtime knobs swept
P=$R/tools/probe/launch_floor_probe
$P > $OUT/floor_default.jsonl 2>&1
AMD_OPT_FLUSH=0 $P > $OUT/floor_optflush0.jsonl 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $P > $OUT/floor_nopktcapture.jsonl 2>&1
HIP_FORCE_DEV_KERNARG=0 $P > $OUT/floor_hostkernarg.jsonl 2>&1
python3 tools/gather_cost.py > $OUT/gather_cost.json 2> $OUT/gather_cost.err
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
    elif "value" in d:
        print(os.path.basename(f), "value %.0f ms %.3f" % (d["value"], d["ms_per_step"]))
PY
cat $OUT/gather_cost.json; cat $OUT/floor_default.jsonl
