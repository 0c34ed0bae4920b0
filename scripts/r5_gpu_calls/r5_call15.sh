#!/usr/bin/env bash
# launch sequence of one C2 decode position (which launches are torch's, how large the gaps are)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_call15
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2s -o t -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --quick --schedule sequential > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/prof_c2s/t_results.db --by-grid --top 0 --sequence 400 --csv $OUT/c2_seq_kernel_stats.csv > $OUT/c2_sequence.txt 2>&1
tail -150 $OUT/c2_sequence.txt | cut -c1-150
