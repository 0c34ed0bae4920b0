#!/usr/bin/env bash
# BASELINE config C5: ImageLabelEncoder + CaptioningTransformer, fp16, beam 10, the full 300-template sweep sharded
# 38/38/38/38/37/37/37/37 over 8 MI355X (uneven shards: one padded all_gather of token ids per sweep).
#   scripts/run_c5.sh [N_GPUS=8] [SWEEPS=10]
set -euo pipefail
cd "$(dirname "$0")/.."
N=${1:-8}; SWEEPS=${2:-10}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
python __graft_entry__.py build
if [ "$N" -eq 1 ]; then exec python bench.py --workload c5 --dtype f16 --steps "$SWEEPS"; fi
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29512}" \
    bench.py --gpus "$N" --workload c5 --dtype f16 --steps "$SWEEPS"
