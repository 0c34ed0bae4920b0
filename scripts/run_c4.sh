#!/usr/bin/env bash
# BASELINE config C4: CaptioningTransformer, batch 2048 image-sharded over the 8 MI355X of one node (256 images per GPU),
# one process per GPU over RCCL/xGMI, ONE all_gather of token ids per batch.  The launcher (torch.distributed.run) starts
# before anything touches a GPU; no process that has initialised HIP ever re-execs.
#   scripts/run_c4.sh [N_GPUS=8] [STEPS=10] [WARMUP=3]
set -euo pipefail
cd "$(dirname "$0")/.."
N=${1:-8}; STEPS=${2:-10}; WARMUP=${3:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
python __graft_entry__.py build           # every rank loads the prebuilt library; nothing is compiled under the launcher
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29511}" \
    bench.py --gpus "$N" --workload c3 --steps "$STEPS" --warmup "$WARMUP" --no-cpu
