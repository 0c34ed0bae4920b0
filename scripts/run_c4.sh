#!/usr/bin/env bash
# BASELINE config C4: CaptioningTransformer, batch 2048 image-sharded over the 8 MI355X of one node (256 images per GPU), ONE
# all_gather of token ids per batch.  bench.py starts its own ranks (child torch.distributed.run, before any GPU call).
#   scripts/run_c4.sh [N_GPUS=8] [STEPS=10] [WARMUP=3]
cd "$(dirname "$0")/.." && exec python bench.py --gpus "${1:-8}" --workload c3 --steps "${2:-10}" --warmup "${3:-3}" --no-cpu
