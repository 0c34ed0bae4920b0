#!/usr/bin/env bash
# BASELINE config C5: ImageLabelEncoder + CaptioningTransformer, fp16, beam 10, the full 300-template sweep sharded
# 38/38/38/38/37/37/37/37 over 8 MI355X (uneven shards: one padded all_gather of token ids per sweep).
#   scripts/run_c5.sh [N_GPUS=8] [SWEEPS=10]
cd "$(dirname "$0")/.." && exec python bench.py --gpus "${1:-8}" --workload c5 --dtype f16 --steps "${2:-10}"
