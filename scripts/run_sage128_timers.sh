#!/bin/bash
# per-op timer table of config 3 (SAGE 3-layer, hidden 128) -- GAIB_SYNC_TIMERS=1 syncs after every op
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DATA=${1:-/tmp/gaib_data}
mkdir -p "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" ogbn-products "$DATA" > /dev/null
export DATASET_PATH="$DATA/"
GAIB_SYNC_TIMERS=1 "$ROOT/bin/gpu_train_sage" ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0 | grep -E "Epoch   9|Average|time:"
GAIB_SYNC_TIMERS=1 "$ROOT/bin/gpu_train_gcn" ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0 | grep -E "Epoch   9|Average|time:"
rm -rf "$DATA"
