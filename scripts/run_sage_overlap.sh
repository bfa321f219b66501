#!/bin/bash
# A/B of the side-stream overlap on the SAGE products config (no per-op syncs).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DATA=${1:-/tmp/gaib_data}
mkdir -p "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" ogbn-products "$DATA"
export DATASET_PATH="$DATA/"
for v in 0 1; do
  echo "=== GAIB_OVERLAP=$v gpu_train_sage ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0"
  GAIB_OVERLAP=$v "$ROOT/bin/gpu_train_sage" ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0 | grep -E "Epoch   [5-9]|Average"
  echo "=== GAIB_OVERLAP=$v gpu_train_gcn ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0"
  GAIB_OVERLAP=$v "$ROOT/bin/gpu_train_gcn" ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0 | grep -E "Epoch   [5-9]|Average"
done
rm -rf "$DATA"
