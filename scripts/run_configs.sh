#!/bin/bash
# End-to-end runs of the BASELINE.json configs through the trainer CLI on one MI355X, on synthetic
# datasets written in the reference's binary format (scripts/make_synth_dataset.py).
#   config 2: cora GCN 2-layer D=16 on the real cora topology / labels / split (tests/golden/cora) + seeded features
#             (parity against the oracle model: tests/test_gpu_driver.py)
#   config 3: ogbn-products GraphSAGE 3-layer (scripts/run-sage-products.sh: hidden 256, lr 0.01, 10 epochs)
#   config 4: reddit GAT 2-layer hidden 64, 8 heads
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DATA=${1:-/tmp/gaib_data}
mkdir -p "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" ogbn-products "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" reddit "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" cora "$DATA"
export DATASET_PATH="$DATA/"
echo "=== config 2: gpu_train_gcn cora 200 32 softmax 16 0 0 0.01 2 0 50 0"
"$ROOT/bin/gpu_train_gcn" cora 200 32 softmax 16 0 0 0.01 2 0 50 0 | grep -E "Epoch (  0|  1| 50|100|150|199) |Average|Test acc"
echo "=== config 3: gpu_train_sage ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0 (GAIB_SYNC_TIMERS=1)"
GAIB_SYNC_TIMERS=1 "$ROOT/bin/gpu_train_sage" ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0
echo "=== config 3 at the BASELINE width: gpu_train_sage ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0"
"$ROOT/bin/gpu_train_sage" ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0
echo "=== GCN on the same graph: gpu_train_gcn ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0"
"$ROOT/bin/gpu_train_gcn" ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0
echo "=== config 4: GAIB_GAT_HEADS=8 gpu_train_gat reddit 10 32 softmax 64 0 0 0.01 2 0 50 0"
GAIB_GAT_HEADS=8 GAIB_SYNC_TIMERS=1 "$ROOT/bin/gpu_train_gat" reddit 10 32 softmax 64 0 0 0.01 2 0 50 0
rm -rf "$DATA"
