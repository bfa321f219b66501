#!/bin/bash
# BASELINE config 2 (cora GCN 2-layer D=16) and the citeseer / GAT / SAGE variants through the trainer CLI, call by call
# (GAIB_EPOCH_GRAPH=0) against recorded epochs (HIP graphs, =1): average epoch time and the final log lines.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DATA=${1:-/tmp/gaib_data_eg}
mkdir -p "$DATA"
python "$ROOT/scripts/make_synth_dataset.py" cora "$DATA" > /dev/null
export DATASET_PATH="$DATA/"
for arch in gcn sage gat; do
  for mode in 0 1 0 1; do
    echo "=== gpu_train_$arch cora 400 32 softmax 16 0 0 0.01 2 0 500 0   GAIB_EPOCH_GRAPH=$mode"
    GAIB_EPOCH_GRAPH=$mode "$ROOT/bin/gpu_train_$arch" cora 400 32 softmax 16 0 0 0.01 2 0 500 0 2>&1 | grep -E "Epoch (  1|399) |Average|Test acc|recorded"
  done
done
rm -rf "$DATA"
