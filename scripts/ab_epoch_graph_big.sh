ROOT=$(cd "$(dirname "$0")/.." && pwd)
DATA=/tmp/gaib_data_eb; mkdir -p $DATA
python $ROOT/scripts/make_synth_dataset.py ogbn-products $DATA > /dev/null
python $ROOT/scripts/make_synth_dataset.py reddit $DATA > /dev/null
export DATASET_PATH=$DATA/
for m in 0 1; do
echo "== GAIB_EPOCH_GRAPH=$m sage products hidden 128"
GAIB_EPOCH_GRAPH=$m $ROOT/bin/gpu_train_sage ogbn-products 8 32 softmax 128 0 0 0.01 3 0 5 0 2>&1 | grep -E "Epoch   [1567]|Average|Test|recorded|rror"
echo "== GAIB_EPOCH_GRAPH=$m gat reddit 8 heads"
GAIB_GAT_HEADS=8 GAIB_EPOCH_GRAPH=$m $ROOT/bin/gpu_train_gat reddit 8 32 softmax 64 0 0 0.01 2 0 5 0 2>&1 | grep -E "Epoch   [1567]|Average|Test|recorded|rror"
done
rm -rf $DATA
