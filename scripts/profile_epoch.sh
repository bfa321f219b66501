#!/bin/bash
# rocprofv3 --kernel-trace --stats of whole training epochs through the trainer CLI (BASELINE configs 3 and 4 on the
# synthetic datasets of scripts/make_synth_dataset.py): where an epoch goes beyond the layer kernels.
# Run on the GPU box: gpurun -- bash scripts/profile_epoch.sh r05 ; summaries land in gpurun_out/prof_epoch_<round>/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ROUND=${1:-r05}
OUT=$ROOT/gpurun_out/prof_epoch_$ROUND
mkdir -p $OUT
DATA=/tmp/gaib_data_pe
mkdir -p $DATA
python $ROOT/scripts/make_synth_dataset.py ogbn-products $DATA > /dev/null
python $ROOT/scripts/make_synth_dataset.py reddit $DATA > /dev/null
export DATASET_PATH=$DATA/
cd /tmp && export TMPDIR=/tmp
run() {  # name, binary, args...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_raw -- "$@" > $OUT/${name}.log 2>&1
  grep -E "Epoch   [4-9]|Average" $OUT/${name}.log | tail -3
  ( cd $ROOT && python3 scripts/summarize_rocprof.py stats $OUT/${name}_raw $OUT/${name}_kernel_stats.csv | head -24 )
  rm -rf $OUT/${name}_raw
}
echo "=== config 3 at hidden 128: gpu_train_sage ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0"
run sage128 $ROOT/bin/gpu_train_sage ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0
echo "=== the same with GAIB_CACHE_INPUT_AGG=0 (layer 0 re-aggregates its constant input every epoch, as the reference does)"
GAIB_CACHE_INPUT_AGG=0 run sage128_nocache $ROOT/bin/gpu_train_sage ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0
echo "=== GCN hidden 128"
run gcn128 $ROOT/bin/gpu_train_gcn ogbn-products 10 32 softmax 128 0 0 0.01 3 0 50 0
echo "=== config 3 as scripted (hidden 256): gpu_train_sage ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0"
run sage256 $ROOT/bin/gpu_train_sage ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0
echo "=== config 4: GAIB_GAT_HEADS=8 gpu_train_gat reddit 10 32 softmax 64 0 0 0.01 2 0 50 0"
export GAIB_GAT_HEADS=8
run gat8 $ROOT/bin/gpu_train_gat reddit 10 32 softmax 64 0 0 0.01 2 0 50 0
rm -rf $DATA
