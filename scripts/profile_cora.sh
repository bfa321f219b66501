#!/bin/bash
# rocprofv3 --kernel-trace --stats of BASELINE config 2 (cora GCN 2-layer D=16) through the trainer CLI; GAIB_EPOCH_GRAPH=0|1.
# Run on the GPU box: gpurun -- bash scripts/profile_cora.sh ; the summary lands in gpurun_out/prof_cora/cora_kernel_stats.csv
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_cora
mkdir -p $OUT
python $ROOT/scripts/make_synth_dataset.py cora /tmp/gd > /dev/null
export DATASET_PATH=/tmp/gd/
cd /tmp && export TMPDIR=/tmp
export GAIB_EPOCH_GRAPH=${GAIB_EPOCH_GRAPH:-0}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cora_stats -- $ROOT/bin/gpu_train_gcn cora 200 32 softmax 16 0 0 0.01 2 0 500 0 > $OUT/prof_cora.log 2>&1
tail -3 $OUT/prof_cora.log
cd $ROOT && python3 scripts/summarize_rocprof.py stats $OUT/cora_stats $OUT/cora_kernel_stats.csv
rm -rf $OUT/cora_stats
cut -c1-220 $OUT/cora_kernel_stats.csv
