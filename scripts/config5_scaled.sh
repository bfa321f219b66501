#!/bin/bash
# BASELINE config 5 (ogbn-papers100M GCN D=128, vertex-partitioned, halo rows + dW all-reduce) end to end at reduced
# scale on ONE GPU: the papers100M-shaped dataset at --scale S (default 0.02: 2.2 M vertices, 65 M edges, 128 features,
# 172 classes) through bin/gpu_train_gcn as 1 process and as R processes (default 5: the box allows 6 GPU processes)
# over the peer-to-peer transport; the two log files must agree.  The real thing needs 8 GPUs.
# usage (GPU box): bash scripts/config5_scaled.sh [scale] [ranks]
S=${1:-0.02}
R=${2:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/config5
mkdir -p $OUT
DATA=/tmp/gaib_data_c5
mkdir -p $DATA
python $ROOT/scripts/make_synth_dataset.py ogbn-papers100M $DATA --scale $S
export DATASET_PATH=$DATA/
ARGS="ogbn-papers100M 8 32 softmax 128 0 0 0.01 2 0 4 0"
echo "=== 1 process: gpu_train_gcn $ARGS"
$ROOT/bin/gpu_train_gcn $ARGS > $OUT/world1.log 2> $OUT/world1.err
grep -E "Epoch|Average|Test acc" $OUT/world1.log
echo "=== $R ranks on one GPU, ONE command: GAIB_RANKS=$R gpu_train_gcn $ARGS  (the launcher starts and supervises the ranks;"
echo "    more ranks than devices: they agree on the peer-to-peer transport by themselves)"
GAIB_RANKS=$R GAIB_COMM_TIMEOUT_S=300 $ROOT/bin/gpu_train_gcn $ARGS > $OUT/world${R}_r0.log 2> $OUT/world${R}.err
rc=$?
echo "ranks exit: $rc"
grep -E "Epoch|Average|Test acc" $OUT/world${R}_r0.log
grep -h "rank .* of" $OUT/world${R}_r0.log | head -$R
python3 - <<P
import re, sys
a = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", open("$OUT/world1.log").read())
b = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", open("$OUT/world${R}_r0.log").read())
ok = len(a) == len(b) > 0 and all(abs(float(x[0]) - float(y[0])) <= 2e-3 and abs(float(x[1]) - float(y[1])) <= 0.01 for x, y in zip(a, b))
print("loss curves agree:", ok, a[-1], b[-1])
sys.exit(0 if ok and $rc == 0 else 1)
P
st=$?
rm -rf $DATA
exit $st
