#!/bin/bash
# Kernel statistics + HBM-traffic counters (separate rocprofv3 passes) of the hidden-256 SAGE layer step on the products
# shape (scripts/run-sage-products.sh's width): the two 128-column K-slab launches of each aggregation, the heavy rows,
# the self-term products and the weight gradients.  GPU box: bash scripts/profile_sage256.sh ; output: gpurun_out/prof_sage256/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_sage256
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/scripts/microbench_layers.py --only 256->256 --steps 3"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
cd $ROOT && python3 scripts/summarize_rocprof.py stats $OUT/stats $OUT/sage256_kernel_stats.csv | head -12 \
  && python3 scripts/summarize_rocprof.py pmc $OUT/sage256_pmc_summary.json fetch=$OUT/fetch write=$OUT/write | head -30
rm -rf $OUT/stats $OUT/fetch $OUT/write
