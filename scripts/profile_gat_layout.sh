#!/bin/bash
# the one-sweep GAT backward with three per-vertex tables (h, grad, records) against ONE interleaved
# [h | grad | records] row per vertex (option gat_interleave = 1): kernel time plus L2 hits / misses and L2 -> fabric bytes of
# both layouts, counters in passes of their own (never with trace domains).
#   gpurun -- 'bash scripts/profile_gat_layout.sh'   -> gpurun_out/prof_gat_layout/ (copy the summaries into profiles/r04/)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_gat_layout
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload gat-reddit --steps 5 --warmup 2 --no-cpu-baseline --no-parity --sustain-s 0"
for layout in tables interleaved; do
  if [ $layout = interleaved ]; then export GAIB_OPTS="gat_interleave=1"; else unset GAIB_OPTS; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${layout}_stats" -- python3 $ARGS > "$OUT/${layout}_stats.log" 2>&1
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/${layout}_tcc" -- python3 $ARGS > "$OUT/${layout}_tcc.log" 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${layout}_fetch" -- python3 $ARGS > "$OUT/${layout}_fetch.log" 2>&1
  ( cd "$ROOT" && python3 scripts/summarize_rocprof.py stats "$OUT/${layout}_stats" "$OUT/gat_${layout}_kernel_stats.csv" \
    && python3 scripts/summarize_rocprof.py pmc "$OUT/gat_${layout}_pmc_summary.json" tcc="$OUT/${layout}_tcc" fetch="$OUT/${layout}_fetch" )
  rm -rf "$OUT/${layout}_stats" "$OUT/${layout}_tcc" "$OUT/${layout}_fetch"
done
ls -la "$OUT"
