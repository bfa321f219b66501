#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/profile_bench.sh'): kernel-trace stats of the bench
# step, then the HBM-traffic counters in their own passes (no trace domains combined with --pmc).
# Everything lands under gpurun_out/prof_r01b/; copy the summaries into profiles/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r01b
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/mfma" -- python3 $ARGS > "$OUT/mfma.log" 2>&1
cd "$ROOT"
python3 scripts/summarize_rocprof.py stats "$OUT/stats" "$OUT/bench_kernel_stats.csv"
python3 scripts/summarize_rocprof.py pmc "$OUT/bench_pmc_summary.json" fetch="$OUT/fetch" write="$OUT/write" mfma="$OUT/mfma"
grep -h '"metric"' "$OUT/stats.log" | tail -1 > "$OUT/bench_under_rocprof.json"
python3 bench.py --steps 10 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
tail -c 600 "$OUT/bench_n1.json"
