#!/bin/bash
# Run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r02 [gcn|gat|all]'
# Per workload: rocprofv3 --kernel-trace --stats of the bench command, then the HBM-traffic counters in their OWN
# passes (FETCH_SIZE and WRITE_SIZE separately, never combined with trace domains), summarised into small files under
# gpurun_out/prof_<round>/ -- copy them into profiles/<round>/.  Every rocprofv3 call is bounded; the program after
# `--` is python3 itself (no env / bash -c hop).
ROUND=${1:-r02}
WHAT=${2:-all}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_one() {  # name, bench args
  local name=$1; shift
  local ARGS="$ROOT/bench.py $* --steps 5 --warmup 2 --no-cpu-baseline --no-locality --sustain-s 0"
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_stats" -- python3 $ARGS > "$OUT/${name}_stats.log" 2>&1
  timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${name}_fetch" -- python3 $ARGS > "$OUT/${name}_fetch.log" 2>&1
  timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${name}_write" -- python3 $ARGS > "$OUT/${name}_write.log" 2>&1
  ( cd "$ROOT" && python3 scripts/summarize_rocprof.py stats "$OUT/${name}_stats" "$OUT/${name}_kernel_stats.csv" \
    && python3 scripts/summarize_rocprof.py pmc "$OUT/${name}_pmc_summary.json" fetch="$OUT/${name}_fetch" write="$OUT/${name}_write" )
  grep -h '"metric"' "$OUT/${name}_stats.log" | tail -1 > "$OUT/${name}_bench_under_rocprof.json"
  rm -rf "$OUT/${name}_stats" "$OUT/${name}_fetch" "$OUT/${name}_write"   # raw traces are large; the summaries are what is kept
}
[ "$WHAT" = gcn ] || [ "$WHAT" = all ] && run_one gcn
[ "$WHAT" = gat ] || [ "$WHAT" = all ] && run_one gat --workload gat-reddit
cd "$ROOT"
if [ "$WHAT" = gcn ] || [ "$WHAT" = all ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- python3 scripts/calibrate_fetch.py > "$OUT/calib_fetch.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -- python3 scripts/calibrate_fetch.py > "$OUT/calib_write.log" 2>&1
  python3 scripts/summarize_rocprof.py pmc "$OUT/calib_pmc.json" fetch="$OUT/calib_fetch" write="$OUT/calib_write" > /dev/null 2>&1
  tail -3 "$OUT/calib_fetch.log" > "$OUT/calib_known_bytes.txt"
  rm -rf "$OUT/calib_fetch" "$OUT/calib_write"
fi
if [ "$WHAT" = gcn ] || [ "$WHAT" = all ]; then
  # the planted-locality leg of the bench record: FETCH_SIZE of the fused kernel on the natural order
  ( cd /tmp && timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/loc_fetch" -- python3 "$ROOT/scripts/locality_study.py" --kernel fused --order degree > "$OUT/locality_fused_fetch.jsonl" 2> "$OUT/locality_fused_fetch.log" )
  python3 scripts/locality_study.py --parse "$OUT/loc_fetch" >> "$OUT/locality_fused_fetch.jsonl"
  rm -rf "$OUT/loc_fetch"
fi
ls -la "$OUT"
