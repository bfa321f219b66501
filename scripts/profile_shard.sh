#!/bin/bash
# rocprofv3 kernel statistics of ONE rank's shard step (scripts/papers_shard.py: rank 0 of 8, GCN 128 -> 128 forward + backward on
# the vertex-range partition, the exchange replaced by a resident halo table), per generator and in the mode the library's rule
# picks (auto) -- the per-kernel evidence behind DESIGN.md section 6's table.
#   gpurun -- 'bash scripts/profile_shard.sh r04'   -> gpurun_out/prof_shard_<round>/shard_<shape>_<boundary>_kernel_stats.csv
ROUND=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_shard_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for shape in ogbn-products ogbn-papers100M; do
  for boundary in uniform clustered; do
    name=shard_$(echo $shape | sed 's/ogbn-//; s/100M//')_${boundary}
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}" -- python3 "$ROOT/scripts/papers_shard.py" \
      --shape $shape --cut 0.1 --boundary $boundary --mode auto --steps 3 > "$OUT/${name}.jsonl" 2> "$OUT/${name}.err"
    ( cd "$ROOT" && python3 scripts/summarize_rocprof.py stats "$OUT/${name}" "$OUT/${name}_kernel_stats.csv" ) > "$OUT/${name}_top.txt" 2>&1
    rm -rf "$OUT/${name}"
    if [ $shape = ogbn-products ] && [ $boundary = uniform ]; then
      # the N > 1 headline case's dominant kernel (the owned-column pass of the split): its L2 -> fabric bytes, counters in
      # their own passes (never with a trace domain) -> profiles/hbm_traffic.json "partitioned_products_uniform"
      for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 500 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${name}_$ctr" -- python3 "$ROOT/scripts/papers_shard.py" \
          --shape $shape --cut 0.1 --boundary $boundary --mode auto --steps 3 > /dev/null 2> "$OUT/${name}_$ctr.err"
      done
      ( cd "$ROOT" && python3 scripts/summarize_rocprof.py pmc "$OUT/${name}_pmc_summary.json" fetch="$OUT/${name}_FETCH_SIZE" write="$OUT/${name}_WRITE_SIZE" ) > /dev/null 2>&1
      rm -rf "$OUT/${name}_FETCH_SIZE" "$OUT/${name}_WRITE_SIZE"
    fi
  done
done
# round 5: rank 0's share of the N > 1 HEADLINE -- the N = 1 bench graph in 8 vertex ranges (random order, cut 7/8): kernel
# statistics, and the L2 -> fabric bytes of its dominant kernel (the halo-column half: the fused kernel continuing the owned-column
# sums) -> profiles/hbm_traffic.json "partitioned_products_strong"
name=shard_products_strong
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}" -- python3 "$ROOT/scripts/papers_shard.py" \
  --strong --mode auto --steps 3 > "$OUT/${name}.jsonl" 2> "$OUT/${name}.err"
( cd "$ROOT" && python3 scripts/summarize_rocprof.py stats "$OUT/${name}" "$OUT/${name}_kernel_stats.csv" ) > "$OUT/${name}_top.txt" 2>&1
rm -rf "$OUT/${name}"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${name}_$ctr" -- python3 "$ROOT/scripts/papers_shard.py" \
    --strong --mode auto --steps 3 > /dev/null 2> "$OUT/${name}_$ctr.err"
done
( cd "$ROOT" && python3 scripts/summarize_rocprof.py pmc "$OUT/${name}_pmc_summary.json" fetch="$OUT/${name}_FETCH_SIZE" write="$OUT/${name}_WRITE_SIZE" ) > /dev/null 2>&1
rm -rf "$OUT/${name}_FETCH_SIZE" "$OUT/${name}_WRITE_SIZE"
ls -la "$OUT"
