#!/bin/bash
# Run on the GPU box: per-kernel times and HBM-traffic counters of the 8-head GAT layer (reddit-shaped graph).
# Every rocprofv3 call is bounded (a counter pass that aborts inside the profiler otherwise sits until gpurun's limit).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_gat
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/scripts/microbench_layers.py --only 8_heads --steps 2"
[ "$1" != "pmc" ] && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/scripts/microbench_layers.py --only 8_heads --steps 5 > "$OUT/stats.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $CMD > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $CMD > "$OUT/write.log" 2>&1
cd "$ROOT"
[ "$1" != "pmc" ] && python3 scripts/summarize_rocprof.py stats "$OUT/stats" "$OUT/gat_kernel_stats.csv" | head -16
python3 scripts/summarize_rocprof.py pmc "$OUT/gat_pmc.json" fetch="$OUT/fetch" write="$OUT/write" > /dev/null
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/prof_gat/gat_pmc.json"))
names = set()
for c in d.values(): names |= set(c)
for k in sorted(names):
    if not any(t in k for t in ("softmax", "colsum", "sddmm", "spmm_w64", "spmm_heavy")): continue
    print(k[:80], {c: round(d[c][k]["mean"] / 1e6, 3) for c in d if k in d[c]}, "(GB; FETCH_SIZE x2 on gfx950 for wide reads)")
PY
