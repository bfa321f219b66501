#!/bin/bash
# The fused aggregation + product kernel as a function of the COMMUNITY SIZE of the numbering: planted-locality graphs of the
# products shape (cut 0.1) with blocks of 2 048 .. 65 536 consecutive vertices (1 .. 32 MB of 512-B rows; an XCD's L2 holds 4 MB).
#   gpurun -- 'bash scripts/locality_blocks.sh'    -> gpurun_out/locality_blocks.jsonl
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/locality_blocks.jsonl
: > "$OUT"
for B in 2048 4096 8192 16384 65536; do
  timeout -k 10 200 python3 "$ROOT/scripts/locality_study.py" --kernel fused --order degree-device --block $B 2>/dev/null | grep '"order": "natural"\|"order": "permuted"' >> "$OUT" || exit 1
done
cat "$OUT"
