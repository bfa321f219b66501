#!/bin/bash
# Is the headline's slow drift over the rounds (dominant kernel 7.56 -> 7.62 -> 7.64 ms in BENCH_r01..r03) the code or the
# box?  Runs the round-end trees (git worktrees under .drift/, built beforehand with their own build.py) and HEAD on ONE
# box, interleaved (A B C D A B C D), and prints per run the dominant kernel's average launch, the step and the in-run
# stream-copy rate.   usage: gpurun -- bash scripts/drift_ab.sh [rounds]     (development aid; result in DESIGN 3.9)
set -u
cd "$(dirname "$0")/.."
rounds=${1:-2}
for r in $(seq 1 "$rounds"); do
  for t in ${TREES:-.drift/504face .drift/e3a2d00 .drift/9238372 .}; do
    [ -f "$t/bench.py" ] || continue
    flags="--no-cpu-baseline"
    grep -q -- "--no-parity" "$t/bench.py" && flags="$flags --no-parity --sustain-s 0"
    grep -q -- "--no-locality" "$t/bench.py" && flags="$flags --no-locality"
    out=$(cd "$t" && timeout -k 10 200 python bench.py $flags 2>/dev/null | tail -1)
    python3 - "$t" "$r" <<PY
import json, sys
try:
    d = json.loads('''$out''')
    rf = d.get("roofline", {})
    print(json.dumps(dict(tree=sys.argv[1], round=int(sys.argv[2]), ms_per_step=round(d["ms_per_step"], 4),
                          kernel_ms=rf.get("avg_launch_ms"), stream_copy_gbs=rf.get("peak_measured_stream_copy") or rf.get("stream_copy_gbs"))))
except Exception as e:
    print(json.dumps(dict(tree=sys.argv[1], error=str(e)[:200])))
PY
  done
done
