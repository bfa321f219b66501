#!/bin/bash
# Run GPU steps one after another on the GPU box: usage  bash scripts/gpu_chain.sh OUTDIR  "name|seconds|command" ...
# Every step runs under its own `timeout -k 10`; its stdout / stderr go to OUTDIR/name.{out,err} (files that grow while it
# runs, so a long step is never silent).  A step that FAILS (a test that does not pass) does not stop the chain; a step that is
# KILLED (exit 124 / 137: its time limit, a hang) does -- nothing is started on a GPU that may be in a bad state.
OUT=$1; shift
mkdir -p "$OUT"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.out" 2> "$OUT/$name.err"
  rc=$?
  echo "   rc=$rc after $(( $(date +%s) - t0 ))s; tail: $(tail -c 300 "$OUT/$name.out" | tr '\n' ' ' | cut -c1-300)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   KILLED: stopping the chain"; exit $rc; fi
done
exit 0
