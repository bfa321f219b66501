#!/bin/bash
# Round 6 counter passes: (a) the narrow dense products at 2.45 M rows -- kernel stats, then
# FETCH_SIZE, WRITE_SIZE and the SQ wave / busy counters, each in a --pmc pass of its own (never with trace domains);
# (b) the GAT backward sweep at shift 0 (real column ids) and shift 8 (every gather from a 0.58 MB window): the SQ
# instruction-mix counters.  The program after `--` is python3 itself.
#   gpurun -- 'bash scripts/profile_gemm_narrow.sh [gemm|gat|all]'  -> gpurun_out/prof_r06/*.json
WHAT=${1:-all}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pmc_pass() {  # name, counters (quoted), program args...
  local name=$1 ctrs=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
  echo "pmc $name rc=$?"
}
if [ "$WHAT" = gemm ] || [ "$WHAT" = all ]; then
  G="$ROOT/scripts/gemm_narrow.py --reps 5"
  timeout 300 python3 $G > "$OUT/gemm_narrow_timing.jsonl" 2> "$OUT/gemm_narrow_timing.err"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gemm_stats" -- python3 $G > "$OUT/gemm_stats.log" 2>&1
  pmc_pass gemm_fetch "FETCH_SIZE" $G
  pmc_pass gemm_write "WRITE_SIZE" $G
  pmc_pass gemm_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" $G
  pmc_pass gemm_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" $G
  ( cd "$ROOT" && python3 scripts/summarize_rocprof.py stats "$OUT/gemm_stats" "$OUT/gemm_narrow_kernel_stats.csv" > "$OUT/gemm_stats_top.txt" 2>&1
    python3 scripts/summarize_rocprof.py pmc "$OUT/gemm_narrow_pmc_raw.json" fetch="$OUT/gemm_fetch" write="$OUT/gemm_write" sq="$OUT/gemm_sq" tcc="$OUT/gemm_tcc"
    # per SHAPE: gemm_narrow.py launches 3 + 5 main kernels per shape, shape after shape (the split-K reduce and the probe dropped)
    python3 scripts/summarize_rocprof.py pmcseq "$OUT/gemm_narrow_pmc_by_shape.json" 8 "splitk_reduce|skinny_reduce|probe|copy_kernel" fetch="$OUT/gemm_fetch" write="$OUT/gemm_write" sq="$OUT/gemm_sq" tcc="$OUT/gemm_tcc" )
  rm -rf "$OUT/gemm_stats" "$OUT/gemm_fetch" "$OUT/gemm_write" "$OUT/gemm_sq" "$OUT/gemm_tcc"
fi
if [ "$WHAT" = gat ] || [ "$WHAT" = all ]; then
  for k in 0 8; do
    pmc_pass gat_sq_a_k$k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "$ROOT/scripts/gat_l2_ceiling.py" $k
    pmc_pass gat_sq_b_k$k "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "$ROOT/scripts/gat_l2_ceiling.py" $k
    pmc_pass gat_sq_c_k$k "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM" "$ROOT/scripts/gat_l2_ceiling.py" $k
    cp "$OUT/gat_sq_a_k$k.log" "$OUT/gat_timing_k$k.jsonl"
    ( cd "$ROOT" && python3 scripts/summarize_rocprof.py pmc "$OUT/gat_bwd_sq_k${k}_raw.json" a="$OUT/gat_sq_a_k$k" b="$OUT/gat_sq_b_k$k" c="$OUT/gat_sq_c_k$k" )
    rm -rf "$OUT/gat_sq_a_k$k" "$OUT/gat_sq_b_k$k" "$OUT/gat_sq_c_k$k"
  done
fi
ls -la "$OUT"
