#!/bin/bash
# The hit / miss pair behind scripts/gat_l2_ceiling.py: L2 counters of the GAT backward sweep on the real column
# ids (k = 0) and with every column id shifted right by 5 bits (a 4.7 MB window: resident in every XCD's L2), in a --pmc pass of
# their own (never with trace domains); the program after `--` is python3 itself.
#   gpurun -- 'bash scripts/profile_gat_ceiling.sh'  -> gpurun_out/prof_gat_ceiling/gat_ceiling_tcc.json
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_gat_ceiling
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for k in 0 5; do
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/tcc_k$k" -- python3 "$ROOT/scripts/gat_l2_ceiling.py" $k > "$OUT/k$k.jsonl" 2> "$OUT/k$k.err"
  ( cd "$ROOT" && python3 scripts/summarize_rocprof.py pmc "$OUT/k${k}_pmc_summary.json" tcc="$OUT/tcc_k$k" ) > /dev/null 2>&1
  rm -rf "$OUT/tcc_k$k"
done
python3 - "$OUT" <<'P'
import json, sys
from pathlib import Path
out = Path(sys.argv[1])
res = {"what": "gat_bwd_fused_chunk_kernel<16,8,4,true> per launch, reddit shape 8 heads x 8: L2 counters on the real column ids (k = 0) "
               "and with the ids shifted right by 5 bits (every gather from a 4.7 MB window), scripts/profile_gat_ceiling.sh"}
for k in (0, 5):
    j = json.loads((out / f"k{k}_pmc_summary.json").read_text())
    rec = {}
    for ctr in ("TCC_HIT_sum", "TCC_MISS_sum"):
        for name, v in j.get(ctr, {}).items():
            if "gat_bwd_fused_chunk_kernel" in name:
                rec[ctr] = v["mean"]
    if rec:
        rec["l2_hit_fraction"] = rec["TCC_HIT_sum"] / (rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"])
    line = [json.loads(l) for l in (out / f"k{k}.jsonl").read_text().splitlines() if l.startswith("{")]
    if line:
        rec["backward_call_ms"] = line[-1]["backward_ms"]
    res[f"shift_{k}"] = rec
(out / "gat_ceiling_tcc.json").write_text(json.dumps(res, indent=1) + "\n")
print(json.dumps(res, indent=1))
P
