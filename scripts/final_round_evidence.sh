#!/bin/bash
# The round's evidence in one GPU call: the judged bench line (N = 1) with parity and CPU baseline, the GAT bench line,
# the kernel statistics + PMC passes of both (scripts/profile_round.sh), the layer microbenchmarks and the config runs.
# usage (GPU box): bash scripts/final_round_evidence.sh r02 ; everything lands under gpurun_out/evidence_<round>/
ROUND=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/evidence_$ROUND
mkdir -p $OUT
cd $ROOT
echo "== bench.py (N=1)"; python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 600 $OUT/bench_n1.json; echo
echo "== bench.py --workload gat-reddit"; python bench.py --workload gat-reddit > $OUT/bench_gat_reddit.json 2> $OUT/bench_gat.err; tail -c 400 $OUT/bench_gat_reddit.json; echo
echo "== profile_round"; bash scripts/profile_round.sh $ROUND all > $OUT/profile_round.log 2>&1; cp gpurun_out/prof_$ROUND/*.csv gpurun_out/prof_$ROUND/*.json gpurun_out/prof_$ROUND/*.txt $OUT/ 2>/dev/null; ls $OUT
echo "== layers"; python scripts/microbench_layers.py > $OUT/microbench_layers.jsonl 2> $OUT/microbench_layers.err; python - <<'P' > $OUT/microbench_layers_ms_per_step.txt
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for l in open([os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f == "microbench_layers.jsonl"][-1]):
    if l.startswith("{"):
        j = json.loads(l); print(j["layer"], round(j["ms_per_step"], 2))
P
cat $OUT/microbench_layers_ms_per_step.txt
echo "== configs"; bash scripts/run_configs.sh > $OUT/config_runs.log 2>&1; grep -E "===|Average|Test acc" $OUT/config_runs.log
echo "== cora profile"; bash scripts/profile_cora.sh > $OUT/cora_profile.log 2>&1; cp gpurun_out/prof_cora/cora_kernel_stats.csv $OUT/cora_kernel_stats.csv 2>/dev/null; tail -3 $OUT/cora_profile.log
