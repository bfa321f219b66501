#!/bin/bash
# round 6, the final tree's records in one GPU call (every step bounded, outputs under gpurun_out/evidence_r06/):
#   the judged N = 1 line (with other_configs), the GAT line, the five epoch records, perf_guard (baseline rewritten),
#   the N = 2 / 4 rehearsals on one device, gat-reddit across 2 ranks
O=gpurun_out/evidence_r06
bash scripts/gpu_chain.sh $O \
 "bench_n1|300|python bench.py" \
 "bench_gat_reddit|200|python bench.py --workload gat-reddit" \
 "epoch_sage256|200|python bench.py --workload epoch-sage-products" \
 "epoch_sage128|200|python bench.py --workload epoch-sage-products --hidden 128" \
 "epoch_gcn|200|python bench.py --workload epoch-gcn-products" \
 "epoch_gat|200|python bench.py --workload epoch-gat-reddit" \
 "epoch_cora|200|python bench.py --workload epoch-gcn-cora" \
 "perf_guard|500|python scripts/perf_guard.py --update" \
 "bench_n2|400|python bench.py --gpus 2" \
 "bench_n4|400|python bench.py --gpus 4" \
 "gat_n2|300|python bench.py --gpus 2 --workload gat-reddit"
cp profiles/perf_baseline.json $O/perf_baseline.json
