#!/bin/bash
# round 6: the packed-math GAT backward sweep (gat_bwd_fused_pk_kernel) -- parity suites, then timings next to round 5's kernel
bash scripts/gpu_chain.sh ${1:-gpurun_out/r06_gat_pk} \
 "gat_ops|600|python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py tests/test_gpu_fuzz.py -x -q -k 'gat or GAT'" \
 "gat_full|600|python -m pytest tests/test_gpu_fullsize.py -x -q -k 'gat or GAT'" \
 "fuzz_layer|500|python scripts/fuzz_gat_layer.py 60" \
 "ceiling_new|300|python scripts/gat_l2_ceiling.py 0 8" \
 "ceiling_old|300|GAIB_OPTS=gat_bwd_pk=0 python scripts/gat_l2_ceiling.py 0 8" \
 "bench_gat|500|python bench.py --workload gat-reddit" \
 "products_breakdown|400|python scripts/gat_products_breakdown.py"
