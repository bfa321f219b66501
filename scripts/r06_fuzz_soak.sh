#!/bin/bash
# round 6: every randomised sweep on the final tree (new seeds), one JSON summary line each -> gpurun_out/fuzz_r06/
O=gpurun_out/fuzz_r06
bash scripts/gpu_chain.sh $O \
 "fuzz_spmm|200|python scripts/fuzz_spmm.py 400 61" \
 "fuzz_fused|300|python scripts/fuzz_fused.py 500 62" \
 "fuzz_gat|300|python scripts/fuzz_gat.py 400 63" \
 "fuzz_sgemm|200|python scripts/fuzz_sgemm.py --seconds 120 --seed 64" \
 "fuzz_gat_layer|200|python scripts/fuzz_gat_layer.py --seconds 90 --seed 65" \
 "fuzz_layers|200|python scripts/fuzz_layers.py --seconds 90 --seed 66" \
 "fuzz_aggregation|200|python scripts/fuzz_aggregation.py --seconds 90 --seed 67" \
 "fuzz_rows_and_order|200|python scripts/fuzz_rows_and_order.py --seconds 60 --seed 68" \
 "fuzz_part_ipc|250|python scripts/fuzz_partition.py --seconds 90 --seed 69 --transport ipc" \
 "fuzz_part_rccl|250|python scripts/fuzz_partition.py --seconds 90 --seed 70 --transport fake-rccl" \
 "fuzz_trainer|300|python scripts/fuzz_trainer.py --seconds 120 --seed 71"
for f in $O/*.out; do echo "== $(basename $f .out)"; tail -1 $f | cut -c1-600; done > $O/summary.log
