#!/bin/bash
# Full-size run of bench.py's N = 2 code path on ONE GPU (both ranks on cuda:0).  Backend (first argument):
#   ipc  (default) the exchange behind the C ABI over hipIpc peer-to-peer pull -- the data path the 8-GPU run uses,
#        except that RCCL is replaced by device-to-device copies between the two processes
#   gloo the torch.distributed path through the host
# Validates partition building, halo pack / exchange / accumulate passes and the dW all-reduce end to end at the bench's
# real sizes (2 x 126 M edges; both partition-quality ends).  The timing is meaningless (two ranks share one GPU).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2 LOCAL_RANK=0 GAIB_DIST_BACKEND=${1:-ipc} GAIB_COMM_TIMEOUT_S=300
mkdir -p gpurun_out
RANK=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/two_ranks_r1.log 2>&1 &
P1=$!
RANK=0 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/two_ranks_r0.err | tail -1
wait $P1
echo "rank 1 exit: $?"
grep "\[bench r" gpurun_out/two_ranks_r0.err gpurun_out/two_ranks_r1.log | cut -c1-220
