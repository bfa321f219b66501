#!/bin/bash
# round 6: rank 0's share of the strong headline (the N = 1 bench graph in N vertex ranges) with the exchange in time slices --
# the K slices consumed one to one (the cost of the pieces), and 4 slices consumed as the library's rule chooses at three link rates
O=${1:-gpurun_out/r06_shard}
S="python scripts/papers_shard.py --strong --mode split --direct-send --steps 5"
bash scripts/gpu_chain.sh $O \
 "w2_same|300|$S --world 2 --pieces 1 2 4 8" \
 "w4_same|300|$S --world 4 --pieces 1 2 4 8" \
 "w8_same|300|$S --world 8 --pieces 1 2 4 8" \
 "w2_rule153|200|$S --world 2 --pieces 4 --consume rule --link-gbs 153" \
 "w2_rule100|200|$S --world 2 --pieces 4 --consume rule --link-gbs 100" \
 "w2_rule75|200|$S --world 2 --pieces 4 --consume rule --link-gbs 75" \
 "w8_rule153|200|$S --world 8 --pieces 4 --consume rule --link-gbs 153" \
 "w8_rule75|200|$S --world 8 --pieces 4 --consume rule --link-gbs 75" \
 "weak8_rule|300|python scripts/papers_shard.py --shape ogbn-products --cut 0.1 --boundary uniform clustered --mode auto --pieces 4 --consume rule --steps 5"
