import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import perf_guard as pg
from graphaibench_amd import layers as L
for name, graph, d, heads in (("reddit 8x8", "reddit", 64, 8), ("reddit 8x16", "reddit", 128, 8), ("reddit 8x4", "reddit", 32, 8), ("products 1x64", "ogbn-products", 64, 1)):
    print(name, round(pg.layer_step(L.GAT, graph, d, d, True, heads=heads), 3), flush=True)
