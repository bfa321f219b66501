"""A/B of the NN / NT GEMM kernels INSIDE a layer step (alternating in one process): SAGE 256 -> 256 forward + backward on
the products-shaped graph, sgemm_variant 40 (LDS-tiled) vs 41 (persistent streaming kernel).  What decides the default."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import layers as L, synth
ctx = L.init(0)
sg = synth.make("ogbn-products", device="cuda")
g0 = ctx.graph(sg.rowptr, sg.colidx)
nv, D = g0.nv, 256
lg = L.LGraph.adopt(g0)
layer = L.Layer(L.SAGE, 1, nv, D, D, lg, act=True)
layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda")); layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
fo = torch.empty(nv, D, device="cuda"); go = torch.empty(nv, D, device="cuda")
def step():
    layer.forward(fo); layer.backward(fo, go)
for rnd in range(3):
    for variant in (40, 0):
        ctx.set_option("sgemm_variant", variant)
        for _ in range(2): step()
        torch.cuda.synchronize(); ctx.prof_reset(); ctx.prof_enable(True); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5 * 1e3; ctx.prof_enable(False)
        n, ms = ctx.prof_get("sgemm")
        print("variant", variant, round(el, 2), "ms/step; sgemm", round(ms / 5, 2), "launches", n / 5, flush=True)
ctx.set_option("sgemm_variant", 0)
