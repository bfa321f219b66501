"""A/B of the weight-gradient kernels inside the layer step (sgemm_variant 0 = register-resident kernels with interleaved sets, 33 = contiguous K ranges, 30 = LDS kernels, 31 = LDS kernel for the masked form only), GCN and SAGE 128->128
on the products-shaped graph: wall time per step and the HIP-event time of the GEMM launches."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import layers as L, synth
ctx = L.init(0)
sg = synth.make("ogbn-products", device="cuda")
g0 = ctx.graph(sg.rowptr, sg.colidx); g1 = g0.add_selfloop(); g0.close()
nv, D = g1.nv, 128
lg = L.LGraph.adopt(g1)
for kind, nm in ((L.GCN, "gcn"), (L.SAGE, "sage")):
    layer = L.Layer(kind, 1, nv, D, D, lg, act=True)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda")); layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    fo = torch.empty(nv, D, device="cuda"); go = torch.empty(nv, D, device="cuda")
    def step():
        layer.forward(fo); layer.backward(fo, go)
    for rnd in range(2):
        for variant in (0, 33, 30, 31):
            ctx.set_option("sgemm_variant", variant)
            for _ in range(2): step()
            torch.cuda.synchronize(); ctx.prof_reset(); ctx.prof_enable(True); t0 = time.perf_counter()
            for _ in range(8): step()
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8 * 1e3; ctx.prof_enable(False)
            n, ms = ctx.prof_get("sgemm")
            print(nm, "variant", variant, round(el, 3), "ms/step; sgemm", round(ms / 8, 3), "launches", n / 8, flush=True)
    del layer
