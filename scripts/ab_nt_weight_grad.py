"""A/B of the non-temporal weight gradients inside the layer step (round 6): sgemm_variant 0 = the masked forms non-temporal (the
default), 39 = without, 29 = the plain form non-temporal too, 28 = the LDS-ring masked 256 x 256 form non-temporal too.  GCN and SAGE layers at 128 and 256 on the products-shaped graph,
alternating in one process: wall time per step and the HIP-event time of the GEMM launches.
    python scripts/ab_nt_weight_grad.py [rounds=3]"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import layers as L, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = L.init(0)
sg = synth.make("ogbn-products", device="cuda")
g0 = ctx.graph(sg.rowptr, sg.colidx); g1 = g0.add_selfloop(); g0.close()
nv = g1.nv
lg = L.LGraph.adopt(g1)
for kind, nm, D in ((L.GCN, "gcn", 128), (L.SAGE, "sage", 128), (L.SAGE, "sage", 256)):
    layer = L.Layer(kind, 1, nv, D, D, lg, act=True)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda")); layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    fo = torch.empty(nv, D, device="cuda"); go = torch.empty(nv, D, device="cuda")
    def step():
        layer.forward(fo); layer.backward(fo, go)
    for rnd in range(rounds):
        for variant in (0, 39, 29, 28):
            ctx.set_option("sgemm_variant", variant)
            for _ in range(2): step()
            torch.cuda.synchronize(); ctx.prof_reset(); ctx.prof_enable(True); t0 = time.perf_counter()
            for _ in range(8): step()
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8 * 1e3; ctx.prof_enable(False)
            n, ms = ctx.prof_get("sgemm")
            print(nm, D, "variant", variant, round(el, 3), "ms/step; sgemm", round(ms / 8, 3), "launches", n / 8, flush=True)
    ctx.set_option("sgemm_variant", 0)
    layer.close(); del layer, fo, go
    torch.cuda.empty_cache()
