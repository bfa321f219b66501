"""Row form vs edge-stream form (option spmm_flat, spmm_flat_ring) of the fused aggregation on the planted-locality graph and on
the random order.  Development aid."""
import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth

def ev_ms(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

ctx = capi.Context(0)
D = 128
for name in ("planted", "random"):
    if name == "planted":
        sg = synth.planted_locality("ogbn-products", block=16384, cut=0.1, seed=42, device="cuda", selfloops=True)
        g = ctx.graph(sg.rowptr, sg.colidx)
    else:
        sg = synth.make("ogbn-products", seed=42, device="cuda")
        g = ctx.graph(sg.rowptr, sg.colidx).add_selfloop()
    nv = g.nv
    x = torch.randn(nv, D, device="cuda"); W = torch.randn(D, D, device="cuda") * 0.1
    agg, y = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    res = {}
    for flat, ring, pre in ((0, 1, 1), (0, 1, 0), (0, 1, 1), (0, 1, 0), (1, 1, 0), (1, 0, 0)):
        ctx.set_option("spmm_flat", flat); ctx.set_option("spmm_flat_ring", ring); ctx.set_option("spmm_prefetch_ids", pre)
        for scratch in (False, True):
            res.setdefault(f"flat={flat},ring={ring},prefetch_ids={pre},agg_scratch={int(scratch)}", []).append(round(
                ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True, agg_scratch=scratch), 10), 3))
    ctx.set_option("spmm_prefetch_ids", 1)
    ctx.set_option("spmm_flat", -1); ctx.set_option("spmm_flat_ring", -1)
    print(json.dumps(dict(graph=name, ne=g.ne, fused_ms=res)), flush=True)
    del sg, x, agg, y
    torch.cuda.empty_cache()
