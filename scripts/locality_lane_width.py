import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth
def ev_ms(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ctx = capi.Context(0)
D = 128
for block, cut in ((16384, 0.1), (2048, 0.0)):
    sg = synth.planted_locality("ogbn-products", block=block, cut=cut, seed=42, device="cuda", selfloops=True)
    g = ctx.graph(sg.rowptr, sg.colidx); nv = sg.nv
    x = torch.randn(nv, D, device="cuda"); agg = torch.empty(nv, D, device="cuda"); ref = torch.empty(nv, D, device="cuda")
    ctx.set_option("spmm_xcd_swizzle", 1)
    res = {}
    ctx.spmm(g, capi.W_GCN, x, ref)
    for v in (0, 2, 4, 32):
        ctx.set_option("spmm_variant", v)
        res[f"variant={v}"] = round(ev_ms(lambda: ctx.spmm(g, capi.W_GCN, x, agg)), 3)
        ctx.spmm(g, capi.W_GCN, x, agg); res[f"same_bits_{v}"] = bool(torch.equal(agg, ref))
    ctx.set_option("spmm_variant", 0); ctx.set_option("spmm_xcd_swizzle", 2)
    print(json.dumps(dict(block=block, cut=cut, ne=g.ne, ms=res)), flush=True)
    g.close(); del sg, x, agg, ref; torch.cuda.empty_cache()
