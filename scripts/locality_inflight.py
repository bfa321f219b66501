"""Gathers in flight per CU against the cache-resident gather rate: the plain row kernel (8 waves per SIMD) with 16 and with 8
gathers in flight per wave, next to the fused kernel (4 waves per SIMD, 16 in flight), on planted-locality graphs.
Development aid."""
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
    g = ctx.graph(sg.rowptr, sg.colidx)
    nv = sg.nv
    x = torch.randn(nv, D, device="cuda"); W = torch.randn(D, D, device="cuda") * 0.1
    agg, y = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    res = {}
    ctx.set_option("spmm_xcd_swizzle", 1)
    for u in (0, 8):
        ctx.set_option("spmm_unroll", u)
        res[f"row_kernel_unroll={u or 16}"] = round(ev_ms(lambda: ctx.spmm(g, capi.W_GCN, x, agg)), 3)
    ctx.set_option("spmm_unroll", 0)
    ctx.set_option("spmm_xcd_swizzle", 2)
    res["fused"] = round(ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True)), 3)
    print(json.dumps(dict(block=block, cut=cut, ne=g.ne, ms=res, ps_per_edge={k: round(v * 1e9 / g.ne, 1) for k, v in res.items()})), flush=True)
    g.close(); del sg, x, agg, y
    torch.cuda.empty_cache()
