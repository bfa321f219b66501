"""What is the planted-locality leg bound by?  The one-pass fused aggregation (products shape, D = 128) on planted graphs whose
communities shrink from 16 384 rows (8 MB of feature rows, twice an XCD's L2) to 2 048 (1 MB), and whose share of edges that
LEAVE the community goes from 0.1 to 0: time per edge, with the XCD-affine chunk matched to the community.
    python scripts/locality_ceiling.py        (development aid)"""
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
for block, cut in ((2048, 0.1), (4096, 0.1), (8192, 0.1), (16384, 0.1), (16384, 0.02), (16384, 0.0), (2048, 0.02), (2048, 0.0), (512, 0.0)):
    sg = synth.planted_locality("ogbn-products", block=block, cut=cut, seed=42, device="cuda", selfloops=True)
    g = ctx.graph(sg.rowptr, sg.colidx)
    nv = sg.nv
    x = torch.randn(nv, D, device="cuda"); W = torch.randn(D, D, device="cuda") * 0.1
    agg, y = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    res = {}
    for tx in (block // 16, 1024, 0):
        ctx.set_option("spmm_tile_xcd", tx)
        res[f"tile_xcd={tx}"] = round(ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True)), 3)
    ctx.set_option("spmm_tile_xcd", -1)
    # the unfused pieces on the same graph: the row kernel (8 waves per SIMD, no LDS, no product) and the dense product alone
    rows_ms = {}
    for sw in (2, 1):
        ctx.set_option("spmm_xcd_swizzle", sw)
        rows_ms[f"xcd_swizzle={sw}"] = round(ev_ms(lambda: ctx.spmm(g, capi.W_GCN, x, agg)), 3)
    ctx.set_option("spmm_xcd_swizzle", 2)
    gemm_ms = round(ev_ms(lambda: ctx.sgemm(agg, W, y)), 3)
    best = min(res.values())
    print(json.dumps(dict(block=block, cut=cut, ne=g.ne, one_pass_ms=res, row_kernel_ms=rows_ms, dense_product_ms=gemm_ms, ps_per_edge=round(best * 1e9 / g.ne, 1))), flush=True)
    g.close(); del sg, x, agg, y
    torch.cuda.empty_cache()
