import json, sys, torch
sys.path.insert(0, '/root/repo')
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
for block in (2048, 4096, 8192, 16384):
    sg = synth.planted_locality("ogbn-products", block=block, cut=0.1, seed=42, device="cuda", selfloops=True)
    g = ctx.graph(sg.rowptr, sg.colidx)
    nv = sg.nv
    x = torch.randn(nv, D, device="cuda"); W = torch.randn(D, D, device="cuda") * 0.1
    agg, y = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    res = {}
    for tx in (block // 16, block // 8, 1024, 0):
        ctx.set_option("spmm_tile_xcd", tx)
        res[f"tile_xcd={tx}"] = round(ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True)), 3)
    ctx.set_option("spmm_tile_xcd", -1)
    print(json.dumps(dict(block=block, ne=g.ne, one_pass_ms=res)), flush=True)
    g.close(); del sg, x, agg, y
    torch.cuda.empty_cache()
