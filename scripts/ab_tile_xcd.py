"""XCD-affine tile supply of the fused aggregation + product kernel (option spmm_tile_xcd = chunk length in tiles, 0 = one
global counter): kernel time on a products-sized graph with planted locality in its natural (block-contiguous) numbering,
on the same graph randomly relabelled, and on bench.py's own graph -- one process, alternating options.
    python scripts/ab_tile_xcd.py [chunk lengths ...]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))
from graphaibench_amd import capi, synth  # noqa: E402
import locality_study as ls  # noqa: E402

D, REPS = 128, 6


def time_fused(ctx, g, x, W, opts):
    out, y = torch.empty_like(x), torch.empty_like(x)
    res = {}
    for v in opts:
        ctx.set_option("spmm_tile_xcd", v)
        ctx.spmm_gemm(g, capi.W_GCN, x, out, W, y)
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(REPS):
            ctx.spmm_gemm(g, capi.W_GCN, x, out, W, y)
        ctx.prof_enable(False)
        n, ms = ctx.prof_get("spmm_gemm_fused")
        nh, mh = ctx.prof_get("spmm_heavy")
        ctx.prof_reset()
        res[v] = round(ms / REPS, 3)
    ctx.set_option("spmm_tile_xcd", 0)
    return res


def main():
    opts = [int(v) for v in sys.argv[1:]] or [0, 16, 128, 512, 1024, 2048, 0]
    ctx = capi.Context(0)
    nv0, nnz0, max_deg, _, _ = synth.SHAPES["ogbn-products"]
    a, b = ls.planted_graph(nv0, nnz0, max_deg, 16384, 0.1, seed=42)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    x = torch.randn(nv0, D, device="cuda", generator=gen)
    W = torch.randn(D, D, device="cuda", generator=gen) * 0.1
    rp, ci = ls.csr_with_selfloops(nv0, a, b)
    g = ctx.graph(rp, ci)
    print(json.dumps({"graph": "planted locality, natural order", "fused_ms_by_chunk_tiles": time_fused(ctx, g, x, W, opts)}), flush=True)
    g.close()
    perm = torch.randperm(nv0, device="cuda", generator=gen)
    rp, ci = ls.csr_with_selfloops(nv0, a, b, relabel=perm)
    del a, b
    g = ctx.graph(rp, ci)
    print(json.dumps({"graph": "planted locality, randomly relabelled", "fused_ms_by_chunk_tiles": time_fused(ctx, g, x, W, opts)}), flush=True)
    g.close()
    del rp, ci
    sg = synth.make("ogbn-products", seed=42, device="cuda")
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g = g0.add_selfloop()
    g0.close()
    print(json.dumps({"graph": "bench.py's graph (random order)", "fused_ms_by_chunk_tiles": time_fused(ctx, g, x[:g.nv].contiguous(), W, opts)}), flush=True)


if __name__ == "__main__":
    main()
