"""Experiment: can the 4 MB L2 of an XCD hold what a community's rows gather, if the gather is cut into
COLUMN SEGMENTS?  On the planted-locality graph (synth.planted_locality: communities of 16 384 consecutive ids, 90 % of a
vertex's edges inside its community) the fused aggregation moves 31.9 GB for a 3.0 GB compulsory set: an XCD works through
one community at a time (tile supply in 1 024-tile chunks), but a community's feature rows are 8 MB -- twice its L2.
Here every row's edges are split by the segment of the community their column falls into (K segments of 16 384 / K ids;
edges that leave the community go with the last segment), one graph per segment, and the aggregation runs as K passes over
ALL rows -- pass s gathers from segment s only (2.7 MB at K = 3), continues the partial sums of pass s - 1
(gaib_spmm_ex(GAIB_ACCUMULATE)), the last pass carries the dense product (gaib_spmm_gemm(GAIB_ACCUMULATE)).  Built from
the existing entry points: what it would be worth BEFORE anything is built into the library.

    python scripts/locality_colseg.py [--k 2 3 4] [--reps 6]

Prints one JSON line per K: time of the K passes against the one-pass fused kernel on the same graph, and the distance of the
two results (another summation order).  Development aid; the numbers are quoted in DESIGN.md 3.10.
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi, synth  # noqa: E402

D, BLOCK = 128, 16384


def ev_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[2, 3, 4])
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    ctx = capi.Context(0)
    sg = synth.planted_locality("ogbn-products", block=BLOCK, cut=0.1, seed=42, device="cuda", selfloops=True)
    nv = sg.nv
    rp, ci = sg.rowptr, sg.colidx.to(torch.int64)
    deg = (rp[1:] - rp[:-1])
    rows = torch.repeat_interleave(torch.arange(nv, device="cuda"), deg)
    degf = deg.to(torch.float32)
    vd = torch.where(degf > 0, degf.rsqrt(), torch.zeros_like(degf))
    inv = torch.where(degf > 0, 1.0 / degf, torch.zeros_like(degf))
    g_full = ctx.graph(rp, sg.colidx)
    g_full.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
    x = torch.randn(nv, D, device="cuda")
    W = torch.randn(D, D, device="cuda") * 0.1
    agg = torch.empty(nv, D, device="cuda")
    y_ref = torch.empty(nv, D, device="cuda")
    one = lambda: ctx.spmm_gemm(g_full, capi.W_GCN, x, agg, W, y_ref, relu=True)
    t_one = ev_ms(one, args.reps)
    one()
    agg_ref = agg.clone()
    for K in args.k:
        seg_w = BLOCK // K
        same = (ci // BLOCK) == (rows // BLOCK)
        seg = torch.where(same, torch.clamp((ci % BLOCK) // seg_w, max=K - 1), torch.full_like(ci, K - 1))
        graphs, shares = [], []
        for s in range(K):
            m = seg == s
            cnt = torch.bincount(rows[m], minlength=nv)
            rps = torch.zeros(nv + 1, dtype=torch.int64, device="cuda")
            torch.cumsum(cnt, 0, out=rps[1:])
            g = ctx.graph(rps, ci[m].to(torch.int32).contiguous())
            g.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
            graphs.append(g)
            shares.append(float(m.float().mean()))
        del seg, same
        y = torch.empty(nv, D, device="cuda")

        def passes():
            ctx.spmm(graphs[0], capi.W_GCN, x, agg)
            for s in range(1, K - 1):
                ctx.spmm(graphs[s], capi.W_GCN, x, agg, accumulate=True)
            ctx.spmm_gemm(graphs[K - 1], capi.W_GCN, x, agg, W, y, relu=True, accumulate=True)

        res = {}
        # the row passes with one contiguous range of rows per XCD (an XCD then sits in one community at a time), the fused
        # pass with its XCD-affine 1 024-tile chunks
        for name, opts in (("default", {}), ("xcd-affine", {"spmm_xcd_swizzle": 1, "spmm_tile_xcd": 1024})):
            for k_, v_ in opts.items():
                ctx.set_option(k_, v_)
            try:
                res[name] = ev_ms(passes, args.reps)
            finally:
                if opts:
                    ctx.set_option("spmm_xcd_swizzle", 2)
                    ctx.set_option("spmm_tile_xcd", -1)
        passes()
        err_agg = float((agg - agg_ref).abs().max() / agg_ref.abs().max())
        err_y = float((y - y_ref).abs().max() / y_ref.abs().max())
        print(json.dumps(dict(graph=sg.name, K=K, edge_share_per_segment=[round(v, 3) for v in shares], one_pass_fused_ms=round(t_one, 3),
                              k_passes_ms=res, agg_inf_vs_one_pass=err_agg, y_inf_vs_one_pass=err_y)), flush=True)
        for g in graphs:
            g.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
