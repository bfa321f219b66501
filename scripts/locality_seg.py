"""The column-segmented fused aggregation (csrc/spmm_seg.hip, option spmm_seg = K) against the one-pass fused kernel on the
planted-locality graph (synth.planted_locality: communities of 16 384 consecutive ids, 90 % of a vertex's edges inside its
community) -- VERDICT r3 #5.  Same edge order per row, so agg and y must be the one-pass kernel's BIT FOR BIT.

    python scripts/locality_seg.py [--k 2 3 4 6] [--sync 0 1 2] [--slack 32] [--reps 6] [--small-only]

First a correctness sweep on small graphs (ragged sizes, heavy rows, blocks smaller than a round), then one JSON line per
(K, sync) on the products-sized planted graph.  Development aid; numbers quoted in DESIGN.md 3.10."""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi, synth  # noqa: E402


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


def small_cases(ctx):
    """bit-identity with the one-pass kernel on graphs that exercise the corners"""
    gen = torch.Generator(device="cuda").manual_seed(7)
    bad = 0
    for nv, avg, block, K, D, Dout, kind in ((5003, 9, 1024, 3, 128, 128, capi.W_GCN), (70001, 20, 4096, 4, 128, 64, capi.W_GCN),
                                             (40000, 30, 16384, 2, 64, 128, capi.W_MEAN), (9000, 14, 512, 5, 100, 47, capi.W_GCN),
                                             (33333, 20, 2048, 3, 32, 16, capi.W_MEAN_T)):
        deg = torch.randint(0, 2 * avg, (nv,), device="cuda", generator=gen)
        deg[::997] = 3000  # heavy rows (threshold 1 024)
        deg[5] = 0
        rp = torch.zeros(nv + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(deg, 0, out=rp[1:])
        ne = int(rp[-1])
        rows = torch.repeat_interleave(torch.arange(nv, device="cuda"), deg)
        # 80 % of the edges near the row's block, the rest anywhere; duplicates are fine for an aggregation
        near = (rows // block) * block + torch.randint(0, block, (ne,), device="cuda", generator=gen)
        far = torch.randint(0, nv, (ne,), device="cuda", generator=gen)
        col = torch.where(torch.rand(ne, device="cuda", generator=gen) < 0.8, near, far).clamp_(max=nv - 1)
        key = rows * nv + col
        col = (torch.sort(key).values % nv).to(torch.int32)
        g = ctx.graph(rp, col)
        x = torch.randn(nv, D, device="cuda", generator=gen)
        W = torch.randn(D, Dout, device="cuda", generator=gen) * 0.1
        agg0, y0 = torch.empty(nv, D, device="cuda"), torch.empty(nv, Dout, device="cuda")
        agg1, y1 = torch.full((nv, D), 7.0, device="cuda"), torch.full((nv, Dout), 7.0, device="cuda")
        ctx.set_option("spmm_seg", 0)
        ctx.spmm_gemm(g, kind, x, agg0, W, y0, relu=True)
        for sync in (0, 1, 2):
            ctx.set_option("spmm_seg", K)
            ctx.set_option("spmm_seg_block", block)
            ctx.set_option("spmm_seg_sync", sync)
            agg1.fill_(7.0)
            y1.fill_(7.0)
            ctx.spmm_gemm(g, kind, x, agg1, W, y1, relu=True)
            torch.cuda.synchronize()
            # (bit patterns: a column without edges has an infinite 1 / degree, and NaN != NaN)
            same = bool(torch.equal(agg0.view(torch.int32), agg1.view(torch.int32)) and torch.equal(y0.view(torch.int32), y1.view(torch.int32)))
            bad += not same
            print(json.dumps(dict(case="small", nv=nv, ne=ne, block=block, K=K, D=D, Dout=Dout, kind=kind, sync=sync, same_bits=same)),
                  flush=True)
        ctx.set_option("spmm_seg", 0)
        g.close()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[2, 3, 4, 6])
    ap.add_argument("--sync", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--slack", type=int, nargs="+", default=[32])
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--small-only", action="store_true")
    args = ap.parse_args()
    ctx = capi.Context(0)
    bad = small_cases(ctx)
    if bad:
        print(json.dumps(dict(error=f"{bad} small cases differ from the one-pass kernel")))
        return 1
    if args.small_only:
        return 0
    D, BLOCK = 128, 16384
    sg = synth.planted_locality("ogbn-products", block=BLOCK, cut=0.1, seed=42, device="cuda", selfloops=True)
    nv = sg.nv
    g = ctx.graph(sg.rowptr, sg.colidx)
    x = torch.randn(nv, D, device="cuda")
    W = torch.randn(D, D, device="cuda") * 0.1
    agg0, y0 = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    agg1, y1 = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    ctx.set_option("spmm_seg", 0)
    t_one = ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg0, W, y0, relu=True), args.reps)
    ctx.set_option("spmm_seg_block", BLOCK)
    for K in args.k:
        for sync in args.sync:
            for slack in (args.slack if sync else [0]):
                ctx.set_option("spmm_seg", K)
                ctx.set_option("spmm_seg_sync", sync)
                ctx.set_option("spmm_seg_slack", slack)
                t = ev_ms(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg1, W, y1, relu=True), args.reps)
                same = bool(torch.equal(agg0, agg1) and torch.equal(y0, y1))
                print(json.dumps(dict(graph=sg.name, nv=nv, ne=g.ne, K=K, sync=sync, slack=slack, one_pass_ms=round(t_one, 3),
                                      segmented_ms=round(t, 3), same_bits=same)), flush=True)
    ctx.set_option("spmm_seg", 0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
