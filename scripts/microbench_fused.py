"""A/B of gaib_spmm_gemm (aggregation fused with the dense product) against gaib_spmm + gaib_sgemm
on the bench graph.  usage: python scripts/microbench_fused.py [scale]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth  # noqa: E402


def timeit(fn, ctx, n=6):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    ctx = capi.Context(0)
    sg = synth.make("ogbn-products", seed=42, device="cuda", scale=scale)
    g = ctx.graph(sg.rowptr, sg.colidx).add_selfloop()
    n, D = g.nv, 128
    x = torch.randn(n, D, device="cuda")
    agg = torch.empty(n, D, device="cuda")
    y = torch.empty(n, D, device="cuda")
    W = torch.randn(D, D, device="cuda") * 0.1
    print(f"nv={n} ne={g.ne}")
    for fuse in (0, 1):
        ctx.set_option("spmm_fuse", fuse)
        for transW, relu, scratch in ((False, True, False), (True, False, True)):
            t = timeit(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, transW=transW, relu=relu,
                                             agg_scratch=scratch), ctx)
            print(f"fuse={fuse} transW={transW} relu={relu} agg_scratch={scratch}: {t:.3f} ms")
    ctx.set_option("spmm_fuse", 1)
    # second row-local product in the same store (SAGE self term): 128 outputs -> 2-row strips, 48 outputs -> 8-row strips
    for n_out in (128, 48):
        Wn = torch.randn(D, n_out, device="cuda") * 0.1
        Ws = torch.randn(D, n_out, device="cuda") * 0.1
        yo = torch.empty(n, n_out, device="cuda")
        t1 = timeit(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, Wn, yo, relu=True), ctx)
        t2 = timeit(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, Wn, yo, relu=True, rows2=x, W2=Ws), ctx)
        print(f"n_out={n_out}: single product {t1:.3f} ms, with self term {t2:.3f} ms")
    for thr in (1024,):
        ctx.set_option("spmm_heavy_threshold", thr)
        t = timeit(lambda: ctx.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True), ctx)
        print(f"heavy threshold {thr}: fused fwd {t:.3f} ms  {ctx.graph_stats(g)}")
    ctx.set_option("spmm_heavy_threshold", 1024)
    t = timeit(lambda: ctx.spmm(g, capi.W_GCN, x, agg), ctx)
    print(f"plain spmm: {t:.3f} ms")


if __name__ == "__main__":
    main()
