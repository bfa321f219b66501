"""How many CUs does the fused aggregation need, and what do the others buy?  The aggregation is HBM / Infinity-Cache
bound with the matrix pipes 7-14 % busy; a GEMM is MFMA bound.  The fused kernel is one persistent workgroup per CU
(all of a CU's LDS and registers), so with fewer workgroups (option spmm_fuse_cus) whole CUs stay free for a GEMM
enqueued on ANOTHER stream.  Measured here, on the products-shaped graph:
  1. the fused kernel alone at 256 ... 128 workgroups,
  2. fused kernel || GEMM on a second context / stream, against the two in sequence,
at D = 128 (the bench layer: weight gradient 128x128, K = 2.45 M) and D = 256 (SAGE hidden 256: the two K-slab launches
next to the 256x256 weight gradients and the self-term product).
usage: python scripts/ab_cu_share.py [scale]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth  # noqa: E402


def wall(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()  # the aggregation gets the CUs first
    ca = capi.Context(0, stream=sa.cuda_stream)
    cb = capi.Context(0, stream=sb.cuda_stream)
    sg = synth.make("ogbn-products", seed=42, device="cuda", scale=scale)
    torch.cuda.synchronize()  # torch made the arrays on its own stream; the contexts run on theirs
    g = ca.graph(sg.rowptr, sg.colidx).add_selfloop()
    n = g.nv
    print(f"nv={n} ne={g.ne}", flush=True)
    for D in (128, 256):
        x = torch.randn(n, D, device="cuda")
        agg = torch.empty(n, D, device="cuda")
        y = torch.empty(n, D, device="cuda")
        W = torch.randn(D, D, device="cuda") * 0.1
        ga = torch.randn(n, D, device="cuda")
        gb = torch.randn(n, D, device="cuda")
        dW = torch.empty(D, D, device="cuda")
        y2 = torch.empty(n, D, device="cuda")
        torch.cuda.synchronize()

        def fused():
            ca.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True)

        def tn():
            cb.sgemm(ga, gb, dW, transA=True)

        def nn():
            cb.sgemm(ga, W, y2)

        t_tn, t_nn = wall(tn), wall(nn)
        print(f"D={D}: weight gradient alone {t_tn:.3f} ms, streaming product alone {t_nn:.3f} ms", flush=True)
        for cus in (256, 240, 224, 208, 192, 176, 160, 128):
            ca.set_option("spmm_fuse_cus", cus if cus < 256 else 0)
            t_f = wall(fused)

            def both_tn():
                fused()   # the persistent workgroups take their CUs first
                tn()

            def both_nn():
                fused()
                nn()

            def both_all():
                fused()
                tn()
                nn()

            t1, t2, t3 = wall(both_tn), wall(both_nn), wall(both_all)
            print(f"D={D} fused on {cus} CUs: alone {t_f:.3f} ms | with weight gradient {t1:.3f} (serial {t_f + t_tn:.3f}) | "
                  f"with streaming product {t2:.3f} (serial {t_f + t_nn:.3f}) | with both {t3:.3f} "
                  f"(serial {t_f + t_tn + t_nn:.3f})", flush=True)
        ca.set_option("spmm_fuse_cus", 0)
        del x, agg, y, ga, gb, y2


if __name__ == "__main__":
    main()
