"""Does a transfer that starts AFTER the persistent fused aggregation find CUs to run on?  (a default for the
CU reserve under RCCL, from a one-GPU A/B with a copy kernel standing in for RCCL's send / recv kernels.)
The fused kernel is one 1 024-thread workgroup per CU that holds all of the CU's registers until its last tile.  A kernel
enqueued on another stream right after it -- the exchange of a partitioned aggregation becomes ready at the same moment as the
interior pass: both wait for the pack -- therefore cannot start anywhere until the fused kernel ends, unless whole CUs were left
free.  Here: the products-shaped graph's fused aggregation on stream A; right behind it, on stream B, a copy of `--mb` MB (what
one exchange receives: 250 MB on the clustered generator, 3.3 GB on the uniform one) by a plain copy kernel.  Per reserve
(CUs left free = option comm_reserve_cus applied through GAIB_OVERLAPS_TRANSFER's code path, i.e. spmm_fuse_cus = 256 - reserve):
when the copy finished, when the aggregation finished, and what the aggregation costs alone on that many CUs.
    python scripts/ab_cu_reserve.py [--mb 250 3300]          (development aid; result in DESIGN.md 3.5 / 3.10)"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth  # noqa: E402

D = 128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, nargs="+", default=[250, 3300])
    ap.add_argument("--reserve", type=int, nargs="+", default=[0, 8, 16, 32, 64])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ca = capi.Context(0, stream=sa.cuda_stream)
    sg = synth.make("ogbn-products", seed=42, device="cuda")
    torch.cuda.synchronize()
    g = ca.graph(sg.rowptr, sg.colidx).add_selfloop()
    n = g.nv
    x = torch.randn(n, D, device="cuda")
    agg = torch.empty(n, D, device="cuda")
    y = torch.empty(n, D, device="cuda")
    W = torch.randn(D, D, device="cuda") * 0.1
    torch.cuda.synchronize()

    def fused():
        ca.spmm_gemm(g, capi.W_GCN, x, agg, W, y, relu=True)

    for mb in args.mb:
        src = torch.empty(mb << 18, device="cuda")  # mb MB of floats
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(2):
                dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sb):
            e0.record(sb)
            dst.copy_(src)
            e1.record(sb)
        torch.cuda.synchronize()
        copy_alone = e0.elapsed_time(e1)
        for reserve in args.reserve:
            ca.set_option("spmm_fuse_cus", 256 - reserve if reserve else 0)
            for _ in range(2):
                fused()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(sa)
            for _ in range(args.reps):
                fused()
            a1.record(sa)
            torch.cuda.synchronize()
            alone = a0.elapsed_time(a1) / args.reps
            t_copy, t_fused = [], []
            for _ in range(args.reps):
                torch.cuda.synchronize()
                start, f_done, c_done = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                start.record(sa)
                sb.wait_event(start)  # both become ready together, the aggregation is ENQUEUED first
                fused()
                f_done.record(sa)
                with torch.cuda.stream(sb):
                    dst.copy_(src)
                    c_done.record(sb)
                torch.cuda.synchronize()
                t_fused.append(start.elapsed_time(f_done))
                t_copy.append(start.elapsed_time(c_done))
            med = lambda v: sorted(v)[len(v) // 2]
            print(json.dumps(dict(copy_mb=mb, copy_alone_ms=round(copy_alone, 3), reserve_cus=reserve, fused_alone_ms=round(alone, 3),
                                  fused_with_copy_ms=round(med(t_fused), 3), copy_done_at_ms=round(med(t_copy), 3),
                                  copy_hidden=bool(med(t_copy) <= med(t_fused)))), flush=True)
        ca.set_option("spmm_fuse_cus", 0)
        del src, dst
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
