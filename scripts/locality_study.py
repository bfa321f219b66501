"""How much of the aggregation's traffic is the vertex ORDER (traffic / b_min = 19 on the
bench graph, whose vertex ids are randomly permuted on purpose)?

A products-sized power-law graph with PLANTED locality (vertices in blocks of `--block` consecutive ids; a share
1 - cut of every vertex's edges stays inside its block) is aggregated (gaib_spmm, GCN weights, D = 128) in three orders:

  natural    block-contiguous ids: what a dataset with community structure in its numbering looks like
  permuted   the same graph under a random relabelling: what bench.py measures (no locality left)
  reordered  the permuted graph relabelled again by an order computed FROM THE PERMUTED GRAPH ALONE at graph-create
             time: reverse Cuthill-McKee (scipy, host) or hubs-first degree order -- rows keep their edge order, so
             every output row is BIT-IDENTICAL to the permuted run's (checked)

Prints one JSON line per order: kernel time (HIP events, gaib_prof), algorithmic GB/s, and -- when run under
`rocprofv3 --pmc FETCH_SIZE` -- scripts/locality_study.py --parse <dir> turns the per-dispatch counters into
traffic / b_min per order.

    python scripts/locality_study.py [--scale 1.0] [--block 16384] [--cut 0.1] [--order rcm|degree]
"""
import argparse
import csv
import glob
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi, synth  # noqa: E402

D = 128
REPS = 5


def planted_graph(nv, nnz, max_deg, block, cut, seed, device="cuda"):
    """symmetric edge list (u, v), duplicate-free, of graphaibench_amd.synth.planted_locality's graph (same draws: bench.py's
    `roofline.planted_locality` leg measures the graph this study's `natural` order is)"""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    w = synth._weights(nv, nnz / nv, max_deg, torch.device(device))
    w = w[torch.randperm(nv, generator=gen, device=device)]  # hubs spread over the blocks
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    m = nnz // 2
    keys = []
    step = 1 << 24
    for s in range(0, m, step):
        k = min(step, m - s)
        u = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, generator=gen, device=device)).clamp_(max=nv - 1)
        b0 = (u // block) * block
        b1 = torch.clamp(b0 + block, max=nv)
        lo = torch.where(b0 > 0, cdf[(b0 - 1).clamp_(min=0)], torch.zeros_like(cdf[b0]))
        hi = cdf[b1 - 1]
        r = torch.rand(k, dtype=torch.float64, generator=gen, device=device)
        local = torch.rand(k, dtype=torch.float64, generator=gen, device=device) >= cut
        target = torch.where(local, lo + r * (hi - lo), r)
        v = torch.searchsorted(cdf, target).clamp_(max=nv - 1)
        keep = u != v
        u, v = u[keep], v[keep]
        keys.append(torch.minimum(u, v) * nv + torch.maximum(u, v))
    key = torch.unique(torch.cat(keys))
    a, b = key // nv, key % nv
    return a, b


def csr_with_selfloops(nv, a, b, relabel=None, device="cuda"):
    """CSR (int64 rowptr, int32 colidx, rows sorted by column) of the symmetric graph + self loops under `relabel`"""
    if relabel is not None:
        a, b = relabel[a], relabel[b]
    i = torch.arange(nv, dtype=torch.int64, device=device)
    key = torch.cat([a * nv + b, b * nv + a, i * nv + i])
    key, _ = torch.sort(key)
    rows = key // nv
    cols = (key - rows * nv).to(torch.int32)
    rowptr = torch.zeros(nv + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.bincount(rows, minlength=nv), 0, out=rowptr[1:])
    return rowptr, cols


def relabel_keep_row_order(rowptr, cols, new_of_old):
    """the same graph with vertex v renamed new_of_old[v]; every row keeps the ORDER of its edges (so its fp32 sum is
    the same sequence of additions): rows are moved whole, column ids renamed in place -- rows are no longer sorted"""
    nv = rowptr.numel() - 1
    old_of_new = torch.empty_like(new_of_old)
    old_of_new[new_of_old] = torch.arange(nv, dtype=new_of_old.dtype, device=new_of_old.device)
    deg = rowptr[1:] - rowptr[:-1]
    ndeg = deg[old_of_new]
    nrp = torch.zeros_like(rowptr)
    torch.cumsum(ndeg, 0, out=nrp[1:])
    # edge k of new row r = edge k of old row old_of_new[r]
    src_start = rowptr[:-1][old_of_new]
    idx = torch.repeat_interleave(src_start - nrp[:-1], ndeg) + torch.arange(int(nrp[-1]), device=rowptr.device)
    ncols = new_of_old[cols[idx].long()].to(torch.int32)
    return nrp, ncols


def order_from_graph(rowptr, cols, how):
    """new id of every vertex, computed from the (permuted) graph alone"""
    nv = rowptr.numel() - 1
    t0 = time.time()
    if how == "degree":
        deg = rowptr[1:] - rowptr[:-1]
        old_of_new = torch.argsort(deg, descending=True, stable=True)
    else:
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee

        m = sp.csr_matrix((np.ones(cols.numel(), np.int8), cols.cpu().numpy(), rowptr.cpu().numpy()), shape=(nv, nv))
        old_of_new = torch.from_numpy(reverse_cuthill_mckee(m, symmetric_mode=True).astype(np.int64)).to(rowptr.device)
    new_of_old = torch.empty(nv, dtype=torch.int64, device=rowptr.device)
    new_of_old[old_of_new] = torch.arange(nv, dtype=torch.int64, device=rowptr.device)
    return new_of_old, time.time() - t0


KERNEL = "w64"
_W = None


def run(ctx, name, rowptr, cols, x, extra=None):
    """KERNEL w64: gaib_spmm (spmm_w64_kernel: rows assigned to waves statically, in order).  fused: gaib_spmm_gemm
    (spmm_gemm_kernel: 16-row tiles taken off a counter in order -- what bench.py's layer runs); `out` is the aggregate."""
    global _W
    g = ctx.graph(rowptr, cols)
    out = torch.empty_like(x)
    if KERNEL == "fused":
        if _W is None:
            _W = torch.randn(D, D, device="cuda") * 0.1
        y = torch.empty_like(x)
        call = lambda: ctx.spmm_gemm(g, capi.W_GCN, x, out, _W, y)
    else:
        call = lambda: ctx.spmm(g, capi.W_GCN, x, out)
    call()  # builds the per-edge weights, warms up
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(REPS):
        call()
    ctx.prof_enable(False)
    nl, ml = ctx.prof_get("spmm_gemm_fused" if KERNEL == "fused" else "spmm_light")
    nh, mh = ctx.prof_get("spmm_heavy")
    ctx.prof_reset()
    nv, ne = g.nv, g.ne
    ms = (ml + mh) / REPS
    alg = ne * (4 * D + 8) + nv * 4 * D + (nv + 1) * 8
    rec = dict(order=name, kernel=KERNEL, nv=nv, ne=ne, D=D, spmm_ms=ms, alg_GBs=alg / ms / 1e6, b_min_bytes=2 * nv * 4 * D + 4 * ne,
               launches_per_order=REPS + 1)
    rec.update(extra or {})
    print(json.dumps(rec), flush=True)
    g.close()
    return out


def parse(d):
    """per-order FETCH_SIZE of the spmm_w64 dispatches (KB, x2 half-count correction) from a rocprofv3 --pmc run"""
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if ("spmm_w64_kernel" in r["Kernel_Name"] or "spmm_gemm_kernel" in r["Kernel_Name"])
            and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = REPS + 1
    names = ["natural", "permuted", "reordered"]
    for k, name in enumerate(names):
        seg = rows[k * per + 1:(k + 1) * per]  # skip the warm-up dispatch
        if seg:
            kb = sum(float(r["Counter_Value"]) for r in seg) / len(seg)
            print(json.dumps({"order": name, "FETCH_SIZE_KB": kb, "fetch_bytes_x2": kb * 1024 * 2}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--block", type=int, default=16384)
    ap.add_argument("--cut", type=float, default=0.1)
    ap.add_argument("--order", choices=["rcm", "degree", "bfs-device", "degree-device"], default="rcm",
                    help="rcm / degree: computed here (scipy on the host / torch); *-device: gaib_graph_reorder, the library's own "
                         "opt-in relabelling (order AND relabelled graph built on the device)")
    ap.add_argument("--kernel", choices=["w64", "fused"], default="w64")
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (gaib_set_option), repeatable")
    ap.add_argument("--parse", default=None)
    args = ap.parse_args()
    global KERNEL
    KERNEL = args.kernel
    if args.parse:
        return parse(args.parse)
    ctx = capi.Context(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    nv0, nnz0, max_deg, _, _ = synth.SHAPES["ogbn-products"]
    nv, nnz = int(nv0 * args.scale), int(nnz0 * args.scale)
    a, b = planted_graph(nv, nnz, min(max_deg, nv // 4), args.block, args.cut, seed=42)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    x_nat = torch.randn(nv, D, device="cuda", generator=gen)
    # natural order
    rp, ci = csr_with_selfloops(nv, a, b)
    out_nat = run(ctx, "natural", rp, ci, x_nat, dict(block=args.block, cut=args.cut))
    # random relabelling (what bench.py's generator does)
    perm = torch.randperm(nv, device="cuda", generator=gen)  # new id of natural vertex v
    rp_p, ci_p = csr_with_selfloops(nv, a, b, relabel=perm)
    del a, b
    x_p = torch.empty_like(x_nat)
    x_p[perm] = x_nat
    out_p = run(ctx, "permuted", rp_p, ci_p, x_p)
    err = (out_p[perm] - out_nat).abs().max().item() / out_nat.abs().max().item()
    del out_nat, x_nat
    # recover an order from the permuted graph alone; rows keep their edge order
    if args.order.endswith("-device"):
        gp = ctx.graph(rp_p, ci_p)
        ctx.sync()
        t0 = time.time()
        gr, new_of_old, _ = gp.reorder(capi.ORDER_BFS if args.order.startswith("bfs") else capi.ORDER_DEGREE)
        ctx.sync()
        secs = time.time() - t0
        rp_r, ci_r = gr.rowptr().clone(), gr.colidx().clone()
        gr.close()
        gp.close()
    else:
        new_of_old, secs = order_from_graph(rp_p, ci_p, args.order)
        rp_r, ci_r = relabel_keep_row_order(rp_p, ci_p, new_of_old)
    x_r = torch.empty_like(x_p)
    x_r[new_of_old] = x_p
    out_r = run(ctx, "reordered", rp_r, ci_r, x_r, dict(method=args.order, ordering_seconds=secs,
                                                        natural_vs_permuted_rel_err=err))
    same = torch.equal(out_r[new_of_old], out_p)
    print(json.dumps({"reordered_rows_bit_identical_to_permuted": bool(same)}), flush=True)
    assert same


if __name__ == "__main__":
    main()
