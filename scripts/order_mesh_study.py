"""What a relabelling is worth on a graph WITH geometry: a 3-D grid (6-neighbour stencil + self loops, products-sized:
135^3 = 2.46 M vertices) aggregated at D = 128 in five numberings -- the grid's own (x fastest), a random permutation of it
(what a file without locality looks like), and that permutation re-ordered on the device by gaib_graph_reorder: hubs first
(degree), breadth-first levels (bfs), Cuthill-McKee inside the levels (cm).  Kernel time of gaib_spmm (GCN weights) and of
the fused aggregation + product, the re-ordering's own time, and bit-identity of the outputs once un-permuted.
    python scripts/order_mesh_study.py [--side 135]
The power-law generators of bench.py have no numbering that helps (hubs tie every block to every other);
meshes and road-like graphs are where the order decides."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi  # noqa: E402

D = 128


def grid_csr(side: int, relabel=None, stencil: int = 7):
    """symmetric CSR of the side^3 grid with self loops, rows sorted (stencil 7: face neighbours; 27: the whole 3 x 3 x 3 box);
    relabel: [n] int64 new id of every vertex"""
    n = side ** 3
    ids = torch.arange(n, device="cuda")
    x, y, z = ids % side, (ids // side) % side, ids // (side * side)
    src, dst = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if stencil == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                    continue
                ok = (x + dx >= 0) & (x + dx < side) & (y + dy >= 0) & (y + dy < side) & (z + dz >= 0) & (z + dz < side)
                src.append(ids[ok])
                dst.append(ids[ok] + dx + dy * side + dz * side * side)
    a, b = torch.cat(src), torch.cat(dst)
    if relabel is not None:
        a, b = relabel[a], relabel[b]
    key, _ = torch.sort(a * n + b)
    rows, cols = key // n, (key % n).to(torch.int32)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
    return rowptr, cols


def timed(ctx, call, key, reps=6):
    call()
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(reps):
        call()
    ctx.prof_enable(False)
    n, ms = ctx.prof_get(key)
    nh, mh = ctx.prof_get("spmm_heavy")
    ctx.prof_reset()
    return (ms + mh) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=135)
    ap.add_argument("--stencil", type=int, choices=[7, 27], default=7)
    args = ap.parse_args()
    ctx = capi.Context(0)
    side = args.side
    n = side ** 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    x = torch.randn(n, D, device="cuda", generator=gen)
    W = torch.randn(D, D, device="cuda", generator=gen) * 0.1
    perm = torch.randperm(n, device="cuda", generator=gen)  # new id of grid vertex v
    out = {}

    def measure(name, g, xg, extra=None):
        y, agg, z = torch.empty_like(xg), torch.empty_like(xg), torch.empty_like(xg)
        rec = dict(order=name, stencil=args.stencil, nv=g.nv, ne=g.ne,
                   spmm_ms=timed(ctx, lambda: ctx.spmm(g, capi.W_GCN, xg, y), "spmm_light"),
                   fused_ms=timed(ctx, lambda: ctx.spmm_gemm(g, capi.W_GCN, xg, agg, W, z), "spmm_gemm_fused"),
                   near_frac=ctx.graph_locality(g))
        rec.update(extra or {})
        print(json.dumps(rec), flush=True)
        out[name] = y
        return rec

    rp, ci = grid_csr(side, None, args.stencil)
    g_nat = ctx.graph(rp, ci)
    g_nat.compute_vertex_data()
    measure("grid order", g_nat, x)
    rp, ci = grid_csr(side, perm, args.stencil)
    g_perm = ctx.graph(rp, ci)
    g_perm.compute_vertex_data()
    x_perm = torch.empty_like(x)
    x_perm[perm] = x  # the vertex that was v is now perm[v]
    measure("randomly permuted", g_perm, x_perm)
    y_perm = out["randomly permuted"]
    for name, method in (("permuted, then degree", capi.ORDER_DEGREE), ("permuted, then bfs", capi.ORDER_BFS),
                         ("permuted, then cm", capi.ORDER_CM)):
        ctx.sync()
        t0 = time.perf_counter()
        r, new_of_old, old_of_new = g_perm.reorder(method)
        ctx.sync()
        secs = time.perf_counter() - t0
        xr = x_perm[old_of_new].contiguous()
        measure(name, r, xr, dict(reorder_s=round(secs, 4)))
        same = torch.equal(out[name][new_of_old], y_perm)
        print(json.dumps({"order": name, "rows_bit_identical_to_permuted": bool(same)}), flush=True)
        r.close()


if __name__ == "__main__":
    main()
