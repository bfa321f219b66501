"""What does the two-table gather of the one-pass form cost?  One rank's rows over [owned | halo] columns (rank 0 of 8, uniform
generator, products-shaped range) aggregated with the dense product
  (a) by gaib_spmm_gemm_2t: the rank's own rows and the halo table as TWO allocations, a scalar select per gather, outputs through
      the row map (csrc/spmm_part.hip), and
  (b) by the plain fused kernel on the same CSR over ONE contiguous [owned | halo] table (a copy nobody wants to pay per step).
Same edges, same order, same bits.  If (a) == (b), what the one-pass step loses against the single-GPU step is the workload (the
gathers reach 8.8 M distinct rows instead of 2.4 M), not the kernel.   python scripts/ab_two_tables.py   (development aid)"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi, dist as gd, synth  # noqa: E402

D, WORLD = 128, 8


def ev_ms(fn, reps=6):
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
    ctx = capi.Context(0)
    rows = synth.block_rows("ogbn-products", 0, WORLD, seed=42, cut_fraction=0.1, device="cuda", selfloops=True)
    nv = rows.n_local
    rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(rows.rowptr, rows.colidx_global, 0, nv)
    del rows
    n_halo = int(halo.numel())
    degf = deg.to(torch.float32)
    vd = torch.where(degf > 0, degf.rsqrt(), torch.zeros_like(degf))
    inv = torch.where(degf > 0, 1.0 / degf, torch.zeros_like(degf))
    pick = torch.randint(0, nv, (n_halo,), device="cuda")
    g_own = ctx.graph(rp_own, ci_own)
    g_own.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
    g_halo = ctx.graph(rp_halo, ci_halo, ncols=n_halo)
    g_halo.set_vertex_norm(vd, vd[pick], inv[pick], row_inv_deg=inv)
    cls = ctx.split_classes(g_own, g_halo, interior=False, bnd_own=False, bnd_halo=False, bnd_full=True, all_boundary=True)
    g2 = cls["bnd_full"]
    # the same CSR as an ordinary rectangular graph over one table
    g1 = ctx.graph(g2.rowptr(), g2.colidx(), ncols=nv + n_halo)
    g1.set_vertex_norm(vd, torch.cat([vd, vd[pick]]), torch.cat([inv, inv[pick]]), row_inv_deg=inv)
    x_own = torch.randn(nv, D, device="cuda")
    x_halo = torch.randn(n_halo, D, device="cuda")
    x_all = torch.cat([x_own, x_halo])
    W = torch.randn(D, D, device="cuda") * 0.1
    agg1, y1 = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    agg2, y2 = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    t_one = ev_ms(lambda: ctx.spmm_gemm(g1, capi.W_GCN, x_all, agg1, W, y1, relu=True))
    t_two = ev_ms(lambda: ctx.spmm_gemm_2t(g2, capi.W_GCN, x_own, x_halo, nv, agg2, W, y2, relu=True))
    t_own = ev_ms(lambda: ctx.spmm_gemm(g_own, capi.W_GCN, x_own, agg1, W, y1, relu=True))
    ctx.spmm_gemm(g1, capi.W_GCN, x_all, agg1, W, y1, relu=True)
    ctx.spmm_gemm_2t(g2, capi.W_GCN, x_own, x_halo, nv, agg2, W, y2, relu=True)
    print(json.dumps(dict(nv=nv, n_halo=n_halo, ne=g2.ne, one_contiguous_table_ms=round(t_one, 3), two_tables_ms=round(t_two, 3),
                          owned_columns_only_ms=round(t_own, 3), ne_owned_columns=g_own.ne,
                          same_bits=bool(torch.equal(agg1, agg2) and torch.equal(y1, y2)))))


if __name__ == "__main__":
    main()
