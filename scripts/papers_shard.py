"""BASELINE.json config 5 on ONE GPU: what one rank computes for an ogbn-papers100M-shaped graph (111 M vertices,
3.2 G stored edges, D = 128) partitioned 8 ways by vertex range -- the shard's GCN hidden layer 128 -> 128 forward +
backward through the same C++ layer code and the same owned-column / halo-column split the multi-GPU path uses
(graphaibench_amd/dist.py), with the all-to-all replaced by a halo table that is already resident.  The send lists
are exact (the graph is symmetric: the rows a peer needs from this rank are the rows of this rank with an edge into
the peer's range), so the pack runs as in the real step; what is NOT measured is the wire time, which is reported as
bytes and as the time 7 xGMI links would need at the guide's 153 GB/s per link.

    python scripts/papers_shard.py [--cut 0.1 0.875] [--rank 0] [--steps 5]

cut = fraction of a rank's edges that leave its vertex range: 0.875 = a random vertex order (7/8 of the neighbours
live elsewhere), 0.1 = a locality-preserving order (METIS-like).  Development aid + the numbers quoted in DESIGN 6.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi, dist as gd, layers as L, synth  # noqa: E402

WORLD, D = 8, 128
XGMI_LINK_GBS = 153.0


def run(ctx, rank, cut, steps, scale, shape="ogbn-papers100M"):
    t0 = time.time()
    # papers: the named graph is the GLOBAL one (each rank owns 1/8 of it); products: bench.py's weak-scaling
    # workload, one products-shaped range per rank
    per_rank = scale / WORLD if shape == "ogbn-papers100M" else scale
    rows = synth.block_rows(shape, rank, WORLD, seed=42, cut_fraction=cut, device="cuda", scale=per_rank,
                            selfloops=True)
    nv = rows.n_local
    lo, hi = rank * nv, (rank + 1) * nv
    rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(rows.rowptr, rows.colidx_global, lo, hi)
    del rows
    n_halo = int(halo.numel())
    # exact send lists by symmetry: (peer, own row) pairs over the halo-column edges
    deg_h = rp_halo[1:] - rp_halo[:-1]
    rows_h = torch.repeat_interleave(torch.arange(nv, device="cuda"), deg_h)
    peer = halo[ci_halo.to(torch.int64)] // nv
    send_key = torch.unique(peer * nv + rows_h)
    del rows_h, peer
    send_idx = (send_key % nv).contiguous()
    send_counts = torch.bincount(send_key // nv, minlength=WORLD).tolist()
    recv_counts = torch.bincount(halo // nv, minlength=WORLD).tolist()
    del send_key
    # normalisers: own rows from their full degrees; halo columns from a stand-in of the same distribution (their
    # owners would send them once at setup; values do not change the timing)
    degf = deg.to(torch.float32)
    vd = torch.where(degf > 0, degf.rsqrt(), torch.zeros_like(degf))
    inv = torch.where(degf > 0, 1.0 / degf, torch.zeros_like(degf))
    pick = torch.randint(0, nv, (max(n_halo, 1),), device="cuda")
    g_own = ctx.graph(rp_own, ci_own)
    g_own.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
    lg = L.LGraph.adopt(g_own)
    g_halo = ctx.graph(rp_halo, ci_halo, ncols=max(n_halo, 1))
    g_halo.set_vertex_norm(vd, vd[pick], inv[pick], row_inv_deg=inv)
    ne_own, ne_halo = int(ci_own.numel()), int(ci_halo.numel())
    del rp_own, ci_own, rp_halo, ci_halo, pick
    halo_table = torch.randn(max(n_halo, 1), D, device="cuda")
    sendbuf = torch.empty(max(send_idx.numel(), 1), D, device="cuda")
    pack_calls = [0]

    # the halo plans pack in SOURCE order (gaib_gather_scatter_rows: a row that several peers list is read once)
    pack_row, pack_slot = torch.sort(send_idx, stable=True)
    pack_slot = pack_slot.contiguous()

    def begin(length, src_ptr):  # pack the rows the 7 peers need (the all-to-all would start here)
        if send_idx.numel():
            capi._check(ctx.lib.gaib_gather_scatter_rows(ctx.h, send_idx.numel(), pack_row.data_ptr(), pack_slot.data_ptr(),
                                                         length, src_ptr, sendbuf.data_ptr()), "gaib_gather_scatter_rows")
            pack_calls[0] += 1

    def end(length):  # (and finish here: the table is resident instead)
        return halo_table.data_ptr()

    lg.set_halo(g_halo, begin, end)
    layer = L.Layer(L.GCN, 1, nv, D, D, lg, act=True)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda"))
    layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    fo = torch.empty(nv, D, device="cuda")
    go = torch.empty(nv, D, device="cuda")
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    def step():
        layer.forward(fo)
        layer.backward(fo, go)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ctx.prof_enable(False)
    exchanges = (pack_calls[0] * steps) // (steps + 2) // steps if send_idx.numel() else 2
    br = {}
    for k in ("spmm_gemm_fused", "spmm_light", "spmm_heavy", "sgemm", "gather_rows"):
        n, t = ctx.prof_get(k)
        if n:
            br[k] = round(t / steps, 3)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        begin(D, layer.ptr(L.FEAT_IN))
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) / 3 * 1e3
    per_link = max(max(send_counts), max(recv_counts)) * D * 4
    out = dict(config="%s-shaped, vertex-range x8, rank %d's shard on one GPU" % (shape, rank), cut_fraction=cut,
               scale=scale, n_own=nv, ne_own_columns=ne_own, ne_halo_columns=ne_halo, n_halo_rows=n_halo,
               send_rows=int(send_idx.numel()), halo_table_gb=n_halo * D * 4 / 1e9,
               exchanges_per_step=exchanges,
               send_gb_per_exchange=send_idx.numel() * D * 4 / 1e9, recv_gb_per_exchange=n_halo * D * 4 / 1e9,
               xgmi_ms_per_exchange_at_153GBs_per_link=per_link / (XGMI_LINK_GBS * 1e9) * 1e3,
               compute_ms_per_step=ms, pack_ms_per_exchange=pack_ms, breakdown_ms_per_step=br,
               gedges_per_s_compute_only=2 * (ne_own + ne_halo) / ms / 1e6, setup_s=round(setup_s, 1),
               hbm_gb_in_use=torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9)
    print(json.dumps(out), flush=True)
    del layer, lg, g_halo, halo_table, sendbuf, fo, go
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cut", type=float, nargs="+", default=[0.1, 0.875])
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the whole graph (CPU-sized smoke runs)")
    ap.add_argument("--shape", default="ogbn-papers100M", choices=["ogbn-papers100M", "ogbn-products"],
                    help="ogbn-products = one rank of bench.py --gpus 8 (weak scaling, a products-shaped range per rank)")
    args = ap.parse_args()
    ctx = L.init(0)
    for cut in args.cut:
        run(ctx, args.rank, cut, args.steps, args.scale, args.shape)


if __name__ == "__main__":
    main()
