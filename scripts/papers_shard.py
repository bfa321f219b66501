"""BASELINE.json config 5 on ONE GPU: what one rank computes for an ogbn-papers100M-shaped graph (111 M vertices,
3.2 G stored edges, D = 128) partitioned 8 ways by vertex range -- the shard's GCN hidden layer 128 -> 128 forward +
backward through the same C++ layer code and the same owned-column / halo-column split the multi-GPU path uses
(graphaibench_amd/dist.py), with the all-to-all replaced by a halo table that is already resident.  The send lists
are exact (the graph is symmetric: the rows a peer needs from this rank are the rows of this rank with an edge into
the peer's range), so the pack runs as in the real step; what is NOT measured is the wire time, which is reported as
bytes and as the time 7 xGMI links would need at the guide's 153 GB/s per link.

    python scripts/papers_shard.py [--cut 0.1 0.875] [--rank 0] [--steps 5] [--boundary uniform clustered]
                                   [--mode split classes onepass]

boundary: where the generator puts the cut edges (synth.block_rows) -- anywhere (uniform: nearly every row then has a remote
neighbour) or on a boundary band of the range (clustered: what a METIS / breadth-first partition looks like).
mode: how the rank aggregates (LearningGraph::partition_mode): split = round 3's column split over all rows; classes =
interior rows in one pass, boundary rows by the column split; onepass = interior rows in one pass, boundary rows in one pass
over [owned | halo] after the halo rows arrived; auto = the library's rule.  The record says what of a step can overlap the
exchange (`overlappable_ms`) next to the time 7 xGMI links would need.

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
S_PIECES = [1]  # --pieces
S_CONSUME = "same"  # --consume
XGMI_LINK_GBS = 153.0


MODES = {"split": L.LGraph.PART_SPLIT, "classes": L.LGraph.PART_CLASSES, "onepass": L.LGraph.PART_ONEPASS,
         "onepass_all": L.LGraph.PART_ONEPASS_ALL, "auto": L.LGraph.PART_AUTO}


def strong_rows(ctx, rank, scale):
    """rank's rows of bench.py's N = 1 graph (products shape, seed 42, random vertex order) cut into WORLD vertex ranges: the
    HEADLINE case of `bench.py --gpus 8` since round 5 (north_star's curve).  Ranges are ceil(n / WORLD) rows (the last one
    shorter), as gd.partition_bounds cuts them."""
    sg = synth.make("ogbn-products", seed=42, device="cuda", scale=scale)
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g1 = g0.add_selfloop()
    g0.close()
    del sg
    rp_all, ci_all, n = g1.rowptr(), g1.colidx(), g1.nv
    g1.close()
    per = -(-n // WORLD)
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    e0, e1 = int(rp_all[lo]), int(rp_all[hi])
    rows = synth.BlockRows((rp_all[lo:hi + 1] - e0).contiguous(), ci_all[e0:e1].to(torch.int64).contiguous(), n, hi - lo)
    return rows, per


def run(ctx, rank, cut, steps, scale, shape="ogbn-papers100M", boundary="uniform", band=0.2, modes=("split",), opts="", strong=False,
        direct_send=False):
    t0 = time.time()
    # papers: the named graph is the GLOBAL one (each rank owns 1/8 of it); products: bench.py's weak-scaling
    # workload, one products-shaped range per rank; strong: the N = 1 bench graph itself in WORLD ranges
    per_rank = scale / WORLD if shape == "ogbn-papers100M" else scale
    if strong:
        rows, nv = strong_rows(ctx, rank, scale)  # (nv: the range length that maps a global id to its owner)
        cut, boundary = (WORLD - 1) / WORLD, "strong: the bench graph's random order"
    else:
        rows = synth.block_rows(shape, rank, WORLD, seed=42, cut_fraction=cut, device="cuda", scale=per_rank,
                                selfloops=True, boundary=boundary, band=band)
        nv = rows.n_local
    lo, hi = rank * nv, rank * nv + rows.n_local
    rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(rows.rowptr, rows.colidx_global, lo, hi)
    del rows
    n_halo = int(halo.numel())
    # exact send lists by symmetry: (peer, own row) pairs over the halo-column edges
    per = nv          # rows per range (owner of global id v = v // per)
    nv = hi - lo      # this rank's rows
    deg_h = rp_halo[1:] - rp_halo[:-1]
    rows_h = torch.repeat_interleave(torch.arange(nv, device="cuda"), deg_h)
    peer = halo[ci_halo.to(torch.int64)] // per
    send_key = torch.unique(peer * per + rows_h)
    del rows_h, peer
    send_idx = (send_key % per).contiguous()
    send_counts = torch.bincount(send_key // per, minlength=WORLD).tolist()
    recv_counts = torch.bincount(halo // per, minlength=WORLD).tolist()
    del send_key
    # normalisers: own rows from their full degrees; halo columns from a stand-in of the same distribution (their
    # owners would send them once at setup; values do not change the timing)
    degf = deg.to(torch.float32)
    vd = torch.where(degf > 0, degf.rsqrt(), torch.zeros_like(degf))
    inv = torch.where(degf > 0, 1.0 / degf, torch.zeros_like(degf))
    pick = torch.randint(0, nv, (max(n_halo, 1),), device="cuda")
    ne_own, ne_halo = int(ci_own.numel()), int(ci_halo.numel())
    halo_table = torch.randn(max(n_halo, 1), D, device="cuda")
    setup_s = time.time() - t0
    x_in = torch.randn(nv, D, device="cuda")
    g_in = torch.randn(nv, D, device="cuda")
    shard = dict(locals())  # (one dict for all modes: the first mode leaves its output sample in it)
    for mode in modes:
        for k in (S_PIECES or [1]):
            run_mode(ctx, mode, shard, steps, pieces=k)


def piece_ranges(ctx, recv_counts, K):
    """[(begin, end, piece)] rows of the halo table (grouped by owner rank) that travel in slice k: the library's arithmetic"""
    import ctypes as C

    out, off = [], 0
    for rows in recv_counts:
        for k in range(K):
            lo, hi = C.c_int64(), C.c_int64()
            capi._check(ctx.lib.gaib_halo_piece_slice(int(rows), K, k, C.byref(lo), C.byref(hi)), "gaib_halo_piece_slice")
            if hi.value > lo.value:
                out.append((off + lo.value, off + hi.value, k))
        off += int(rows)
    return out


def modelled_stall_ms(marks, wire_ms):
    """one exchange whose K slices land at (k + 1) / K of wire_ms, against the kernel time between the marks (begin, wait 0,
    ..., end): the compute stream stalls where it reaches a wait before that slice is there.  marks = ms between consecutive
    callbacks: [owned-column pass, piece 0, ..., piece K - 2]; the last piece starts behind end()."""
    K = len(marks)
    t = stall = 0.0
    for k in range(K):
        t += marks[k]          # the work enqueued before wait k (k = K - 1: before end)
        arrive = wire_ms * (k + 1) / K
        if arrive > t:
            stall += arrive - t
            t = arrive
    return stall


def run_mode(ctx, mode, S, steps, pieces=1):
    """one mode on the shard held in S (run()'s locals); pieces: time slices of the exchange (gaib_halo_set_pieces)"""
    (rank, cut, scale, shape, boundary, band, nv, n_halo, ne_own, ne_halo, halo_table, send_idx, send_counts, recv_counts, vd, inv,
     pick, setup_s) = (S[k] for k in ("rank", "cut", "scale", "shape", "boundary", "band", "nv", "n_halo", "ne_own", "ne_halo",
                                      "halo_table", "send_idx", "send_counts", "recv_counts", "vd", "inv", "pick", "setup_s"))
    g_own = ctx.graph(S["rp_own"], S["ci_own"])
    g_own.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
    lg = L.LGraph.adopt(g_own)
    g_halo = ctx.graph(S["rp_halo"], S["ci_halo"], ncols=max(n_halo, 1))
    g_halo.set_vertex_norm(vd, vd[pick], inv[pick], row_inv_deg=inv)
    lg.set_partition_mode(MODES[mode])
    lg.set_halo_link_rows(max(max(send_counts), max(recv_counts)))
    sendbuf = torch.empty(max(send_idx.numel(), 1), D, device="cuda")
    pack_calls = [0]

    # the halo plans pack in SOURCE order (gaib_gather_scatter_rows: a row that several peers list is read once)
    pack_row, pack_slot = torch.sort(send_idx, stable=True)
    pack_slot = pack_slot.contiguous()

    def begin(length, src_ptr):  # pack the rows the 7 peers need (the all-to-all would start here)
        # --direct-send: complete halos over RCCL -- every peer's list is this rank's whole row range and is sent straight from
        # the matrix (comm.hip, round 5): no pack
        if send_idx.numel() and not S.get("direct_send"):
            capi._check(ctx.lib.gaib_gather_scatter_rows(ctx.h, send_idx.numel(), pack_row.data_ptr(), pack_slot.data_ptr(),
                                                         length, src_ptr, sendbuf.data_ptr()), "gaib_gather_scatter_rows")
            pack_calls[0] += 1
        if timing[0]:  # what runs between here and end() is what the exchange can hide under
            windows.append([mark()])

    def mark():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def wait_piece(k):  # slice k has landed (the table is resident: nothing to wait for)
        if timing[0] and windows:
            windows[-1].append(mark())
        return halo_table.data_ptr()

    def end(length):  # (and finish here: the table is resident instead)
        if timing[0] and windows:
            windows[-1].append(mark())
        return halo_table.data_ptr()

    timing, windows = [False], []

    lg.set_halo(g_halo, begin, end)
    if pieces > 1:
        lg.set_halo_pieces(pieces, piece_ranges(ctx, recv_counts, pieces), wait_piece)
        if S_CONSUME != "rule":  # the K slices consumed in K pieces (--consume same) or in a given number; rule: the library's choice
            lg.set_halo_consumption(pieces if S_CONSUME == "same" else int(S_CONSUME))
    mode_used, n_bnd, bnd_edges = lg.partition_mode(D)
    pieces_used = lg.halo_pieces(D)
    layer = L.Layer(L.GCN, 1, nv, D, D, lg, act=True)
    layer.write(L.FEAT_IN, S["x_in"])
    layer.write(L.GRAD_IN, S["g_in"])
    fo = torch.empty(nv, D, device="cuda")
    go = torch.empty(nv, D, device="cuda")
    torch.cuda.synchronize()

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
    timing[0] = True  # one more step with the begin..end windows timed (events would perturb the timed steps)
    step()
    torch.cuda.synchronize()
    timing[0] = False
    overlap_ms = [w[0].elapsed_time(w[-1]) for w in windows]
    marks = [[a.elapsed_time(b) for a, b in zip(w[:-1], w[1:])] for w in windows]  # per exchange: owned pass, piece 0, ...
    exchanges = (pack_calls[0] * steps) // (steps + 2) // steps if send_idx.numel() else 2
    br = {}
    for k in ("spmm_gemm_fused", "spmm_light", "spmm_heavy", "sgemm", "gather_rows", "part_fused", "part_fused_acc", "part_fused_2t",
              "part_light", "part_light_acc", "part_light_2t"):
        n, t = ctx.prof_get(k)
        if n:
            br[k] = round(t / steps, 3)
    ctx.prof_reset()
    # the modes sum a row's terms in different orders, nothing else: a sample of output and input-gradient rows (every 97th row,
    # so rows beyond the 4-GiB mark of the 7-GB matrices of a config-5 rank are among them) against the first mode's
    step()
    torch.cuda.synchronize()
    sample = (fo[::97].clone(), go[::97].clone())
    ref = S.setdefault("_mode_ref", sample)
    agree = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(sample, ref)]
    t0 = time.perf_counter()
    for _ in range(3):
        begin(D, layer.ptr(L.FEAT_IN))
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) / 3 * 1e3
    per_link = max(max(send_counts), max(recv_counts)) * D * 4
    out = dict(config="%s-shaped, vertex-range x%d, rank %d's shard on one GPU" % (shape, WORLD, rank), cut_fraction=cut,
               boundary=boundary, band=band if boundary == "clustered" else None, mode_asked=mode, opts=S.get("opts", ""),
               mode=L.LGraph.PART_NAMES[mode_used], boundary_rows=n_bnd, boundary_row_share=n_bnd / nv,
               boundary_edges=bnd_edges, boundary_edge_share=bnd_edges / (ne_own + ne_halo) if mode_used else None,
               scale=scale, n_own=nv, ne_own_columns=ne_own, ne_halo_columns=ne_halo, n_halo_rows=n_halo,
               send_rows=int(send_idx.numel()), halo_table_gb=n_halo * D * 4 / 1e9,
               exchanges_per_step=exchanges,
               send_gb_per_exchange=send_idx.numel() * D * 4 / 1e9, recv_gb_per_exchange=n_halo * D * 4 / 1e9,
               xgmi_ms_per_exchange_at_153GBs_per_link=per_link / (XGMI_LINK_GBS * 1e9) * 1e3, world=WORLD,
               compute_ms_per_step=ms, pack_ms_per_exchange=pack_ms, breakdown_ms_per_step=br, direct_send=bool(S.get("direct_send")),
               # per exchange: the kernel time between its begin and its end -- what the wire time can hide under
               overlappable_ms_per_exchange=[round(v, 3) for v in overlap_ms],
               # round 6: the exchange in time slices -- kernel time between the callbacks (owned-column pass, piece 0, ...; the
               # last piece runs behind end()), and the step with the wire modelled at a link rate: compute + the stalls of a
               # compute stream that reaches a wait before its slice has landed (slice k lands at (k + 1) / K of the exchange)
               halo_pieces=pieces_used, halo_slices=pieces, consume=S_CONSUME,
               link_gbs_of_the_rules=float(__import__("os").environ.get("GAIB_LINK_GBS", "100")),
               ms_between_waits_per_exchange=[[round(v, 3) for v in m] for m in marks],
               modelled_ms_per_step={str(g): round(ms + sum(modelled_stall_ms(m, per_link / (g * 1e9) * 1e3) for m in marks), 3)
                                     for g in (153.0, 100.0, 75.0)},
               modelled_stall_ms_per_exchange={str(g): [round(modelled_stall_ms(m, per_link / (g * 1e9) * 1e3), 3) for m in marks]
                                               for g in (153.0, 100.0, 75.0)},
               gedges_per_s_compute_only=2 * (ne_own + ne_halo) / ms / 1e6, setup_s=round(setup_s, 1),
               out_and_grad_inf_vs_first_mode=agree,
               hbm_gb_in_use=torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9)
    print(json.dumps(out), flush=True)
    layer.close()
    lg.close()
    del layer, lg, g_halo, sendbuf, fo, go
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cut", type=float, nargs="+", default=[0.1, 0.875])
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the whole graph (CPU-sized smoke runs)")
    ap.add_argument("--shape", default="ogbn-papers100M", choices=["ogbn-papers100M", "ogbn-products"],
                    help="ogbn-products = one rank of bench.py --gpus 8 (weak scaling, a products-shaped range per rank)")
    ap.add_argument("--boundary", nargs="+", default=["uniform"], choices=["uniform", "clustered"])
    ap.add_argument("--band", type=float, default=0.2, help="clustered: share of a range's ids that form its boundary band")
    ap.add_argument("--mode", nargs="+", default=["split", "classes", "onepass", "onepass_all", "auto"], choices=list(MODES))
    ap.add_argument("--world", type=int, default=8, help="number of vertex ranges (default 8: the node)")
    ap.add_argument("--strong", action="store_true",
                    help="the N = 1 bench graph (products shape, random order) cut into 8 vertex ranges: rank's share of the N > 1 headline")
    ap.add_argument("--direct-send", action="store_true",
                    help="no pack: what a rank computes when every peer takes its whole row range straight from the matrix (complete "
                         "halos over the RCCL transport)")
    ap.add_argument("--pieces", type=int, nargs="+", default=[1],
                    help="time slices of the exchange (gaib_halo_set_pieces): the halo-column half is aggregated piece by piece; "
                         "several values = one record each")
    ap.add_argument("--consume", default="same",
                    help="pieces the rank consumes the slices in: same (= --pieces), rule (the library's choice at --link-gbs), or a number")
    ap.add_argument("--link-gbs", type=float, default=None, help="GAIB_LINK_GBS for the auto rule (default: the library's 100)")
    ap.add_argument("--opts", default="", help="library options, key=value,... (gaib_set_option), e.g. spmm_flat_ring=1")
    args = ap.parse_args()
    global WORLD, S_PIECES, S_CONSUME
    WORLD = args.world
    S_PIECES = args.pieces
    S_CONSUME = args.consume
    if args.link_gbs is not None:
        import os
        os.environ["GAIB_LINK_GBS"] = str(args.link_gbs)
    ctx = L.init(0)
    for kv in filter(None, args.opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k.strip(), int(v))
    if args.strong:
        run(ctx, args.rank, 0.875, args.steps, args.scale, "ogbn-products", "uniform", args.band, args.mode, args.opts, strong=True,
            direct_send=args.direct_send)
        return
    for cut in args.cut:
        for boundary in args.boundary:
            run(ctx, args.rank, cut, args.steps, args.scale, args.shape, boundary, args.band, args.mode, args.opts)


if __name__ == "__main__":
    main()
