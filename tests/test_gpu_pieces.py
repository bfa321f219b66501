"""The halo-column half in PIECES (round 6: gaib_halo_set_pieces / gaib_graph_split_pieces; host/aggregators.cpp halo_half):
an exchange that travels in K time slices, the rank's halo-column graph cut into the K sets of columns that land together,
aggregated piece by piece in accumulate mode.

One process, one GPU: the transport is a resident halo table behind LearningGraph's callback form (the way
scripts/papers_shard.py rehearses a rank), so these tests pin the graph cut and the aggregators' piecewise pass; the two real
transports run it in tests/test_gpu_comm.py.  Criteria:
  * structure: piece k holds exactly the edges whose column lies in piece k's ranges, row by row in the row's order;
  * bits: adding piece 0, 1, ... equals ONE pass over the same halo graph whose rows list their edges piece-major (the
    order the pieces add them in) -- bit for bit on rows below the heavy threshold; with ONE peer the slices are consecutive
    column ranges and that order is the uncut graph's own;
  * the oracle's run on the GLOBAL graph, <= 1e-5.
"""
import numpy as np
import pytest
import torch

from graphaibench_amd import capi, layers as L
from oracle import binding as orc
from test_gpu_classes import KINDS, Shard, dev, feat, make_shard, oracle_rows
from util import LONG_SUM_FLOOR, assert_close, rel_err

pytestmark = pytest.mark.gpu


def piece_ranges(n_cols, peers, K, lib):
    """the halo table's rows [0, n_cols) as `peers` owner segments (ascending ids = grouped by owner), each cut into K slices
    by the library's own arithmetic (gaib_halo_piece_slice) -> [(begin, end, piece)]"""
    import ctypes as C

    bounds = [n_cols * p // peers for p in range(peers + 1)]
    out = []
    for p in range(peers):
        rows = bounds[p + 1] - bounds[p]
        for k in range(K):
            lo, hi = C.c_int64(), C.c_int64()
            capi._check(lib.gaib_halo_piece_slice(rows, K, k, C.byref(lo), C.byref(hi)), "gaib_halo_piece_slice")
            assert (lo.value, hi.value) == (rows * k // K, rows * (k + 1) // K)
            if hi.value > lo.value:
                out.append((bounds[p] + lo.value, bounds[p] + hi.value, k))
    return out


def piece_of_cols(cols, ranges, K):
    pc = np.full(len(cols), -1, np.int64)
    for b, e, k in ranges:
        pc[(cols >= b) & (cols < e)] = k
    assert (pc >= 0).all()
    return pc


def piece_major(rp, ci, ranges, K):
    """the halo graph with every row's edges in (piece, column) order: what the pieces add up in"""
    ci = ci.astype(np.int64)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    order = np.lexsort((ci, piece_of_cols(ci, ranges, K), rows))
    return rp, ci[order].astype(np.int32)


def rows_of(g):
    rp, ci = g.rowptr().cpu().numpy(), g.colidx().cpu().numpy().view(np.uint32)
    return [ci[rp[k]:rp[k + 1]] for k in range(g.nv)]


@pytest.mark.parametrize("peers,K", [(1, 2), (1, 4), (3, 4), (7, 3), (2, 16)])
def test_split_pieces_structure(ctx, peers, K):
    g_o, s = make_shard(ctx)
    nh = len(s.halo)
    ranges = piece_ranges(nh, peers, K, ctx.lib)
    pieces = ctx.split_pieces(s.g_halo, K, ranges)
    assert len(pieces) == K and sum(p.ne for p in pieces) == s.g_halo.ne
    whole = rows_of(s.g_halo)
    for k, p in enumerate(pieces):
        assert p.nv == s.g_halo.nv and p.nc == s.g_halo.nc
        for r, (got, all_) in enumerate(zip(rows_of(p), whole)):
            want = all_[piece_of_cols(all_.astype(np.int64), ranges, K) == k]
            assert np.array_equal(got, want), (k, r)
    # a piece of a row CLASS keeps the class's row map
    cls_pieces = ctx.split_pieces(s.cls["bnd_halo"], K, ranges)
    for p in cls_pieces:
        assert np.array_equal(p.row_map().cpu().numpy(), s.cls["bnd_halo"].row_map().cpu().numpy())
    assert sum(p.ne for p in cls_pieces) == s.cls["bnd_halo"].ne
    for p in pieces + cls_pieces:
        p.close()


def test_split_pieces_refuses_what_it_cannot_place(ctx):
    g_o, s = make_shard(ctx)
    nh = len(s.halo)
    with pytest.raises(capi.GaibError, match="outside every piece"):
        ctx.split_pieces(s.g_halo, 2, [(0, nh // 2, 0)])  # the upper half of the columns belongs to no piece
    with pytest.raises(capi.GaibError, match="overlaps"):
        ctx.split_pieces(s.g_halo, 2, [(0, nh // 2 + 5, 0), (nh // 2, nh, 1)])
    with pytest.raises(capi.GaibError, match="piece"):
        ctx.split_pieces(s.g_halo, 2, [(0, nh, 2)])
    with pytest.raises(capi.GaibError):
        ctx.split_pieces(s.g_halo, 17, [(0, nh, 0)])


@pytest.mark.parametrize("d", [16, 47, 128, 200])
@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("peers,K", [(1, 4), (3, 4), (7, 2)])
def test_pieces_add_up_to_the_piece_major_pass(ctx, d, kind, name, peers, K):
    g_o, s = make_shard(ctx, selfloop=(kind == capi.W_GCN))
    x = feat(g_o.nv, d, 11)
    want = oracle_rows(g_o, kind, x, s.lo, s.hi)
    xo, xh = s.tables(x)
    nh = len(s.halo)
    ranges = piece_ranges(nh, peers, K, ctx.lib)
    # reference: owned-column pass, then ONE pass over the halo graph with the rows' edges piece-major
    rp_m, ci_m = piece_major(s.rp_halo, s.ci_halo, ranges, K)
    if peers == 1:  # consecutive slices of one segment: piece-major IS the column order
        assert np.array_equal(ci_m, s.ci_halo)
    g_m = ctx.graph(rp_m, ci_m, ncols=nh)
    inv = (1.0 / np.diff(g_o.rowptr).astype(np.float64)).astype(np.float32)
    vd = g_o.vertex_data()
    g_m.set_vertex_norm(dev(vd[s.lo:s.hi]), dev(vd[s.halo]), dev(inv[s.halo]), row_inv_deg=dev(inv[s.lo:s.hi]))
    ref = torch.empty(s.n, d, device="cuda")
    ctx.spmm(s.g_own, kind, xo, ref)
    ctx.spmm(g_m, kind, xh, ref, accumulate=True)
    ref = ref.cpu().numpy()
    assert rel_err(ref, want) < 1e-5
    pieces = ctx.split_pieces(s.g_halo, K, ranges)
    out = torch.empty(s.n, d, device="cuda")
    ctx.spmm(s.g_own, kind, xo, out)
    for p in pieces:
        ctx.spmm(p, kind, xh, out, accumulate=True)
    got = out.cpu().numpy()
    light = s.light & np.all([np.diff(p.rowptr().cpu().numpy()) <= 1024 for p in pieces], axis=0)
    assert light.sum() > 0.9 * s.n
    assert np.array_equal(got[light].view(np.uint32), ref[light].view(np.uint32))
    assert rel_err(got, want) < 1e-5
    for p in pieces:
        p.close()
    g_m.close()


class ResidentHalo:
    """LearningGraph's callback transport with the halo table already there: begin / wait_piece / end only count"""

    def __init__(self, table):
        self.table, self.calls = table, []

    def begin(self, length, ptr):
        self.calls.append("begin")

    def wait(self, k):
        self.calls.append(f"wait{k}")
        return self.table.data_ptr()

    def end(self, length):
        self.calls.append("end")
        return self.table.data_ptr()


def _layer_run(lctx, s, g_halo, arch, mode, din, dout, x, gin, want_fwd, tab_fwd, tab_bwd, pieces=None):
    """one layer forward + backward over the shard with a resident halo table (tab_fwd / tab_bwd: the rows an exchange would
    deliver in the forward / backward aggregation); pieces = (K, ranges) or None"""
    vd, inv = s._vd, s._inv
    g_own = lctx.graph(s.rp_own, s.ci_own)  # (a fresh owned-column graph per run: the LGraph owns and frees it)
    g_own.set_vertex_norm(dev(vd[s.lo:s.hi]), dev(vd[s.lo:s.hi]), dev(inv[s.lo:s.hi]), row_inv_deg=dev(inv[s.lo:s.hi]))
    lg = L.LGraph.adopt(g_own)
    tr = ResidentHalo(tab_fwd)
    lg.set_halo(g_halo, tr.begin, tr.end)
    lg.set_partition_mode({"split": L.LGraph.PART_SPLIT, "classes": L.LGraph.PART_CLASSES}[mode])
    if pieces:
        lg.set_halo_pieces(pieces[0], pieces[1], tr.wait)
        lg.set_halo_consumption(pieces[0])  # (the rule would take one piece on a graph this small)
    used, _, _ = lg.partition_mode(din)
    assert L.LGraph.PART_NAMES[used] == mode
    assert lg.halo_pieces(din) == (pieces[0] if pieces else 1)
    layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, s.n, din, dout, lg, True)
    layer.write(L.FEAT_IN, dev(x[s.lo:s.hi]))
    out = torch.full((s.n, dout), float("nan"), device="cuda")
    layer.forward(out)
    L.sync()
    fwd = out.cpu().numpy().copy()
    fwd_calls = list(tr.calls)
    tr.table = tab_bwd
    out.copy_(dev(want_fwd))  # identical relu masks on both sides (the oracle's output)
    layer.write(L.GRAD_IN, dev(gin[s.lo:s.hi]))
    grad_out = torch.full((s.n, din), float("nan"), device="cuda")
    tr.calls.clear()
    layer.backward(out, grad_out)
    L.sync()
    res = dict(fwd=fwd, grad_out=grad_out.cpu().numpy().copy(), W_grad=layer.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy().copy(),
               fwd_calls=fwd_calls, bwd_calls=list(tr.calls))
    layer.close()
    lg.close()
    return res


@pytest.mark.parametrize("arch,din,dout", [("gcn", 128, 128), ("sage", 128, 128), ("gcn", 64, 64), ("gcn", 200, 64)])
@pytest.mark.parametrize("mode", ["split", "classes"])
@pytest.mark.parametrize("peers,K", [(1, 4), (3, 2)])
def test_layer_piece_by_piece_equals_the_piece_major_pass(arch, din, dout, mode, peers, K):
    """a GCN / SAGE layer forward + backward with the halo-column half consumed in K pieces == the same layer over the halo
    graph in piece-major order consumed whole (bit for bit where no row of any pass is heavy), and the GLOBAL oracle.
    128 -> 128 / 64 -> 64: the product rides on the last piece; 200 -> 64: multiply first, the pieces aggregate 64 columns"""
    lctx = L.init(0)
    g_o, s = make_shard(lctx, selfloop=(arch == "gcn"), hub=900)
    s._vd = g_o.vertex_data()
    s._inv = (1.0 / np.diff(g_o.rowptr).astype(np.float64)).astype(np.float32)
    x, gin = feat(g_o.nv, din, 5), feat(g_o.nv, dout, 6)
    lo_ = (orc.GCNLayer if arch == "gcn" else orc.SAGELayer)(1, g_o, din, dout, True)
    want = lo_.forward(x)
    want_go = lo_.backward(gin.copy())
    nh = len(s.halo)
    ranges = piece_ranges(nh, peers, K, lctx.lib)
    # what the exchanges would deliver: forward the halo vertices' input rows (the product's rows where the layer multiplies
    # first), backward their gradient rows behind the d_relu (gcn_layer.cpp:35: masked by the post-activation output)
    tab_fwd = dev(x[s.halo] if din <= dout else (x[s.halo].astype(np.float64) @ lo_.W.astype(np.float64)).astype(np.float32))
    tab_bwd = dev((gin * (want > 0))[s.halo])

    def halo_graph(rp, ci):
        g = lctx.graph(rp, ci, ncols=nh)
        g.set_vertex_norm(dev(s._vd[s.lo:s.hi]), dev(s._vd[s.halo]), dev(s._inv[s.halo]), row_inv_deg=dev(s._inv[s.lo:s.hi]))
        return g

    args = (arch, mode, din, dout, x, gin, want[s.lo:s.hi], tab_fwd, tab_bwd)
    g_m = halo_graph(*piece_major(s.rp_halo, s.ci_halo, ranges, K))
    whole = _layer_run(lctx, s, g_m, *args)
    g_h = halo_graph(s.rp_halo, s.ci_halo)
    piped = _layer_run(lctx, s, g_h, *args, pieces=(K, ranges))
    # the piped run waited piece by piece, the last piece behind end(); the other one only for the end
    assert whole["fwd_calls"] == ["begin", "end"], whole["fwd_calls"]
    for calls in (piped["fwd_calls"], piped["bwd_calls"]):
        assert calls[0] == "begin" and calls[-1] == "end", calls
        assert [c for c in calls if c.startswith("wait")] == [f"wait{k}" for k in range(K - 1)], calls
    rtol = 1e-4 if din <= dout else 2e-4  # (200 -> 64: the table's product was formed on the host in fp64)
    assert_close(piped["fwd"], want[s.lo:s.hi], "forward", floor=LONG_SUM_FLOOR, rtol=rtol)
    assert_close(piped["grad_out"], want_go[s.lo:s.hi], "grad_out", floor=LONG_SUM_FLOOR, rtol=rtol)
    # bits: rows whose every pass is below the heavy threshold (the hub row of this graph is not)
    light = (np.diff(s.rp_own) <= 1024) & (np.diff(s.rp_halo) <= 1024)
    for key in ("fwd", "grad_out"):
        a, b = piped[key], whole[key]
        assert np.isfinite(a).all()
        assert np.array_equal(a[light].view(np.uint32), b[light].view(np.uint32)), key
        assert rel_err(a, b) < 1e-5, key
    assert rel_err(piped["W_grad"], whole["W_grad"]) < 1e-5
    g_m.close()
    g_h.close()


def test_consumption_follows_the_rule_and_divides_the_slices():
    """K slices on the wire are consumed in K' | K pieces: forced, or by the rule -- an exchange priced far above the rank's
    owned-column work is consumed slice by slice, one that the owned-column work covers in one piece (no further pass over the
    partial sums); a forced K' that does not divide K falls to the next divisor"""
    lctx = L.init(0)
    g_o, s = make_shard(lctx, hub=900)
    vd = g_o.vertex_data()
    inv = (1.0 / np.diff(g_o.rowptr).astype(np.float64)).astype(np.float32)
    nh = len(s.halo)
    g_own = lctx.graph(s.rp_own, s.ci_own)
    g_own.set_vertex_norm(dev(vd[s.lo:s.hi]), dev(vd[s.lo:s.hi]), dev(inv[s.lo:s.hi]), row_inv_deg=dev(inv[s.lo:s.hi]))
    g_h = lctx.graph(s.rp_halo, s.ci_halo, ncols=nh)
    g_h.set_vertex_norm(dev(vd[s.lo:s.hi]), dev(vd[s.halo]), dev(inv[s.halo]), row_inv_deg=dev(inv[s.lo:s.hi]))
    lg = L.LGraph.adopt(g_own)
    tr = ResidentHalo(dev(feat(nh, 128, 3)))
    lg.set_halo(g_h, tr.begin, tr.end)
    lg.set_partition_mode(L.LGraph.PART_SPLIT)
    lg.set_halo_pieces(8, piece_ranges(nh, 3, 8, lctx.lib), tr.wait)
    lg.partition_mode(128)

    from util import best_consumption

    def rule(link_rows, K=8, length=128, link_gbs=100.0):
        """LearningGraph::consumption_rule, restated in tests/util.py: the K' | K with the shortest modelled aggregation"""
        return best_consumption(4.0 * length, link_rows * 4.0 * length / (link_gbs * 1e9), len(s.ci_own), len(s.ci_halo), s.n, K)[0]

    seen = set()
    for link_rows in (0, 50, 200, 400, 1000, 5000, 10**9):
        lg.set_halo_link_rows(link_rows)
        assert lg.halo_pieces(128) == rule(link_rows), link_rows
        seen.add(rule(link_rows))
    # nothing to hide -> one piece; a wire comparable to the halo-column half -> several pieces; hours of wire -> one piece again
    # (the pieces would shave the last pass off a wait that is a million times longer)
    assert rule(0) == 1 and rule(10**9) == 1 and max(seen) > 1, seen
    for want, got in ((8, 8), (4, 4), (3, 2), (5, 4), (7, 4), (1, 1), (100, 8)):
        lg.set_halo_consumption(want)
        assert lg.halo_pieces(128) == got, (want, got)
    # an aggregation over the pieces of a forced K' = 2 out of 8 slices waits for slices 3 and (behind end) 7
    x = feat(g_o.nv, 128, 11)
    tr.table = dev(x[s.halo])
    lg.set_halo_consumption(2)
    layer = L.Layer(L.GCN, 1, s.n, 128, 128, lg, True)
    layer.write(L.FEAT_IN, dev(x[s.lo:s.hi]))
    out = torch.empty(s.n, 128, device="cuda")
    tr.calls.clear()
    layer.forward(out)
    L.sync()
    assert tr.calls == ["begin", "wait3", "end"], tr.calls
    lo_ = orc.GCNLayer(1, g_o, 128, 128, True)
    assert_close(out.cpu().numpy(), lo_.forward(x)[s.lo:s.hi], "forward", floor=LONG_SUM_FLOOR)
    layer.close()
    lg.close()
    g_h.close()
