"""GPU parity suite, layer level: GCN_layer / SAGE_layer / GAT_layer ::forward / backward /
update_weight (the host C++ mirror of include/layers/graph_conv_layer.h over the C ABI) against
the oracle's layer compositions on the same seeded inputs and the same Glorot weights.
Tolerance: the north star's 1e-4 (norm-wise), covering the GEMM summation-order difference."""
from pathlib import Path

import numpy as np
import pytest
import torch

from graphaibench_amd import capi, layers as L
from oracle import binding as orc
from util import LONG_SUM_FLOOR, assert_close, random_graph, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module", autouse=True)
def _ctx():
    return L.init(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def feat(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def cora():
    rp = np.fromfile(GOLD / "cora" / "graph.vertex.bin", np.int64)
    ci = np.fromfile(GOLD / "cora" / "graph.edge.bin", np.uint32)
    return rp, ci


def cora_features(seed=0):
    """sparse-binary, row-normalised like the real cora matrix (F = 1433)"""
    rng = np.random.default_rng(seed)
    x = (rng.random((2708, 1433)) < 0.0127).astype(np.float32)
    x[np.arange(2708), rng.integers(0, 1433, 2708)] = 1.0
    return x / x.sum(1, keepdims=True)


@pytest.mark.parametrize("din,dout,level,act", [(1433, 16, 0, True), (16, 7, 1, False), (16, 33, 1, True), (128, 128, 1, True)])
def test_gcn_layer_cora(din, dout, level, act):
    rp, ci = cora()
    g_o = orc.Graph(rp, ci).add_selfloop()
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=True)
    assert g_d.ne == g_o.ne
    x = cora_features() if din == 1433 else feat(2708, din, 1)
    lo = orc.GCNLayer(level, g_o, din, dout, act)
    ld = L.Layer(L.GCN, level, 2708, din, dout, g_d, act)
    # same initial weights: Glorot seed 1 (golden-pinned against libstdc++)
    assert np.array_equal(ld.tensor(L.W_NEIGH, (din, dout)).cpu().numpy(), lo.W)
    xd = dev(x)
    if level == 0:
        ld.set_feat_in(xd)
    else:
        ld.write(L.FEAT_IN, xd)
    out = torch.empty(2708, dout, device="cuda")
    ld.forward(out)
    want = lo.forward(x)
    assert_close(out.cpu().numpy(), want)
    gin = feat(2708, dout, 2)
    ld.write(L.GRAD_IN, dev(gin))
    grad_out = torch.zeros(2708, din, device="cuda") if level > 0 else None
    ld.backward(out, grad_out)
    gin_o = gin.copy()
    want_go = lo.backward(gin_o)
    assert_close(ld.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), lo.W_grad)
    # d_relu ran in place on grad_in (Q9)
    assert_close(ld.tensor(L.GRAD_IN, (2708, dout)).cpu().numpy(), gin_o)
    if level > 0:
        assert_close(grad_out.cpu().numpy(), want_go)


@pytest.mark.parametrize("din,dout,level", [(100, 128, 0), (128, 128, 1), (128, 47, 2), (47, 128, 1)])
def test_sage_layer_powerlaw(din, dout, level):
    rp, ci = random_graph(4096, 32, seed=5, power_law=True)
    g_o = orc.Graph(rp, ci)
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=False)
    n = g_o.nv
    x = feat(n, din, 3)
    lo = orc.SAGELayer(level, g_o, din, dout, True)
    ld = L.Layer(L.SAGE, level, n, din, dout, g_d, True)
    assert np.array_equal(ld.tensor(L.W_SELF, (din, dout)).cpu().numpy(), lo.W_self)
    xd = dev(x)
    ld.set_feat_in(xd) if level == 0 else ld.write(L.FEAT_IN, xd)
    out = torch.empty(n, dout, device="cuda")
    ld.forward(out)
    assert_close(out.cpu().numpy(), lo.forward(x))
    gin = feat(n, dout, 4)
    ld.write(L.GRAD_IN, dev(gin))
    grad_out = torch.zeros(n, din, device="cuda") if level > 0 else None
    ld.backward(out, grad_out)
    want_go = lo.backward(gin.copy())
    assert_close(ld.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), lo.W_neigh_grad)
    assert_close(ld.tensor(L.W_SELF_GRAD, (din, dout)).cpu().numpy(), lo.W_self_grad)
    if level > 0:
        assert_close(grad_out.cpu().numpy(), want_go)


@pytest.fixture(params=[-1, 1], ids=["staged-bwd", "fused-bwd"])
def gat_bwd_mode(request, _ctx):
    """GAT backward through the staged kernels (what these small graphs get by the auto rule) and through the one-sweep
    kernel (gaib_gat_backward_fused, what dense graphs get at 64 columns)"""
    _ctx.set_option("gat_fused_bwd", request.param)
    _ctx.set_option("gat_fused_fwd", request.param)  # (the one-sweep forward with it: row statistics instead of p)
    yield request.param
    _ctx.set_option("gat_fused_bwd", -1)
    _ctx.set_option("gat_fused_fwd", -1)


@pytest.mark.parametrize("din,dout,level", [(100, 64, 0), (64, 64, 1), (64, 8, 1)])
def test_gat_layer(gat_bwd_mode, din, dout, level):
    rp, ci = random_graph(4096, 24, seed=9, power_law=True, hub_deg=1500)
    g_o = orc.Graph(rp, ci).add_selfloop()
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=True)
    n = g_o.nv
    x = feat(n, din, 3)
    lo = orc.GATLayer(level, g_o, din, dout, True, fast=True)
    ld = L.Layer(L.GAT, level, n, din, dout, g_d, True)
    assert np.array_equal(ld.tensor(L.ALPHA_L, (dout,)).cpu().numpy(), lo.alpha_l)
    assert np.array_equal(ld.tensor(L.ALPHA_R, (dout,)).cpu().numpy(), lo.alpha_r)
    xd = dev(x)
    ld.set_feat_in(xd) if level == 0 else ld.write(L.FEAT_IN, xd)
    out = torch.empty(n, dout, device="cuda")
    ld.forward(out)
    assert_close(out.cpu().numpy(), lo.forward(x))
    assert_close(ld.tensor(L.NORM_SCORES, (g_o.ne,)).cpu().numpy(), lo.norm_scores)
    gin = feat(n, dout, 4)
    ld.write(L.GRAD_IN, dev(gin))
    grad_out = torch.zeros(n, din, device="cuda") if level > 0 else None
    ld.backward(out, grad_out)
    want_go = lo.backward(gin.copy())
    assert_close(ld.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), lo.W_grad)
    assert_close(ld.tensor(L.ALPHA_LGRAD, (dout,)).cpu().numpy(), lo.alpha_lgrad)
    assert_close(ld.tensor(L.ALPHA_RGRAD, (dout,)).cpu().numpy(), lo.alpha_rgrad)
    if level > 0:
        assert_close(grad_out.cpu().numpy(), want_go)


def test_gcn_training_steps_track_oracle():
    """5 full-batch steps of a 2-layer GCN on cora topology (hidden 16): forward, softmax loss,
    backward, shared-Adam update (Q6) -- loss curve and weights vs the oracle."""
    from graphaibench_amd import capi

    ctx = L.init(0)
    rp, ci = cora()
    g_o = orc.Graph(rp, ci).add_selfloop()
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=True)
    n, F, H, Cn = 2708, 1433, 16, 7
    x = cora_features()
    labels = np.fromfile(GOLD / "cora" / "graph.vlabel.bin", np.uint8)
    begin, end = 0, 140
    masks = np.zeros(n, np.uint8)
    masks[begin:end] = 1
    lr = 0.01
    # oracle model
    o0, o1 = orc.GCNLayer(0, g_o, F, H, True), orc.GCNLayer(1, g_o, H, Cn, False)
    oopt = orc.Adam(lr)
    # device model
    d0, d1 = L.Layer(L.GCN, 0, n, F, H, g_d, True, lr), L.Layer(L.GCN, 1, n, H, Cn, g_d, False, lr)
    dopt = L.adam(lr)
    xd, labd, md = dev(x), dev(labels), dev(masks)
    d0.set_feat_in(xd)
    h1 = torch.empty(0)  # layer 1's feat_in is owned by layer 1
    logits = torch.zeros(n, Cn, device="cuda")
    probs = torch.zeros(n, Cn, device="cuda")
    loss_v = torch.zeros(n, device="cuda")
    feat1_ptr = d1.ptr(L.FEAT_IN)
    grad1_ptr = d1.ptr(L.GRAD_IN)
    grad0_ptr = d0.ptr(L.GRAD_IN)
    lib = L.load()
    losses_o, losses_d = [], []
    for step in range(5):
        # oracle
        a1 = o0.forward(x)
        lg = o1.forward(a1)
        p, lv = orc.softmax_xent_fwd(lg, labels, begin, end, masks)
        losses_o.append(orc.masked_avg_loss(lv, begin, end, masks))
        g1 = orc.softmax_xent_bwd(p, labels, begin, end, masks)
        g0 = o1.backward(g1)
        o0.backward(g0)
        oopt.update("w0", o0.W_grad, o0.W)  # one shared optimizer, layer order 0..L-1 (net.cpp:229-234)
        oopt.update("w1", o1.W_grad, o1.W)
        # device: layer0.forward(layer1.feat_in); layer1.forward(logits)
        lib.gaibl_layer_forward(d0.h, feat1_ptr)
        d1.forward(logits)
        ctx.softmax_xent(logits, labd, loss_v, probs, begin, end, md)
        losses_d.append(ctx.masked_avg_loss(loss_v, begin, end, md))
        capi._check(ctx.lib.gaib_fill_f32(ctx.h, n * Cn, 0.0, grad1_ptr), "fill")
        capi._check(ctx.lib.gaib_d_softmax_xent(ctx.h, Cn, begin, end, md.data_ptr(), labd.data_ptr(),
                                                probs.data_ptr(), grad1_ptr), "d_softmax_xent")
        lib.gaibl_layer_backward(d1.h, logits.data_ptr(), grad0_ptr)
        lib.gaibl_layer_backward(d0.h, feat1_ptr, None)
        d0.update_weight(dopt)
        d1.update_weight(dopt)
    assert np.allclose(losses_o, losses_d, rtol=1e-4, atol=1e-5), (losses_o, losses_d)
    assert losses_o[-1] < losses_o[0]
    assert rel_err(d0.tensor(L.W_NEIGH, (F, H)).cpu().numpy(), o0.W) < 1e-3
    assert rel_err(d1.tensor(L.W_NEIGH, (H, Cn)).cpu().numpy(), o1.W) < 1e-3


def test_gat_layer_8_heads(gat_bwd_mode):
    """GAT_layer with 8 attention heads (GAT_Aggregator::set_num_heads): forward/backward equal 8
    single-head oracles on the 8-column slices (BASELINE config 4 shape: hidden 64 = 8 x 8)."""
    rp, ci = random_graph(4096, 24, seed=19, power_law=True, hub_deg=1500)
    g_o = orc.Graph(rp, ci).add_selfloop()
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=True)
    n, din, dout, H = g_o.nv, 100, 64, 8
    x = feat(n, din, 3)
    gin = feat(n, dout, 4)
    ld = L.Layer(L.GAT, 1, n, din, dout, g_d, True)
    ld.set_heads(H)
    W = orc.init_glorot(din, dout, 1)
    al, ar = orc.init_glorot(dout, 1, 2).ravel(), orc.init_glorot(dout, 1, 3).ravel()
    ld.write(L.FEAT_IN, dev(x))
    out = torch.empty(n, dout, device="cuda")
    ld.forward(out)
    hfeat = orc.matmul(x, W)
    agg, temp, scores, norm = orc.gat_aggregate_mh(g_o, hfeat, al, ar, H)
    want = orc.relu(agg)
    assert_close(out.cpu().numpy(), want)
    ld.write(L.GRAD_IN, dev(gin))
    grad_out = torch.zeros(n, din, device="cuda")
    ld.backward(out, grad_out)
    g_act = orc.d_relu(gin, want)
    T, ds, ng, lg, rg = orc.gat_d_aggregate_mh(g_o, hfeat, g_act, norm, temp, H)
    assert_close(grad_out.cpu().numpy(), orc.matmul(T, W, False, True))
    assert_close(ld.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), orc.matmul(x, T, True, False))
    assert_close(ld.tensor(L.ALPHA_LGRAD, (dout,)).cpu().numpy(), lg)
    assert_close(ld.tensor(L.ALPHA_RGRAD, (dout,)).cpu().numpy(), rg)


@pytest.mark.parametrize("kind,din,dout", [("gcn", 100, 128), ("sage", 100, 128), ("sage", 64, 64), ("gcn", 128, 47)])
def test_constant_input_keeps_the_aggregate(kind, din, dout):
    """set_input_constant (layer 0 of a full-batch run): the first forward aggregates, later ones run only the dense
    product(s) on the kept A.X -- same outputs and gradients as the layer that re-aggregates every time, also after a
    weight update, and set_feat_in drops the kept aggregate.  (128 -> 47 multiplies first: nothing to keep, flag inert.)"""
    rp, ci = random_graph(3000, 24, seed=11, power_law=True)
    sl = kind == "gcn"
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=sl)
    n = 3000
    K = L.GCN if kind == "gcn" else L.SAGE
    a, b = L.Layer(K, 0, n, din, dout, g_d, True), L.Layer(K, 0, n, din, dout, g_d, True)
    b.set_input_constant(True)
    x = dev(feat(n, din, 3))
    a.set_feat_in(x)
    b.set_feat_in(x)
    oa, ob = torch.empty(n, dout, device="cuda"), torch.empty(n, dout, device="cuda")
    opt_a, opt_b = L.adam(0.01), L.adam(0.01)
    for step in range(3):
        a.forward(oa)
        b.forward(ob)
        assert_close(ob.cpu().numpy(), oa.cpu().numpy())
        gin = dev(feat(n, dout, 20 + step))
        for l, o in ((a, oa), (b, ob)):
            l.write(L.GRAD_IN, gin)
            l.backward(o, None)
        assert_close(b.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), a.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy())
        a.update_weight(opt_a)
        b.update_weight(opt_b)
    x2 = dev(feat(n, din, 4))  # another input: set_feat_in must make the layer aggregate again
    a.set_feat_in(x2)
    b.set_feat_in(x2)
    a.forward(oa)
    b.forward(ob)
    assert_close(ob.cpu().numpy(), oa.cpu().numpy())


@pytest.mark.parametrize("heads", [1, 8])
def test_gat_attention_dropout(heads):
    """score_drop > 0 (GAT_Aggregator attn_drop): a training forward masks and rescales the normalised attention with
    the library's counter RNG (the reference's CUDA path: graph_operations.h:326-331; its OpenMP path has the call
    commented out) and backward goes through the SAME mask.  Checked with the masks read back: forward = the oracle's
    attention . mask . scale aggregated; backward = the formulas of gat_aggregator.cpp:99-200 with d(out)/d(p_e) masked
    and rescaled (d_dropout, graph_operations.h:376-377) and the gradient flowing back along the dropped attention.
    Rate and phase: a test-phase forward drops nothing and equals the oracle; two training forwards draw different masks;
    the mask rate is the requested one."""
    rate = 0.3
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    rp, ci = random_graph(3000, 20, seed=31, power_law=True, hub_deg=1200)
    g_o = orc.Graph(rp, ci).add_selfloop()
    g_d = L.LGraph.from_host(rp, ci, add_selfloop=True)
    n, ne, din, d = g_o.nv, g_o.ne, 48, 64
    dh = d // heads
    x, gin = feat(n, din, 3), feat(n, d, 4)
    ld = L.Layer(L.GAT, 1, n, din, d, g_d, True, score_drop=rate)
    if heads > 1:
        ld.set_heads(heads)
    W = orc.init_glorot(din, d, 1)
    al, ar = orc.init_glorot(d, 1, 2).ravel(), orc.init_glorot(d, 1, 3).ravel()
    hfeat = orc.matmul(x, W)
    agg, temp, _, norm = orc.gat_aggregate_mh(g_o, hfeat, al, ar, heads)
    norm, temp = norm.reshape(ne, heads), temp.reshape(ne, heads)
    ld.write(L.FEAT_IN, dev(x))
    out = torch.empty(n, d, device="cuda")
    # test phase: nothing is dropped
    ld.set_phase(1)
    ld.forward(out)
    assert_close(out.cpu().numpy(), orc.relu(agg), "test-phase forward", floor=LONG_SUM_FLOOR)
    # training phase
    ld.set_phase(0)
    ld.forward(out)

    def masks_now():
        raw = torch.empty(ne * heads, dtype=torch.uint8, device="cuda")
        capi._check(capi.load().gaib_memcpy_d2d(L.load().gaibl_ctx(), raw.data_ptr(), ld.ptr(L.ATTN_MASKS), ne * heads), "d2d")
        L.sync()
        return raw.cpu().numpy().reshape(ne, heads)

    m = masks_now()
    assert set(np.unique(m)) <= {0, 1} and abs(1.0 - m.mean() - rate) < 0.01
    p_drop = (norm * m * scale).astype(np.float32)
    assert_close(ld.tensor(L.NORM_SCORES_DROPPED, (ne, heads)).cpu().numpy(), p_drop, "dropped attention")
    rows = np.repeat(np.arange(n), np.diff(g_o.rowptr))
    col = g_o.colidx.astype(np.int64)
    want = np.zeros((n, d))
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        np.add.at(want[:, sl], rows, p_drop[:, k:k + 1].astype(np.float64) * hfeat[col, sl])
    want = np.maximum(want, 0).astype(np.float32)
    assert_close(out.cpu().numpy(), want, "training forward", floor=LONG_SUM_FLOOR)
    # backward through the same mask (fp64 on the host, head by head)
    out.copy_(dev(want))  # identical relu masks
    ld.write(L.GRAD_IN, dev(gin))
    grad_out = torch.zeros(n, din, device="cuda")
    ld.backward(out, grad_out)
    g_act = np.where(want > 0, gin, 0).astype(np.float64)
    T = np.zeros((n, d))
    lg, rg = np.zeros(d), np.zeros(d)
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        hk = hfeat[:, sl].astype(np.float64)
        p = norm[:, k].astype(np.float64)
        dp = (g_act[rows][:, sl] * hk[col]).sum(1) * m[:, k] * float(scale)
        rowdot = np.zeros(n)
        np.add.at(rowdot, rows, p * dp)
        ds = p * (dp - rowdot[rows])
        ge = ds * np.where(temp[:, k] > 0, 1.0, 0.2)
        cs, rs = np.zeros(n), np.zeros(n)
        np.add.at(cs, col, ge)
        np.add.at(rs, rows, ge)
        lg[sl], rg[sl] = rs @ hk, cs @ hk
        np.add.at(T[:, sl], col, p_drop[:, k:k + 1].astype(np.float64) * g_act[rows][:, sl])  # out_c += (p m s)_(i->c) grad_i
    assert_close(grad_out.cpu().numpy(), T @ W.T.astype(np.float64), "grad_out", floor=LONG_SUM_FLOOR)
    assert_close(ld.tensor(L.W_NEIGH_GRAD, (din, d)).cpu().numpy(), x.T.astype(np.float64) @ T, "W_grad", floor=LONG_SUM_FLOOR)
    assert_close(ld.tensor(L.ALPHA_LGRAD, (d,)).cpu().numpy(), lg, "alpha_l grad", floor=LONG_SUM_FLOOR)
    assert_close(ld.tensor(L.ALPHA_RGRAD, (d,)).cpu().numpy(), rg, "alpha_r grad", floor=LONG_SUM_FLOOR)
    # the next training forward draws another mask
    ld.forward(out)
    assert (masks_now() != m).mean() > 0.2
