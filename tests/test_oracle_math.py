"""CPU suite: the oracle (oracle/gnn_oracle.c) against INDEPENDENT fp64 formulations of the same
math (dense linear algebra with numpy), at sizes where fp64 dense algebra is exact enough.
This is the cross-check for the rows whose reference TUs cannot be built here (a2-a12)."""
import numpy as np
import pytest

from oracle import binding as orc
from util import dense_adj, path_graph, random_graph, rel_err

TOL = 2e-6  # fp32 restatement vs fp64 dense algebra


def _feat(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


@pytest.mark.parametrize("n,deg,d", [(7, 2, 3), (64, 6, 16), (300, 9, 33), (500, 20, 128)])
def test_gcn_aggregate_is_sym_normalised_adjacency(n, deg, d):
    rp, ci = random_graph(n, deg, seed=n)
    g = orc.Graph(rp, ci).add_selfloop()
    x = _feat(n, d, 1)
    out = orc.gcn_aggregate(g, x)
    A = dense_adj(g.rowptr, g.colidx)
    dinv = 1.0 / np.sqrt(A.sum(1))
    ref = (dinv[:, None] * A * dinv[None, :]) @ x.astype(np.float64)
    assert rel_err(out, ref) < TOL


def test_sage_mean_and_transpose():
    rp, ci = random_graph(200, 8, seed=3, power_law=True)
    g = orc.Graph(rp, ci)
    x = _feat(200, 47, 2)
    A = dense_adj(rp, ci)
    deg = np.maximum(A.sum(1), 1)
    fwd = (A / deg[:, None]) @ x.astype(np.float64)
    bwd = (A / deg[None, :]) @ x.astype(np.float64)  # (D^-1 A)^T = A D^-1 for symmetric A
    assert rel_err(orc.sage_aggregate(g, x), fwd) < TOL
    assert rel_err(orc.sage_d_aggregate(g, x), bwd) < TOL


def test_isolated_vertices_give_zero_rows():
    rp = np.array([0, 0, 1, 2, 2], np.int64)
    ci = np.array([2, 1], np.uint32)
    g = orc.Graph(rp, ci)
    x = _feat(4, 5, 0)
    out = orc.sage_aggregate(g, x)
    assert np.all(out[0] == 0) and np.all(out[3] == 0)
    assert np.array_equal(out[1], x[2]) and np.array_equal(out[2], x[1])
    assert g.vertex_data()[0] == 0.0  # lgraph.cpp:30


def test_add_selfloop_sorted_insert():
    rp, ci = path_graph(7)
    g = orc.Graph(rp, ci).add_selfloop()
    assert g.ne == len(ci) + 7
    for v in range(7):
        row = g.colidx[g.rowptr[v]:g.rowptr[v + 1]]
        assert list(row) == sorted(set(row)) and v in row


def _dense_gat(g, h, al, ar):
    n = g.nv
    A = dense_adj(g.rowptr, g.colidx) > 0
    h64 = h.astype(np.float64)
    s = (h64 @ al.astype(np.float64))[:, None] + (h64 @ ar.astype(np.float64))[None, :]
    lr = np.where(s > 0, s, 0.2 * s)
    lr = np.where(A, lr, -np.inf)
    p = np.exp(lr - lr.max(1, keepdims=True))
    p = p / p.sum(1, keepdims=True)
    return s, p, p @ h64


def test_gat_forward_is_masked_softmax_attention():
    rp, ci = random_graph(120, 7, seed=5)
    g = orc.Graph(rp, ci).add_selfloop()
    h = _feat(120, 24, 7)
    al = _feat(1, 24, 8).ravel() * 0.3
    ar = _feat(1, 24, 9).ravel() * 0.3
    out, temp, scores, norm = orc.gat_aggregate(g, h, al, ar)
    s, p, ref = _dense_gat(g, h, al, ar)
    assert rel_err(out, ref) < 1e-5
    rows = np.repeat(np.arange(g.nv), np.diff(g.rowptr))
    assert rel_err(norm, p[rows, g.colidx]) < 1e-5
    assert rel_err(temp, s[rows, g.colidx]) < 1e-5


@pytest.mark.parametrize("fast", [False, True])
def test_gat_backward_matches_autograd_of_the_reference_semantics(fast):
    """d_aggregate against torch autograd of: out = P h_detached, P = softmax(lrelu(a_l.h_i + a_r.h_j))
    where scores depend on alpha only (no gradient through scores into h, Q18)."""
    import torch

    rp, ci = random_graph(60, 6, seed=11)
    g = orc.Graph(rp, ci).add_selfloop()
    n, d = g.nv, 10
    h = _feat(n, d, 1)
    al = (_feat(1, d, 2).ravel() * 0.3)
    ar = (_feat(1, d, 3).ravel() * 0.3)
    gin = _feat(n, d, 4)
    out, temp, scores, norm = orc.gat_aggregate(g, h, al, ar)
    grad_out, ds, ngrad, lg, rg = orc.gat_d_aggregate(g, h, gin, norm, temp, fast=fast)

    A = torch.tensor(dense_adj(g.rowptr, g.colidx) > 0)
    ht = torch.tensor(h, dtype=torch.float64)
    alt = torch.tensor(al, dtype=torch.float64, requires_grad=True)
    art = torch.tensor(ar, dtype=torch.float64, requires_grad=True)
    s = (ht @ alt)[:, None] + (ht @ art)[None, :]
    lr = torch.where(s > 0, s, 0.2 * s).masked_fill(~A, float("-inf"))
    P = torch.softmax(lr, 1)
    loss = ((P @ ht) * torch.tensor(gin, dtype=torch.float64)).sum()
    loss.backward()
    assert rel_err(lg, alt.grad.numpy()) < 1e-4
    assert rel_err(rg, art.grad.numpy()) < 1e-4
    # feature gradient: P^T g  (gat_aggregator.cpp:175,198)
    ref_go = P.detach().numpy().T @ gin.astype(np.float64)
    assert rel_err(grad_out, ref_go) < 1e-5


def test_d_softmax_branches_agree():
    """the O(deg^2) fallback (math_functions.cpp:505-513) and the closed form (:497-504)"""
    rp, ci = random_graph(80, 10, seed=2, hub_deg=60)
    g = orc.Graph(rp, ci).add_selfloop()
    h = _feat(80, 8, 1)
    al, ar = _feat(1, 8, 2).ravel(), _feat(1, 8, 3).ravel()
    _, temp, _, norm = orc.gat_aggregate(g, h, al, ar)
    dp = _feat(1, g.ne, 5).ravel()
    a = orc.gat_softmax_bwd_alpha(g, h, norm, dp, temp, fast=False)
    b = orc.gat_softmax_bwd_alpha(g, h, norm, dp, temp, fast=True)
    for x, y in zip(a, b):
        assert rel_err(x, y) < 1e-5


def test_symmetric_transpose_roundtrip_and_asymmetry():
    rp, ci = random_graph(100, 5, seed=4)
    g = orc.Graph(rp, ci)
    a = _feat(1, g.ne, 0).ravel()
    b = orc.symmetric_csr_transpose(g, a)
    assert np.array_equal(orc.symmetric_csr_transpose(g, b), a)
    A, B = dense_adj(rp, ci, a), dense_adj(rp, ci, b)
    assert np.array_equal(A.T, B)
    bad = orc.Graph(np.array([0, 1, 1], np.int64), np.array([1], np.uint32))
    with pytest.raises(ValueError):
        orc.symmetric_csr_transpose(bad, np.ones(1, np.float32))


@pytest.mark.parametrize("tA,tB,accum", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 0)])
def test_matmul(tA, tB, accum):
    rng = np.random.default_rng(0)
    x, y, z = 37, 19, 53
    A = rng.standard_normal((z, x) if tA else (x, z)).astype(np.float32)
    B = rng.standard_normal((y, z) if tB else (z, y)).astype(np.float32)
    C0 = rng.standard_normal((x, y)).astype(np.float32)
    out = orc.matmul(A, B, bool(tA), bool(tB), C0 if accum else None)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64) + (C0 if accum else 0)
    assert rel_err(out, ref) < 1e-5


def test_relu_and_adam_and_loss():
    x = np.array([-1.0, 0.0, 2.5, -0.0], np.float32)
    assert np.array_equal(orc.relu(x), np.array([0, 0, 2.5, 0], np.float32))
    assert np.array_equal(orc.d_relu(np.ones(4, np.float32), x), np.array([0, 0, 1, 0], np.float32))
    # adam: eps inside the sqrt, beta powers advance per call (optimizer.cpp:22-35)
    opt = orc.Adam(0.01)
    W = np.ones(3, np.float32)
    dW = np.array([0.5, -2.0, 0.0], np.float32)
    opt.update("w", dW, W)
    m = 0.1 * dW.astype(np.float64)
    v = 0.001 * dW.astype(np.float64) ** 2
    ref = 1.0 - 0.01 * (m / 0.1) / np.sqrt(v / 0.001 + 1e-8)
    assert rel_err(W, ref) < 1e-6
    assert abs(opt.b1_t.value - 0.81) < 1e-6 and abs(opt.b2_t.value - 0.998001) < 1e-6
    # softmax-xent: grad divides by (end-begin) (Q8)
    logits = np.random.default_rng(1).standard_normal((6, 5)).astype(np.float32)
    labels = np.array([0, 1, 2, 3, 4, 0], np.uint8)
    probs, losses = orc.softmax_xent_fwd(logits, labels, 1, 5)
    e = np.exp(logits.astype(np.float64) - logits.max(1, keepdims=True))
    p = e / e.sum(1, keepdims=True)
    assert rel_err(probs[1:5], p[1:5]) < 1e-6 and np.all(probs[0] == 0)
    assert rel_err(losses[1:5], -np.log(p[np.arange(1, 5), labels[1:5]])) < 1e-6
    g = orc.softmax_xent_bwd(probs, labels, 1, 5)
    onehot = np.eye(5)[labels]
    assert rel_err(g[1:5], (p[1:5] - onehot[1:5]) / 4.0) < 1e-6
    assert abs(orc.masked_avg_loss(losses, 1, 5) - losses[1:5].mean()) < 1e-6


def test_gcn_layer_forward_backward_against_autograd():
    import torch

    rp, ci = random_graph(90, 6, seed=21)
    g = orc.Graph(rp, ci).add_selfloop()
    for din, dout in [(12, 20), (20, 7)]:  # both branches of gcn_layer.cpp:19-25
        x = _feat(90, din, 1)
        layer = orc.GCNLayer(1, g, din, dout, act=True)
        out = layer.forward(x)
        gin = _feat(90, dout, 2)
        grad_out = layer.backward(gin.copy())
        A = torch.tensor(dense_adj(g.rowptr, g.colidx))
        dinv = A.sum(1).rsqrt()
        Ah = dinv[:, None] * A * dinv[None, :]
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        Wt = torch.tensor(layer.W, dtype=torch.float64, requires_grad=True)
        o = torch.relu(Ah @ (xt @ Wt))
        (o * torch.tensor(gin, dtype=torch.float64)).sum().backward()
        assert rel_err(out, o.detach().numpy()) < 1e-5
        assert rel_err(grad_out, xt.grad.numpy()) < 1e-5
        assert rel_err(layer.W_grad, Wt.grad.numpy()) < 1e-5


def test_sage_layer_forward_backward_against_autograd():
    import torch

    rp, ci = random_graph(70, 5, seed=8, power_law=True)
    # every vertex needs an edge for the mean to be defined in the dense formulation
    deg = np.diff(rp)
    assert deg.min() >= 0
    g = orc.Graph(rp, ci)
    for din, dout in [(9, 14), (14, 6)]:
        x = _feat(70, din, 3)
        layer = orc.SAGELayer(1, g, din, dout, act=True)
        out = layer.forward(x)
        gin = _feat(70, dout, 4)
        grad_out = layer.backward(gin.copy())
        A = torch.tensor(dense_adj(rp, ci))
        M = A / A.sum(1).clamp(min=1)[:, None]
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        Wn = torch.tensor(layer.W_neigh, dtype=torch.float64, requires_grad=True)
        Ws = torch.tensor(layer.W_self, dtype=torch.float64, requires_grad=True)
        o = torch.relu(M @ xt @ Wn + xt @ Ws)
        (o * torch.tensor(gin, dtype=torch.float64)).sum().backward()
        assert rel_err(out, o.detach().numpy()) < 1e-5
        assert rel_err(grad_out, xt.grad.numpy()) < 1e-5
        assert rel_err(layer.W_neigh_grad, Wn.grad.numpy()) < 1e-5
        assert rel_err(layer.W_self_grad, Ws.grad.numpy()) < 1e-5


def test_citeseer_plumbing_cpu_path():
    """BASELINE config 0: citeseer GCN 2-layer hidden 16 on the CPU path (the shipped citeseer has
    topology + labels only: features and the train range are synthesised)."""
    from pathlib import Path

    d = Path(__file__).resolve().parent / "golden" / "citeseer"
    rp = np.fromfile(d / "graph.vertex.bin", np.int64)
    ci = np.fromfile(d / "graph.edge.bin", np.uint32)
    labels = np.fromfile(d / "graph.vlabel.bin", np.uint8)
    n, F, H, C = len(rp) - 1, 64, 16, int(labels.max()) + 1
    assert n == 3312 and len(ci) == 9072 and C == 6
    g = orc.Graph(rp, ci).add_selfloop()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, F)).astype(np.float32) * 0.3
    x[np.arange(n), labels.astype(int)] += 1.0
    masks = np.zeros(n, np.uint8)
    masks[:300] = 1
    l0, l1 = orc.GCNLayer(0, g, F, H, True), orc.GCNLayer(1, g, H, C, False)
    opt = orc.Adam(0.02)
    losses = []
    for _ in range(8):
        a = l0.forward(x)
        logits = l1.forward(a)
        p, lv = orc.softmax_xent_fwd(logits, labels, 0, 300, masks)
        losses.append(orc.masked_avg_loss(lv, 0, 300, masks))
        g1 = orc.softmax_xent_bwd(p, labels, 0, 300, masks)
        g0 = l1.backward(g1)
        l0.backward(g0)
        opt.update("w0", l0.W_grad, l0.W)
        opt.update("w1", l1.W_grad, l1.W)
    assert losses[-1] < losses[0] - 0.05 and all(np.isfinite(losses))


def test_sigmoid_loss_and_micro_f1_against_torch_and_sklearn():
    """multi-label head (sigmoid_loss_layer.cpp, math_functions.cpp:517-621): the loss is the stable
    BCE-with-logits, the gradient divides by (end-begin), masked_accuracy_multi is sklearn's micro F1."""
    import torch
    from sklearn.metrics import f1_score
    rng = np.random.default_rng(11)
    n, c, begin, end = 400, 37, 30, 330
    logits = (rng.standard_normal((n, c)) * 4).astype(np.float32)
    logits[5, :3] = [0.0, 60.0, -60.0]
    labels = (rng.random((n, c)) < 0.3).astype(np.uint8)
    masks = np.zeros(n, np.uint8)
    masks[begin:end:2] = 1  # a strided mask inside the range
    probs, losses = orc.sigmoid_xent_fwd(logits, labels, begin, end, masks)
    sel = masks == 1
    want_p = torch.sigmoid(torch.from_numpy(logits).double()).numpy()
    want_l = torch.nn.functional.binary_cross_entropy_with_logits(
        torch.from_numpy(logits).double(), torch.from_numpy(labels).double(), reduction="none").sum(1).numpy()
    assert np.abs(probs[sel] - want_p[sel]).max() < 1e-6
    assert np.abs(losses[sel] - want_l[sel]).max() < 1e-3 * max(1.0, np.abs(want_l).max() * 1e-2)
    assert np.all(probs[~sel] == 0) and np.all(losses[~sel] == 0)  # rows outside the mask are not touched
    g = orc.sigmoid_xent_bwd(probs, labels, begin, end, masks)
    assert np.abs(g[sel] - (probs[sel] - labels[sel]) / np.float32(end - begin)).max() < 1e-7
    assert np.all(g[~sel] == 0)
    f1, (tp, fp, fn) = orc.masked_f1_micro(probs, labels, begin, end, masks, return_counts=True)
    hot = (probs[sel] > 0.5).astype(np.uint8)
    assert (tp, fp, fn) == (int((hot & labels[sel]).sum()), int((hot & (1 - labels[sel])).sum()),
                            int(((1 - hot) & labels[sel]).sum()))
    assert abs(f1 - f1_score(labels[sel], hot, average="micro")) < 1e-6
    # nothing predicted and nothing true -> 0 (the reference's guards, not NaN)
    assert orc.masked_f1_micro(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.uint8), 0, 4) == 0.0
