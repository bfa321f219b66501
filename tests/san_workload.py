"""Workload for the sanitizer builds of the oracle (tests/test_sanitizers.py runs this file in a child process with
GNN_ORACLE_LIB pointing at oracle/_san/libgnn_oracle_{asan,tsan}.so and the sanitizer runtime preloaded): every
OpenMP loop of oracle/gnn_oracle.c once, several threads, graphs with a hub row, empty rows and a self-loop rebuild.
The known race sites of the reference are inside: the per-thread alpha-gradient partials of GAT backward
(gat_aggregator.cpp:124-165, 56 slots indexed by thread id -- the restatement sizes them by the thread count) and the
per-thread k-slab partials of the TN matmul."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import binding as orc  # noqa: E402
from util import path_graph, random_graph  # noqa: E402


def feat(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def main():
    orc.set_threads(4)
    for rp, ci in (random_graph(400, 8, seed=1, power_law=True, hub_deg=150), path_graph(7),
                   random_graph(64, 0.5, seed=2)):  # the last one has isolated vertices
        g0 = orc.Graph(rp, ci)
        g1 = g0.add_selfloop()
        n = g0.nv
        g1.vertex_data(), g1.edge_data()
        for d in (1, 16, 33):
            x = feat(n, d, d)
            orc.gcn_aggregate(g1, x)
            orc.sage_aggregate(g0, x)
            orc.sage_d_aggregate(g0, x)
            orc.spmm_edge(g1, g1.edge_data(), x)
        x, gin = feat(n, 24, 3), feat(n, 16, 4)
        for level in (0, 1):
            l = orc.GCNLayer(level, g1, 24, 16, True)
            l.forward(x), l.backward(gin.copy())
            l = orc.GCNLayer(level, g1, 16, 24, True)  # the aggregate-first branch
            l.forward(gin), l.backward(x.copy())
            l = orc.SAGELayer(level, g0, 24, 16, True)
            l.forward(x), l.backward(gin.copy())
            for fast in (False, True):
                l = orc.GATLayer(level, g1, 24, 16, True, fast=fast)
                l.forward(x), l.backward(gin.copy())
        h = feat(n, 16, 5)
        al, ar = feat(1, 16, 6).ravel(), feat(1, 16, 7).ravel()
        out, temp, scores, norm = orc.gat_aggregate_mh(g1, h, al, ar, 4)
        orc.gat_d_aggregate_mh(g1, h, gin, norm, temp, 4)
        orc.symmetric_csr_transpose(g1, norm[:, 0].copy())
        orc.sddmm(g1, gin, h)
        # dense / elementwise / optimizer / loss
        A, B = feat(n, 24, 8), feat(24, 16, 9)
        C = orc.matmul(A, B)
        orc.matmul(C, B, False, True)
        orc.matmul(A, C, True, False)
        orc.matmul(A, B, accum_into=C.copy())
        orc.relu(C), orc.d_relu(C, C)
        W = orc.init_glorot(24, 16, 1)
        opt = orc.Adam(0.01)
        opt.update("w", feat(24, 16, 10), W), opt.update("w", feat(24, 16, 11), W)
        labels = np.random.default_rng(12).integers(0, 16, n).astype(np.uint8)
        masks = (np.random.default_rng(13).random(n) < 0.5).astype(np.uint8)
        for mk in (None, masks):
            probs, loss = orc.softmax_xent_fwd(C, labels, 0, n, mk)
            orc.softmax_xent_bwd(probs, labels, 0, n, mk)
            orc.masked_avg_loss(loss, 0, n, mk), orc.masked_accuracy_single(probs, labels, 0, n, mk)
            ml = (np.random.default_rng(14).random((n, 16)) < 0.2).astype(np.uint8)
            probs, loss = orc.sigmoid_xent_fwd(C, ml, 0, n, mk)
            orc.sigmoid_xent_bwd(probs, ml, 0, n, mk)
            orc.masked_f1_micro(probs, ml, 0, n, mk)
        orc.l2norm(C), orc.d_l2norm(C, feat(n, 16, 15))
    print("san_workload done")


if __name__ == "__main__":
    main()
