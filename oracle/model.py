"""TEST INFRASTRUCTURE (see binding.py): the reference's Model (src/gnn/net.cpp:361-615) composed from the CPU restatement's
layers -- same Glorot seeds, same optimizer sharing quirks -- for loss-curve comparisons against the
trainer (tests/test_gpu_driver.py) and as the CPU side of bench.py's epoch workloads.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline / parity legs may import it."""
import numpy as np

from . import binding as orc


class MultiHeadGATLayer:
    """GAT_layer with the head count as an extension (heads independent single-head attentions over column slices of X.W:
    every head IS the reference's single-head aggregator, gat_aggregator.cpp:57-200; the trainer's GAIB_GAT_HEADS).
    heads = 1 is orc.GATLayer's arithmetic."""

    def __init__(self, level, g, din, dout, act, heads):
        self.level, self.g, self.din, self.dout, self.act, self.heads = level, g, din, dout, act, heads
        self.W = orc.init_glorot(din, dout, 1)
        self.alpha_l = orc.init_glorot(dout, 1, 2).ravel()
        self.alpha_r = orc.init_glorot(dout, 1, 3).ravel()

    def forward(self, feat_in):
        self.feat_in = np.ascontiguousarray(feat_in, np.float32)
        self.h = orc.matmul(self.feat_in, self.W)
        out, self.temp, _, self.norm = orc.gat_aggregate_mh(self.g, self.h, self.alpha_l, self.alpha_r, self.heads)
        self.feat_out = orc.relu(out) if self.act else out
        return self.feat_out

    def backward(self, grad_in):
        g = orc.d_relu(grad_in, self.feat_out) if self.act else grad_in  # Q9: masked with the post-activation output
        T, _, _, self.alpha_lgrad, self.alpha_rgrad = orc.gat_d_aggregate_mh(self.g, self.h, np.ascontiguousarray(g), self.norm,
                                                                             self.temp, self.heads, fast=True)
        self.W_grad = orc.matmul(self.feat_in, T, True, False)
        return orc.matmul(T, self.W, False, True) if self.level > 0 else None


class OracleModel:
    def __init__(self, arch, rp, ci, F, H, C, L, lr, heads=1):
        self.arch, self.lr, self.L = arch, lr, L
        g = orc.Graph(rp, ci)
        self.g = g if arch == "sage" else g.add_selfloop()
        last = H if arch == "gat" else C
        dims = [(F if l == 0 else H, H if l < L - 1 else last) for l in range(L)]
        gat = (lambda *a: MultiHeadGATLayer(*a, heads)) if heads > 1 else (lambda *a: orc.GATLayer(*a, fast=True))
        mk = {"gcn": orc.GCNLayer, "sage": orc.SAGELayer, "gat": gat}[arch]
        self.layers = [mk(l, self.g, di, do, l < L - 1) for l, (di, do) in enumerate(dims)]
        self.opt = orc.Adam(lr)                      # shared by GCN / GAT gconv weights (Q6)
        self.optm = [orc.Adam(lr) for _ in range(L)]  # SAGE: per layer
        self.alpha_opt = [orc.Adam(lr) for _ in range(L)]
        if arch == "gat":
            self.Wd = orc.init_glorot(H, C, 1)
            self.dense_opt = orc.Adam(lr)

    def epoch(self, x, labels, begin, end, masks, sigmoid=False):
        acts = [x]
        for l in self.layers:
            acts.append(l.forward(acts[-1]))
        if self.arch == "gat":
            z = orc.l2norm(acts[-1])
            logits = orc.matmul(z, self.Wd)
        else:
            logits = acts[-1]
        if sigmoid:  # labels: [n x C] multi-hot rows
            probs, lv = orc.sigmoid_xent_fwd(logits, labels, begin, end, masks)
            acc = orc.masked_f1_micro(probs, labels, begin, end, masks)
            g = orc.sigmoid_xent_bwd(probs, labels, begin, end, masks)
        else:
            probs, lv = orc.softmax_xent_fwd(logits, labels, begin, end, masks)
            acc = orc.masked_accuracy_single(logits, labels, begin, end, masks)
            g = orc.softmax_xent_bwd(probs, labels, begin, end, masks)
        loss = orc.masked_avg_loss(lv, begin, end, masks)
        if self.arch == "gat":
            dWd = orc.matmul(z, g, True, False)
            gz = orc.matmul(g, self.Wd, False, True)
            self.dense_opt.update("wd", dWd, self.Wd)  # dense_layer::backward updates its own weights
            g = orc.d_l2norm(acts[-1], gz)
        for l in reversed(self.layers):
            g = l.backward(np.ascontiguousarray(g))
        for i, l in enumerate(self.layers):
            if self.arch == "gcn":
                self.opt.update(("w", i), l.W_grad, l.W)
            elif self.arch == "sage":
                self.optm[i].update(("wn", i), l.W_neigh_grad, l.W_neigh)
                self.optm[i].update(("ws", i), l.W_self_grad, l.W_self)
            else:
                self.opt.update(("w", i), l.W_grad, l.W)
                self.alpha_opt[i].update(("al", i), l.alpha_lgrad, l.alpha_l)
                self.alpha_opt[i].update(("ar", i), l.alpha_rgrad, l.alpha_r)
        return loss, acc
