"""ctypes binding of oracle/libgnn_oracle.so (the CPU restatement of the reference's OpenMP path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by graphaibench_amd/.  numpy in, numpy out.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libgnn_oracle.so"
REF_LIB = HERE / "_ref" / "libref_lgraph.so"

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

_lib = None


def build(force: bool = False) -> None:
    if force or not LIB.exists() or LIB.stat().st_mtime < (HERE / "gnn_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "all"], check=True, capture_output=True)


def _declare(l: C.CDLL) -> C.CDLL:
    l.orc_masked_avg_loss.restype = C.c_float
    l.orc_masked_accuracy_single.restype = C.c_float
    l.orc_masked_f1_micro.restype = C.c_float
    return l


def lib() -> C.CDLL:
    """GNN_ORACLE_LIB=<path> loads another build of the same restatement (the sanitizer builds of
    `make -C oracle san`, tests/test_sanitizers.py)"""
    global _lib
    if _lib is None:
        alt = os.environ.get("GNN_ORACLE_LIB")
        if alt:
            _lib = _declare(C.CDLL(alt))
        else:
            build()
            _lib = _declare(C.CDLL(str(LIB)))
    return _lib


def native_lib():
    """the same restatement built with -march=native ON THIS HOST (rebuilt every time: a copy that travelled from
    another machine may use instructions this CPU lacks).  None if the build fails.  Only for bench.py's second
    CPU-baseline figure (the reference Makefile has -march=native commented out)."""
    try:
        subprocess.run(["make", "-B", "-C", str(HERE), "libgnn_oracle_native.so"], check=True, capture_output=True)
        return _declare(C.CDLL(str(HERE / "libgnn_oracle_native.so")))
    except Exception:
        return None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def use_blas(on: bool) -> bool:
    """route matmul through the reference's BLAS call (cblas_sgemm from MKL/OpenBLAS) if one can be
    dlopen'ed; returns whether it is active.  Only the CPU-baseline timing turns this on."""
    return bool(lib().orc_use_blas(int(on)))


def num_threads() -> int:
    return int(lib().orc_num_threads())


class Graph:
    """CSR with int64 rowptr / uint32 colidx on the host."""

    def __init__(self, rowptr, colidx):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.colidx = np.ascontiguousarray(colidx, dtype=np.uint32)
        self.nv = len(self.rowptr) - 1
        self.ne = int(self.rowptr[-1])
        assert len(self.colidx) == self.ne
        self.vd = None

    def add_selfloop(self) -> "Graph":
        rp = np.empty(self.nv + 1, np.int64)
        ci = np.empty(self.ne + self.nv, np.uint32)
        lib().orc_add_selfloop(C.c_int64(self.nv), _p(self.rowptr), _p(self.colidx), _p(rp), _p(ci))
        return Graph(rp, ci)

    def vertex_data(self) -> np.ndarray:
        vd = np.empty(self.nv, np.float32)
        lib().orc_vertex_data(C.c_int64(self.nv), _p(self.rowptr), _p(vd))
        self.vd = vd
        return vd

    def edge_data(self) -> np.ndarray:
        ed = np.empty(self.ne, np.float32)
        lib().orc_edge_data(C.c_int64(self.nv), _p(self.rowptr), _p(self.colidx), _p(ed))
        return ed

    def _struct(self):
        if self.vd is None:
            self.vertex_data()

        class S(C.Structure):
            _fields_ = [("nv", C.c_int64), ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vd", C.c_void_p)]

        return S(self.nv, self.rowptr.ctypes.data, self.colidx.ctypes.data, self.vd.ctypes.data)


# ---- aggregators ---------------------------------------------------------------------------
def gcn_aggregate(g: Graph, x) -> np.ndarray:
    x = _f(x)
    if g.vd is None:
        g.vertex_data()
    out = np.empty_like(x)
    lib().orc_gcn_aggregate(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), _p(g.vd), C.c_int(x.shape[1]), _p(x), _p(out))
    return out


def sage_aggregate(g: Graph, x) -> np.ndarray:
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_sage_aggregate(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(x.shape[1]), _p(x), _p(out))
    return out


def sage_d_aggregate(g: Graph, x) -> np.ndarray:
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_sage_d_aggregate(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(x.shape[1]), _p(x), _p(out))
    return out


def spmm_edge(g: Graph, ew, x) -> np.ndarray:
    x, ew = _f(x), _f(ew)
    out = np.empty((g.nv, x.shape[1]), np.float32)  # (a rectangular matrix has g.nv rows and gathers from x's)
    lib().orc_spmm_edge(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), _p(ew), C.c_int(x.shape[1]), _p(x), _p(out))
    return out


def gat_scores(g: Graph, h, alpha_l, alpha_r, eps=0.2):
    h, al, ar = _f(h), _f(alpha_l), _f(alpha_r)
    temp = np.zeros(g.ne, np.float32)
    scores = np.zeros(g.ne, np.float32)
    norm = np.zeros(g.ne, np.float32)
    lib().orc_gat_scores(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(h.shape[1]), _p(al), _p(ar), _p(h),
                         C.c_float(eps), _p(temp), _p(scores), _p(norm))
    return temp, scores, norm


def gat_aggregate(g: Graph, h, alpha_l, alpha_r):
    h, al, ar = _f(h), _f(alpha_l), _f(alpha_r)
    out = np.empty_like(h)
    temp = np.zeros(g.ne, np.float32)
    scores = np.zeros(g.ne, np.float32)
    norm = np.zeros(g.ne, np.float32)
    lib().orc_gat_aggregate(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(h.shape[1]), _p(al), _p(ar), _p(h),
                            _p(out), _p(temp), _p(scores), _p(norm))
    return out, temp, scores, norm


def sddmm(g: Graph, grad, feat) -> np.ndarray:
    grad, feat = _f(grad), _f(feat)
    out = np.zeros(g.ne, np.float32)
    lib().orc_sddmm(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(grad.shape[1]), _p(grad), _p(feat), _p(out))
    return out


def gat_softmax_bwd_alpha(g: Graph, feat, norm, norm_grad, temp, eps=0.2, fast=False):
    feat = _f(feat)
    norm, norm_grad, temp = _f(norm), _f(norm_grad), _f(temp)
    scores = np.zeros(g.ne, np.float32)
    lg = np.zeros(feat.shape[1], np.float32)
    rg = np.zeros(feat.shape[1], np.float32)
    lib().orc_gat_softmax_bwd_alpha(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(feat.shape[1]), _p(feat),
                                    _p(norm), _p(norm_grad), _p(temp), C.c_float(eps), C.c_int(int(fast)),
                                    _p(scores), _p(lg), _p(rg))
    return scores, lg, rg


def symmetric_csr_transpose(g: Graph, a) -> np.ndarray:
    a = _f(a)
    b = np.zeros(g.ne, np.float32)
    rc = lib().orc_symmetric_csr_transpose(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), _p(a), _p(b))
    if rc != 0:
        raise ValueError("graph is not structurally symmetric")
    return b


def gat_d_aggregate(g: Graph, feat, grad_in, norm, temp, fast=False):
    feat, grad_in = _f(feat), _f(grad_in)
    norm, temp = _f(norm), _f(temp)
    grad_out = np.empty_like(feat)
    scores = np.zeros(g.ne, np.float32)
    ngrad = np.zeros(g.ne, np.float32)
    lg = np.zeros(feat.shape[1], np.float32)
    rg = np.zeros(feat.shape[1], np.float32)
    rc = lib().orc_gat_d_aggregate(C.c_int64(g.nv), _p(g.rowptr), _p(g.colidx), C.c_int(feat.shape[1]), _p(feat),
                                   _p(grad_in), _p(grad_out), _p(norm), _p(temp), C.c_int(int(fast)), _p(scores),
                                   _p(ngrad), _p(lg), _p(rg))
    if rc != 0:
        raise ValueError("graph is not structurally symmetric")
    return grad_out, scores, ngrad, lg, rg


# ---- multi-head GAT = H independent single-head attentions on column slices (BASELINE config 4) ----
def gat_aggregate_mh(g: Graph, h, alpha_l, alpha_r, heads: int):
    """returns out [n,D], temp/scores/norm [ne,heads]"""
    h = _f(h)
    D = h.shape[1]
    dh = D // heads
    out = np.empty_like(h)
    temp = np.empty((g.ne, heads), np.float32)
    scores = np.empty((g.ne, heads), np.float32)
    norm = np.empty((g.ne, heads), np.float32)
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        o, t, s, p = gat_aggregate(g, np.ascontiguousarray(h[:, sl]), np.ascontiguousarray(alpha_l[sl]),
                                   np.ascontiguousarray(alpha_r[sl]))
        out[:, sl], temp[:, k], scores[:, k], norm[:, k] = o, t, s, p
    return out, temp, scores, norm


def gat_d_aggregate_mh(g: Graph, feat, grad_in, norm, temp, heads: int, fast=True):
    feat, grad_in = _f(feat), _f(grad_in)
    D = feat.shape[1]
    dh = D // heads
    grad_out = np.empty_like(feat)
    ds = np.empty((g.ne, heads), np.float32)
    ngrad = np.empty((g.ne, heads), np.float32)
    lg = np.empty(D, np.float32)
    rg = np.empty(D, np.float32)
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        go, d_, ng, l_, r_ = gat_d_aggregate(g, np.ascontiguousarray(feat[:, sl]), np.ascontiguousarray(grad_in[:, sl]),
                                             np.ascontiguousarray(norm[:, k]), np.ascontiguousarray(temp[:, k]),
                                             fast=fast)
        grad_out[:, sl], ds[:, k], ngrad[:, k], lg[sl], rg[sl] = go, d_, ng, l_, r_
    return grad_out, ds, ngrad, lg, rg


# ---- dense / elementwise / optimizer ---------------------------------------------------------
def matmul(A, B, transA=False, transB=False, accum_into=None) -> np.ndarray:
    """reference signature matmul(x, y, z, A, B, C, transA, transB, accum)"""
    A, B = _f(A), _f(B)
    x = A.shape[1] if transA else A.shape[0]
    z = A.shape[0] if transA else A.shape[1]
    y = B.shape[0] if transB else B.shape[1]
    Cm = _f(accum_into).copy() if accum_into is not None else np.empty((x, y), np.float32)
    lib().orc_matmul(C.c_int64(x), C.c_int64(y), C.c_int64(z), _p(A), _p(B), _p(Cm), C.c_int(int(transA)),
                     C.c_int(int(transB)), C.c_int(int(accum_into is not None)))
    return Cm


def relu(x) -> np.ndarray:
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_relu(C.c_int64(x.size), _p(x), _p(out))
    return out


def d_relu(grad, data) -> np.ndarray:
    grad, data = _f(grad), _f(data)
    out = np.empty_like(grad)
    lib().orc_d_relu(C.c_int64(grad.size), _p(grad), _p(data), _p(out))
    return out


def init_glorot(dim_x: int, dim_y: int, seed: int) -> np.ndarray:
    w = np.empty((dim_x, dim_y), np.float32)
    lib().orc_init_glorot(C.c_int64(dim_x), C.c_int64(dim_y), _p(w), C.c_uint(seed))
    return w


class Adam:
    """adam(lr) with its per-weight state map (include/utils/optimizer.h:39-59,99-116)."""

    def __init__(self, lr: float):
        self.alpha = np.float32(lr)
        self.b1_t = C.c_float(0.9)
        self.b2_t = C.c_float(0.999)
        self.state = {}

    def update(self, key, dW, W):
        dW = _f(dW)
        assert W.dtype == np.float32 and W.flags.c_contiguous
        if key not in self.state:
            self.state[key] = (np.zeros(W.size, np.float32), np.zeros(W.size, np.float32))
        m, v = self.state[key]
        lib().orc_adam_update(C.c_int64(W.size), _p(dW), _p(W), _p(m), _p(v), C.c_float(self.alpha),
                              C.byref(self.b1_t), C.byref(self.b2_t))


def softmax_xent_fwd(logits, labels, begin, end, masks=None):
    logits = _f(logits)
    labels = np.ascontiguousarray(labels, np.uint8)
    probs = np.zeros_like(logits)
    losses = np.zeros(logits.shape[0], np.float32)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    lib().orc_softmax_xent_fwd(C.c_int(logits.shape[1]), C.c_int64(begin), C.c_int64(end), _p(m), _p(labels),
                               _p(logits), _p(probs), _p(losses))
    return probs, losses


def softmax_xent_bwd(probs, labels, begin, end, masks=None):
    probs = _f(probs)
    labels = np.ascontiguousarray(labels, np.uint8)
    grad = np.zeros_like(probs)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    lib().orc_softmax_xent_bwd(C.c_int(probs.shape[1]), C.c_int64(begin), C.c_int64(end), _p(m), _p(labels),
                               _p(probs), _p(grad))
    return grad


def sigmoid_xent_fwd(logits, labels, begin, end, masks=None):
    """labels: [n x C] 0/1 bytes -> (sigmoid(logits), per-vertex loss)"""
    logits = _f(logits)
    labels = np.ascontiguousarray(labels, np.uint8)
    assert labels.shape == logits.shape
    probs = np.zeros_like(logits)
    losses = np.zeros(logits.shape[0], np.float32)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    lib().orc_sigmoid_xent_fwd(C.c_int(logits.shape[1]), C.c_int64(begin), C.c_int64(end), _p(m), _p(labels),
                               _p(logits), _p(probs), _p(losses))
    return probs, losses


def sigmoid_xent_bwd(probs, labels, begin, end, masks=None):
    probs = _f(probs)
    labels = np.ascontiguousarray(labels, np.uint8)
    grad = np.zeros_like(probs)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    lib().orc_sigmoid_xent_bwd(C.c_int(probs.shape[1]), C.c_int64(begin), C.c_int64(end), _p(m), _p(labels),
                               _p(probs), _p(grad))
    return grad


def masked_f1_micro(preds, labels, begin, end, masks=None, return_counts=False):
    preds = _f(preds)
    labels = np.ascontiguousarray(labels, np.uint8)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    counts = np.zeros(3, np.int64)
    f1 = float(lib().orc_masked_f1_micro(C.c_int64(begin), C.c_int64(end), C.c_int(preds.shape[1]), _p(m),
                                         _p(preds), _p(labels), _p(counts)))
    return (f1, counts) if return_counts else f1


def masked_avg_loss(losses, begin, end, masks=None) -> float:
    losses = _f(losses)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    return float(lib().orc_masked_avg_loss(C.c_int64(begin), C.c_int64(end), _p(m), _p(losses)))


def masked_accuracy_single(preds, labels, begin, end, masks=None) -> float:
    preds = _f(preds)
    labels = np.ascontiguousarray(labels, np.uint8)
    m = np.ascontiguousarray(masks, np.uint8) if masks is not None else None
    return float(lib().orc_masked_accuracy_single(C.c_int64(begin), C.c_int64(end), C.c_int(preds.shape[1]), _p(m),
                                                  _p(preds), _p(labels)))


def l2norm(x) -> np.ndarray:
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_l2norm(C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(x), _p(out))
    return out


def d_l2norm(feat, grad) -> np.ndarray:
    feat, grad = _f(feat), _f(grad)
    out = np.empty_like(feat)
    lib().orc_d_l2norm(C.c_int64(feat.shape[0]), C.c_int(feat.shape[1]), _p(feat), _p(grad), _p(out))
    return out


# ---- layers -------------------------------------------------------------------------------------
class GCNLayer:
    """GCN_layer (include/layers/graph_conv_layer.h:63-77; src/gnn/gconv/gcn_layer.cpp)."""

    def __init__(self, level, g: Graph, din, dout, act, W=None):
        self.level, self.g, self.din, self.dout, self.act = level, g, din, dout, act
        self.W = init_glorot(din, dout, 1) if W is None else _f(W).copy()  # seed 1 for every layer (Q7)
        n = g.nv
        self.in_temp = np.zeros((n, din), np.float32)
        self.in_temp1 = np.zeros((n, din), np.float32)
        self.out_temp = np.zeros((n, dout), np.float32)
        self.W_grad = np.zeros((din, dout), np.float32)
        self.feat_in = None

    def forward(self, feat_in):
        self.feat_in = _f(feat_in)
        out = np.empty((self.g.nv, self.dout), np.float32)
        s = self.g._struct()
        lib().orc_gcn_layer_forward(C.byref(s), C.c_int(self.din), C.c_int(self.dout), C.c_int(int(self.act)),
                                    _p(self.feat_in), _p(self.W), _p(self.in_temp1), _p(self.out_temp), _p(out))
        self.feat_out = out
        return out

    def backward(self, grad_in):
        """grad_in is modified in place (d_relu, Q9); returns grad_out (None for level 0)."""
        assert grad_in.dtype == np.float32 and grad_in.flags.c_contiguous
        grad_out = np.zeros((self.g.nv, self.din), np.float32) if self.level > 0 else None
        s = self.g._struct()
        lib().orc_gcn_layer_backward(C.byref(s), C.c_int(self.level), C.c_int(self.din), C.c_int(self.dout),
                                     C.c_int(int(self.act)), _p(self.feat_in), _p(self.W), _p(self.feat_out),
                                     _p(grad_in), _p(self.in_temp), _p(self.in_temp1), _p(self.out_temp),
                                     _p(grad_out), _p(self.W_grad))
        return grad_out


class SAGELayer:
    """SAGE_layer (src/gnn/gconv/sage_layer.cpp)."""

    def __init__(self, level, g: Graph, din, dout, act, W_neigh=None, W_self=None):
        self.level, self.g, self.din, self.dout, self.act = level, g, din, dout, act
        self.W_neigh = init_glorot(din, dout, 1) if W_neigh is None else _f(W_neigh).copy()
        self.W_self = init_glorot(din, dout, 2) if W_self is None else _f(W_self).copy()
        n = g.nv
        self.in_temp = np.zeros((n, din), np.float32)
        self.in_temp1 = np.zeros((n, din), np.float32)
        self.out_temp = np.zeros((n, dout), np.float32)
        self.W_neigh_grad = np.zeros((din, dout), np.float32)
        self.W_self_grad = np.zeros((din, dout), np.float32)

    def forward(self, feat_in):
        self.feat_in = _f(feat_in)
        out = np.empty((self.g.nv, self.dout), np.float32)
        s = self.g._struct()
        lib().orc_sage_layer_forward(C.byref(s), C.c_int(self.din), C.c_int(self.dout), C.c_int(int(self.act)),
                                     _p(self.feat_in), _p(self.W_neigh), _p(self.W_self), _p(self.in_temp1),
                                     _p(self.out_temp), _p(out))
        self.feat_out = out
        return out

    def backward(self, grad_in):
        assert grad_in.dtype == np.float32 and grad_in.flags.c_contiguous
        grad_out = np.zeros((self.g.nv, self.din), np.float32) if self.level > 0 else None
        s = self.g._struct()
        lib().orc_sage_layer_backward(C.byref(s), C.c_int(self.level), C.c_int(self.din), C.c_int(self.dout),
                                      C.c_int(int(self.act)), _p(self.feat_in), _p(self.W_neigh), _p(self.W_self),
                                      _p(self.feat_out), _p(grad_in), _p(self.in_temp), _p(self.in_temp1),
                                      _p(self.out_temp), _p(grad_out), _p(self.W_neigh_grad), _p(self.W_self_grad))
        return grad_out


class GATLayer:
    """GAT_layer (src/gnn/gconv/gat_layer.cpp) with its GAT_Aggregator state."""

    def __init__(self, level, g: Graph, din, dout, act, W=None, alpha_l=None, alpha_r=None, fast=False):
        self.level, self.g, self.din, self.dout, self.act, self.fast = level, g, din, dout, act, fast
        self.W = init_glorot(din, dout, 1) if W is None else _f(W).copy()
        # gat_aggregator.cpp:11-12: init_glorot(l, 1, alpha_l, 2) / (l, 1, alpha_r, 3)
        self.alpha_l = init_glorot(dout, 1, 2).ravel() if alpha_l is None else _f(alpha_l).copy()
        self.alpha_r = init_glorot(dout, 1, 3).ravel() if alpha_r is None else _f(alpha_r).copy()
        n, ne = g.nv, g.ne
        self.out_temp = np.zeros((n, dout), np.float32)
        self.temp_scores = np.zeros(ne, np.float32)
        self.scores = np.zeros(ne, np.float32)
        self.norm_scores = np.zeros(ne, np.float32)
        self.norm_scores_grad = np.zeros(ne, np.float32)
        self.W_grad = np.zeros((din, dout), np.float32)
        self.alpha_lgrad = np.zeros(dout, np.float32)
        self.alpha_rgrad = np.zeros(dout, np.float32)

    def forward(self, feat_in):
        self.feat_in = _f(feat_in)
        out = np.empty((self.g.nv, self.dout), np.float32)
        s = self.g._struct()
        lib().orc_gat_layer_forward(C.byref(s), C.c_int(self.din), C.c_int(self.dout), C.c_int(int(self.act)),
                                    _p(self.feat_in), _p(self.W), _p(self.alpha_l), _p(self.alpha_r),
                                    _p(self.out_temp), _p(out), _p(self.temp_scores), _p(self.scores),
                                    _p(self.norm_scores))
        self.feat_out = out
        return out

    def backward(self, grad_in):
        assert grad_in.dtype == np.float32 and grad_in.flags.c_contiguous
        grad_out = np.zeros((self.g.nv, self.din), np.float32) if self.level > 0 else None
        s = self.g._struct()
        rc = lib().orc_gat_layer_backward(C.byref(s), C.c_int(self.level), C.c_int(self.din), C.c_int(self.dout),
                                          C.c_int(int(self.act)), _p(self.feat_in), _p(self.W), _p(self.feat_out),
                                          _p(grad_in), _p(self.out_temp), _p(grad_out), _p(self.W_grad),
                                          _p(self.norm_scores), _p(self.temp_scores), C.c_int(int(self.fast)),
                                          _p(self.scores), _p(self.norm_scores_grad), _p(self.alpha_lgrad),
                                          _p(self.alpha_rgrad))
        if rc != 0:
            raise ValueError("graph is not structurally symmetric")
        return grad_out


# ---- the real reference (oracle/_ref), only where it was built -----------------------------------
def ref_lib():
    """oracle/_ref/libref_lgraph.so: the reference's own lgraph.cpp/reader.cpp, compiled unmodified."""
    if not REF_LIB.exists():
        return None
    os.environ.setdefault("DATASET_PATH", "/tmp/")  # configs.h:5 reads it at load time
    return C.CDLL(str(REF_LIB))


def ref_sample_subgraph(rowptr, colidx, train_masks, n: int, seed: int):
    """the REAL reference's Sampler::select_vertices + generateSubgraph (oracle/_ref, sampler.cpp compiled
    unmodified): -> (kept vertex ids ascending, subgraph rowptr uint32, subgraph colidx uint32) or None without _ref"""
    lib_ = ref_lib()
    if lib_ is None or not hasattr(lib_, "ref_sample_subgraph"):
        return None
    rp = np.ascontiguousarray(rowptr, np.uint32)
    ci = np.ascontiguousarray(colidx, np.uint32)
    m = np.ascontiguousarray(train_masks, np.uint8).copy()
    nv, ne = len(rp) - 1, len(ci)
    kept = np.zeros(max(n, 1), np.uint32)
    sub_ne = C.c_uint32(0)
    lib_.ref_sample_subgraph.restype = C.c_uint32
    k = lib_.ref_sample_subgraph(C.c_uint32(nv), C.c_uint32(ne), _p(rp), _p(ci), _p(m), C.c_uint32(n), C.c_uint(seed),
                                 _p(kept), C.byref(sub_ne), None, None)
    srp = np.zeros(k + 1, np.uint32)
    sci = np.zeros(max(sub_ne.value, 1), np.uint32)
    lib_.ref_sample_subgraph(C.c_uint32(nv), C.c_uint32(ne), _p(rp), _p(ci), _p(m), C.c_uint32(n), C.c_uint(seed),
                             _p(kept), C.byref(sub_ne), _p(srp), _p(sci))
    return kept[:k].copy(), srp, sci[:sub_ne.value].copy()


def ref_masked_graph(rowptr, colidx, masks):
    """the REAL reference's LearningGraph::generate_masked_graph -> (rowptr, colidx) or None without _ref"""
    lib_ = ref_lib()
    if lib_ is None or not hasattr(lib_, "ref_masked_graph"):
        return None
    rp = np.ascontiguousarray(rowptr, np.uint32)
    ci = np.ascontiguousarray(colidx, np.uint32)
    m = np.ascontiguousarray(masks, np.uint8).copy()
    nv, ne = len(rp) - 1, len(ci)
    ne_out = C.c_uint32(0)
    lib_.ref_masked_graph(C.c_uint32(nv), C.c_uint32(ne), _p(rp), _p(ci), _p(m), C.byref(ne_out), None, None)
    rpo = np.zeros(nv + 1, np.uint32)
    cio = np.zeros(max(ne_out.value, 1), np.uint32)
    lib_.ref_masked_graph(C.c_uint32(nv), C.c_uint32(ne), _p(rp), _p(ci), _p(m), C.byref(ne_out), _p(rpo), _p(cio))
    return rpo, cio[:ne_out.value].copy()


REF_PART_LIB = HERE / "_ref" / "libref_partition.so"


def ref_partition(rowptr, colidx, parts: int):
    """the REAL reference's PartitionedGraph::edgecut_induced_partition1D (oracle/_ref/libref_partition.so:
    graph_partition.cc + graph.cc + VertexSet.cc compiled unmodified).  -> list of dicts per subgraph
    (begin, end, idx_map, rowptr, colidx) or None where _ref is not built."""
    if not REF_PART_LIB.exists():
        return None
    lib_ = C.CDLL(str(REF_PART_LIB))
    rp = np.ascontiguousarray(rowptr, np.int64)
    ci = np.ascontiguousarray(colidx, np.uint32)
    nv, ne = len(rp) - 1, len(ci)
    n = lib_.refp_partition(C.c_uint32(nv), C.c_int64(ne), _p(rp), _p(ci), C.c_int(parts))
    out = []
    for i in range(n):
        snv, sne, b, e = C.c_uint32(), C.c_int64(), C.c_uint32(), C.c_uint32()
        lib_.refp_sizes(C.c_int(i), C.byref(snv), C.byref(sne), C.byref(b), C.byref(e))
        idx = np.zeros(snv.value, np.uint32)
        srp = np.zeros(snv.value + 1, np.int64)
        sci = np.zeros(max(sne.value, 1), np.uint32)
        lib_.refp_copy(C.c_int(i), _p(idx), _p(srp), _p(sci))
        out.append(dict(begin=b.value, end=e.value, idx_map=idx, rowptr=srp, colidx=sci[:sne.value].copy()))
    return out
