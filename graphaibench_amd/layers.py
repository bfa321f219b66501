"""ctypes binding of include/gaib_layers.h (libgaib_gnn.so): the host C++ mirror of the
reference's layer/operator API (GCN_layer / SAGE_layer / GAT_layer on a LearningGraph).

Harness plumbing for tests/ and bench.py; device buffers are torch tensors, compute is the C++
layer code calling the HIP kernels through the C ABI.  No fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import capi

LIB_PATH = capi.LIB_DIR / "libgaib_gnn.so"
GCN, SAGE, GAT = 0, 1, 2
(FEAT_IN, GRAD_IN, W_NEIGH, W_NEIGH_GRAD, W_SELF, W_SELF_GRAD, ALPHA_L, ALPHA_R, ALPHA_LGRAD, ALPHA_RGRAD,
 NORM_SCORES, TEMP_SCORES, SCORES, NORM_SCORES_GRAD, NORM_SCORES_DROPPED, ATTN_MASKS) = range(16)

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "gaibl_init": (None, [_i, _vp]),
    "gaibl_ctx": (_vp, []),
    "gaibl_sync": (None, []),
    "gaibl_graph_from_host": (_vp, [C.c_uint32, C.c_uint32, _vp, _vp, _i]),
    "gaibl_graph_adopt": (_vp, [_vp]),
    "gaibl_graph_device": (_vp, [_vp]),
    "gaibl_graph_num_edges": (C.c_uint64, [_vp]),
    "gaibl_graph_free": (None, [_vp]),
    "gaibl_graph_set_halo": (None, [_vp, _vp, _vp, _vp, _vp]),
    "gaibl_layer_create": (_vp, [_i, _i, _i, _i, _i, _vp, _i, _f, _f, _f]),
    "gaibl_layer_free": (None, [_vp]),
    "gaibl_layer_forward": (None, [_vp, _vp]),
    "gaibl_layer_backward": (None, [_vp, _vp, _vp]),
    "gaibl_layer_update_weight": (None, [_vp, _vp]),
    "gaibl_layer_set_feat_in": (None, [_vp, _vp]),
    "gaibl_layer_set_phase": (None, [_vp, _i]),
    "gaibl_layer_set_heads": (None, [_vp, _i]),
    "gaibl_layer_set_input_constant": (None, [_vp, _i]),
    "gaibl_layer_ptr": (_vp, [_vp, _i]),
    "gaibl_sample_subgraph": (C.c_uint32, [C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint,
                                           C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "gaibl_free_host": (None, [_vp]),
    "gaibl_partition_build": (_vp, [C.c_uint32, _vp, _vp, _i, _i]),
    "gaibl_partition_build_gat": (None, [_vp, _vp, _vp]),
    "gaibl_partition_array": (C.c_int64, [_vp, _i, C.POINTER(_vp)]),
    "gaibl_partition_range": (None, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gaibl_partition_free": (None, [_vp]),
    "gaibl_partition_make_graph": (_vp, [_vp, _vp]),
    "gaibl_set_comm": (None, [_vp]),
    "gaibl_graph_halo_plan": (_vp, [_vp]),
    "gaibl_graph_set_halo_plan": (None, [_vp, _vp, _vp]),
    "gaibl_graph_set_partition_mode": (None, [_vp, _i]),
    "gaibl_graph_set_halo_link_rows": (None, [_vp, C.c_int64]),
    "gaibl_graph_set_halo_pieces": (None, [_vp, _i, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_i), _vp]),
    "gaibl_graph_halo_pieces": (_i, [_vp, _i]),
    "gaibl_graph_set_halo_consumption": (None, [_vp, _i]),
    "gaibl_graph_partition_mode": (_i, [_vp, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gaibl_adam_create": (_vp, [_f]),
    "gaibl_adam_free": (None, [_vp]),
    "gaibl_time_op": (C.c_double, [C.c_char]),
    "gaibl_reset_timers": (None, []),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    capi.load()  # libgaib_hip.so first (the rpath also finds it)
    if not LIB_PATH.exists():
        raise capi.GaibError(f"{LIB_PATH} is missing: run `python -m graphaibench_amd.build`")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def init(device: int = 0, stream: int | None = None) -> capi.Context:
    """bind the process context (gpu_context) to `device` and torch's current stream; returns a
    capi.Context view of the same gaib_ctx for direct C-ABI calls."""
    import torch

    lib = load()
    torch.cuda.set_device(device)
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream
    lib.gaibl_init(device, C.c_void_p(stream))
    ctx = capi.Context.__new__(capi.Context)
    ctx.lib = capi.load()
    ctx.device = device
    ctx.h = C.c_void_p(lib.gaibl_ctx())
    ctx.stream_ptr = int(stream or 0)  # the HIP stream every gaib call of this process is enqueued on
    ctx.close = lambda: None  # owned by the C++ side
    return ctx


def sync():
    load().gaibl_sync()


class LGraph:
    """LearningGraph*"""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_host(cls, rowptr, colidx, add_selfloop: bool):
        import numpy as np

        rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
        ci = np.ascontiguousarray(colidx, dtype=np.uint32)
        h = load().gaibl_graph_from_host(len(rp) - 1, len(ci), rp.ctypes.data, ci.ctypes.data, int(add_selfloop))
        return cls(h)

    @classmethod
    def adopt(cls, g: capi.Graph):
        """wrap a device-resident capi.Graph (ownership moves to the LearningGraph)"""
        h = load().gaibl_graph_adopt(g.h)
        g.h = None
        return cls(h)

    def device_graph(self) -> capi.Graph:
        """non-owning capi.Graph view"""
        g = capi.Graph.__new__(capi.Graph)
        g.lib = capi.load()
        g.ctx = None
        g.h = C.c_void_p(load().gaibl_graph_device(self.h))
        g.close = lambda: None
        return g

    @property
    def ne(self) -> int:
        return int(load().gaibl_graph_num_edges(self.h))

    def close(self):
        """LearningGraph::dealloc + delete: the device CSR with its caches; for a graph made by HostPartition.make_graph
        also the halo graph, the exchange plan (close it BEFORE its communicator) and the GAT structures"""
        if getattr(self, "h", None):
            load().gaibl_graph_free(self.h)
            self.h = None

    def set_halo(self, halo_graph: capi.Graph, begin, end):
        """begin(len:int, d_in:int) -> None ; end(len:int) -> int (device pointer of the halo table)"""
        BEGIN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)
        END = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int)
        self._cb = (BEGIN(lambda _u, n, p: begin(int(n), int(p or 0))), END(lambda _u, n: end(int(n))))
        self._halo_graph = halo_graph  # keep alive
        load().gaibl_graph_set_halo(self.h, halo_graph.h, C.cast(self._cb[0], C.c_void_p),
                                    C.cast(self._cb[1], C.c_void_p), None)

    PART_AUTO, PART_SPLIT, PART_CLASSES, PART_ONEPASS, PART_ONEPASS_ALL = -1, 0, 1, 2, 3
    PART_NAMES = {0: "split", 1: "classes", 2: "onepass"}

    def set_partition_mode(self, mode: int):
        """how a partitioned graph aggregates (LearningGraph::partition_mode): PART_SPLIT = the column split over all rows,
        PART_CLASSES = interior rows in one pass + column split of the boundary rows, PART_ONEPASS = interior rows in one
        pass + boundary rows in one pass over [owned | halo]; PART_AUTO = by the rule"""
        load().gaibl_graph_set_partition_mode(self.h, int(mode))

    def set_halo_link_rows(self, rows: int):
        load().gaibl_graph_set_halo_link_rows(self.h, int(rows))

    def partition_mode(self, length: int):
        """(mode, boundary rows, boundary edges): decides and builds the class graphs on first use"""
        nb, be = C.c_int64(), C.c_int64()
        m = load().gaibl_graph_partition_mode(self.h, int(length), C.byref(nb), C.byref(be))
        return m, nb.value, be.value

    def set_halo_pieces(self, n_pieces: int, ranges, wait_piece):
        """callback transports: the exchange lands in n_pieces slices; ranges = [(begin, end, piece), ...] rows of the halo table;
        wait_piece(k:int) -> int (device pointer of the table once slice k is there).  After set_halo."""
        WAIT = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int)
        self._cb_wait = WAIT(lambda _u, k: wait_piece(int(k)))
        nr = len(ranges)
        b = (C.c_int64 * max(nr, 1))(*[int(r[0]) for r in ranges])
        e = (C.c_int64 * max(nr, 1))(*[int(r[1]) for r in ranges])
        pc = (C.c_int * max(nr, 1))(*[int(r[2]) for r in ranges])
        load().gaibl_graph_set_halo_pieces(self.h, int(n_pieces), nr, b, e, pc, C.cast(self._cb_wait, C.c_void_p))

    def set_halo_consumption(self, pieces: int):
        """consume the exchange's K slices in `pieces` | K pieces (LearningGraph::set_halo_consumption); -1 = by the rule"""
        load().gaibl_graph_set_halo_consumption(self.h, int(pieces))

    def halo_pieces(self, length: int) -> int:
        """pieces the halo-column half of an aggregation of `length` columns is consumed in right now (1 = whole; after
        partition_mode)"""
        return int(load().gaibl_graph_halo_pieces(self.h, int(length)))

    def set_halo_plan(self, halo_graph: capi.Graph, plan: "capi.Halo"):
        """the exchange runs behind the C ABI (gaib_halo_exchange_begin/end inside the C++ aggregators)"""
        self._halo_graph, self._halo_plan = halo_graph, plan  # keep alive
        load().gaibl_graph_set_halo_plan(self.h, halo_graph.h, plan.h)


class Layer:
    """GCN_layer / SAGE_layer / GAT_layer"""

    def __init__(self, kind: int, level: int, nv: int, din: int, dout: int, graph: LGraph, act: bool,
                 lr: float = 0.01, feat_drop: float = 0.0, score_drop: float = 0.0):
        self.lib = load()
        self.kind, self.level, self.nv, self.din, self.dout = kind, level, nv, din, dout
        self.graph = graph
        self.h = self.lib.gaibl_layer_create(kind, level, nv, din, dout, graph.h, int(act), lr, feat_drop, score_drop)

    def close(self):
        """every device buffer of the layer and of its aggregator goes back (gconv_state::release)"""
        if getattr(self, "h", None):
            self.lib.gaibl_layer_free(self.h)
            self.h = None

    def forward(self, feat_out):
        self.lib.gaibl_layer_forward(self.h, feat_out.data_ptr())

    def backward(self, feat_out, grad_out=None):
        self.lib.gaibl_layer_backward(self.h, feat_out.data_ptr(), grad_out.data_ptr() if grad_out is not None else None)

    def update_weight(self, opt):
        self.lib.gaibl_layer_update_weight(self.h, opt)

    def set_phase(self, phase: int):
        """set_netphase: 0 TRAIN, 1 TEST, 2 VAL (dropout only while training)"""
        self.lib.gaibl_layer_set_phase(self.h, phase)

    def set_heads(self, heads: int):
        self.lib.gaibl_layer_set_heads(self.h, heads)

    def set_input_constant(self, on: bool = True):
        """the caller promises that feat_in's contents and the graph stay the same between forward calls: the layer
        keeps its aggregated input (gconv_state::set_input_constant)"""
        self.lib.gaibl_layer_set_input_constant(self.h, int(on))

    def set_feat_in(self, t):
        self._feat_keepalive = t
        self.lib.gaibl_layer_set_feat_in(self.h, t.data_ptr())

    def ptr(self, which: int) -> int:
        return self.lib.gaibl_layer_ptr(self.h, which)

    def tensor(self, which: int, shape):
        """copy of a device buffer owned by the layer"""
        import torch

        t = torch.empty(shape, dtype=torch.float32, device="cuda")
        p = self.ptr(which)
        assert p, f"layer has no buffer {which}"
        c = capi.load()
        capi._check(c.gaib_memcpy_d2d(self.lib.gaibl_ctx(), t.data_ptr(), p, t.numel() * 4), "gaib_memcpy_d2d")
        sync()
        return t

    def write(self, which: int, src):
        """overwrite a device buffer owned by the layer from a torch cuda tensor"""
        p = self.ptr(which)
        assert p, f"layer has no buffer {which}"
        src = src.contiguous()
        c = capi.load()
        capi._check(c.gaib_memcpy_d2d(self.lib.gaibl_ctx(), p, src.data_ptr(), src.numel() * 4), "gaib_memcpy_d2d")
        sync()


class HostPartition:
    """VertexRangePartition (include/gnn/partition.h): one rank's share of a vertex-range partitioned graph, built by
    the host C++ from the GLOBAL CSR without communication.  Arrays come back as numpy copies."""

    _NAMES = ("rowptr_own", "colidx_own", "rowptr_halo", "colidx_halo", "degree", "halo_gids", "halo_degree",
              "recv_counts", "send_counts", "send_idx")

    _GAT_NAMES = ("rowptr_full", "colidx_full", "rowptr_t", "colidx_t", "tperm")

    def __init__(self, rowptr, colidx, rank: int, world: int, gat: bool = False):
        import numpy as np

        rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
        ci = np.ascontiguousarray(colidx, dtype=np.uint32)
        self.lib = load()
        self.rank, self.world = rank, world
        self.h = self.lib.gaibl_partition_build(len(rp) - 1, rp.ctypes.data, ci.ctypes.data if len(ci) else None, rank,
                                                world)
        if gat:  # the [owned | halo] column space + its transpose (GAT layers need them)
            self.lib.gaibl_partition_build_gat(self.h, rp.ctypes.data, ci.ctypes.data if len(ci) else None)
        lo, hi = C.c_int64(), C.c_int64()
        self.lib.gaibl_partition_range(self.h, C.byref(lo), C.byref(hi))
        self.lo, self.hi = lo.value, hi.value
        for which, name in enumerate(self._NAMES + (self._GAT_NAMES if gat else ())):
            ptr = _vp()
            n = self.lib.gaibl_partition_array(self.h, which, C.byref(ptr))
            dt = np.uint32 if name.startswith("colidx") or name == "tperm" else np.int64
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32 if dt == np.uint32 else C.c_int64)),
                                        (n,)).copy() if n > 0 else np.zeros(0, dt)
            setattr(self, name, arr.astype(dt, copy=False))

    def make_graph(self, comm=None) -> "LGraph":
        """upload + halo plan on `comm` (capi.Comm, None for world 1) -> LearningGraph"""
        return LGraph(self.lib.gaibl_partition_make_graph(self.h, comm.h if comm is not None else None))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gaibl_partition_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_comm(comm) -> None:
    """gpu_context::set_comm: every optimizer step of the C++ layers first sums its gradient over the ranks"""
    load().gaibl_set_comm(comm.h if comm is not None else None)


def adam(lr: float):
    return load().gaibl_adam_create(lr)


def adam_free(opt) -> None:
    """delete the optimizer and its per-weight moment buffers on the device"""
    load().gaibl_adam_free(opt)


def sample_subgraph(rowptr, colidx, train_masks, n: int, m: int, seed: int):
    """host-only: GraphSAINT frontier sampling + induced subgraph (Sampler class); returns
    (sub_rowptr uint32[snv+1], sub_colidx uint32[sne], kept_ids uint32[snv])"""
    import numpy as np

    lib = load()
    rp = np.ascontiguousarray(rowptr, np.uint32)
    ci = np.ascontiguousarray(colidx, np.uint32)
    mk = np.ascontiguousarray(train_masks, np.uint8)
    a, b, c = _vp(), _vp(), _vp()
    snv = lib.gaibl_sample_subgraph(len(rp) - 1, len(ci), rp.ctypes.data, ci.ctypes.data, mk.ctypes.data, n, m, seed,
                                    C.byref(a), C.byref(b), C.byref(c))
    srp = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint32)), (snv + 1,)).copy()
    sne = int(srp[-1])
    sci = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint32)), (max(sne, 1),))[:sne].copy()
    ids = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint32)), (max(snv, 1),))[:snv].copy()
    for p in (a, b, c):
        lib.gaibl_free_host(p)
    return srp, sci, ids
