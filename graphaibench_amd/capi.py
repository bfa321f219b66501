"""ctypes binding of the C ABI in include/gaib.h (libgaib_hip.so).

This is plumbing for tests/, bench.py and the multi-GPU driver: torch supplies device memory
(`tensor.data_ptr()`), streams and torch.distributed; every compute call goes through the C ABI
into the hand-written gfx950 kernels.  There is NO fallback: if the library is missing or a call
fails, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_DIR = Path(__file__).resolve().parent / "lib"
LIB_PATH = LIB_DIR / "libgaib_hip.so"

# gaib_weight_kind
W_GCN, W_MEAN, W_MEAN_T, W_EDGE, W_EDGE_T = 0, 1, 2, 3, 4

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); kept in sync with include/gaib.h (tests/test_abi.py checks both ways)
SIGNATURES = {
    "gaib_last_error": (C.c_char_p, []),
    "gaib_version": (C.c_char_p, []),
    "gaib_ctx_create": (_i, [_i, _vp, _pp]),
    "gaib_ctx_destroy": (_i, [_vp]),
    "gaib_ctx_set_stream": (_i, [_vp, _vp]),
    "gaib_ctx_own_stream": (_i, [_vp]),
    "gaib_sync": (_i, [_vp]),
    "gaib_capture_begin": (_i, [_vp]),
    "gaib_capture_end": (_i, [_vp, _pp]),
    "gaib_capture_abort": (_i, [_vp]),
    "gaib_exec_launch": (_i, [_vp, _vp]),
    "gaib_exec_nodes": (_i64, [_vp]),
    "gaib_exec_elapsed_ms": (_i, [_vp, _vp]),
    "gaib_exec_destroy": (_i, [_vp]),
    "gaib_host_alloc": (_i, [_vp, C.c_size_t, _pp]),
    "gaib_host_free": (_i, [_vp, _vp]),
    "gaib_memcpy_d2h_async": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gaib_masked_avg_loss_dev": (_i, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "gaib_masked_accuracy_single_dev": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "gaib_masked_f1_counts_dev": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "gaib_adam_step_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp]),
    "gaib_side_begin": (_i, [_vp]),
    "gaib_side_end": (_i, [_vp]),
    "gaib_side_wait": (_i, [_vp]),
    "gaib_malloc": (_i, [_vp, C.c_size_t, _pp]),
    "gaib_free": (_i, [_vp, _vp]),
    "gaib_memcpy_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gaib_memcpy_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gaib_memcpy_d2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gaib_fill_f32": (_i, [_vp, _i64, _f, _vp]),
    "gaib_scale_f32": (_i, [_vp, _i64, _f, _vp]),
    "gaib_graph_create": (_i, [_vp, _i64, _i64, _vp, _i, _vp, _i, _pp]),
    "gaib_graph_create_rect": (_i, [_vp, _i64, _i64, _i64, _vp, _i, _vp, _i, _pp]),
    "gaib_graph_destroy": (_i, [_vp]),
    "gaib_graph_add_selfloop": (_i, [_vp, _vp, _pp]),
    "gaib_graph_nv": (_i64, [_vp]),
    "gaib_graph_ne": (_i64, [_vp]),
    "gaib_graph_nc": (_i64, [_vp]),
    "gaib_graph_rowptr": (_vp, [_vp]),
    "gaib_graph_colidx": (_vp, [_vp]),
    "gaib_graph_compute_vertex_data": (_i, [_vp, _vp]),
    "gaib_graph_vertex_data": (_vp, [_vp]),
    "gaib_graph_compute_edge_data": (_i, [_vp, _vp]),
    "gaib_graph_edge_data": (_vp, [_vp]),
    "gaib_graph_set_vertex_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gaib_graph_device_bytes": (_i64, [_vp]),
    "gaib_spmm": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "gaib_spmm_acc": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "gaib_spmm_ex": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i]),
    "gaib_spmm_gemm": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i]),
    "gaib_spmm_gemm2": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i]),
    "gaib_spmm_mh": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i]),
    "gaib_graph_split_classes": (_i, [_vp, _vp, _vp, _pp, _pp, _pp, _pp, C.POINTER(_i64), C.POINTER(_i64), _i]),
    "gaib_graph_split_pieces": (_i, [_vp, _vp, _i, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i), _pp]),
    "gaib_graph_set_row_map": (_i, [_vp, _vp, _vp, _i64]),
    "gaib_graph_row_map": (_vp, [_vp]),
    "gaib_spmm_2t": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _i]),
    "gaib_spmm_gemm_2t": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _vp, _vp, _i, _vp, _i]),
    "gaib_spmm_gemm_fusable": (_i, [_vp, _i, _i, _i, _i]),
    "gaib_gat_scores_mh": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "gaib_sddmm_mh": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gaib_gat_softmax_bwd_alpha_mh": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "gaib_gat_softmax_bwd_alpha_ex": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gaib_gat_softmax_bwd_alpha_re": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gaib_gat_softmax_bwd_rows": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp]),
    "gaib_edge_gather_perm": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "gaib_edge_rowsum": (_i, [_vp, _vp, _i, _vp, _vp]),
    "gaib_gat_alpha_grads": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gaib_gat_backward_fused": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "gaib_gat_forward_fused": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _f, _i, _vp, _vp]),
    "gaib_edge_transpose_mh": (_i, [_vp, _vp, _i, _vp, _vp]),
    "gaib_gat_scores": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "gaib_sddmm": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "gaib_gat_softmax_bwd_alpha": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "gaib_edge_transpose": (_i, [_vp, _vp, _vp, _vp]),
    "gaib_sgemm": (_i, [_vp, _i, _i, _i64, _i64, _i64, _vp, _vp, _i, _vp]),
    "gaib_sgemm_drelu": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i, _vp]),
    "gaib_sgemm_ex": (_i, [_vp, _i, _i, _i64, _i64, _i64, _vp, _vp, _i, _vp]),
    "gaib_bias_add": (_i, [_vp, _i64, _i, _vp, _vp]),
    "gaib_colsum": (_i, [_vp, _i64, _i, _vp, _vp]),
    "gaib_rng_uniform": (_i, [_vp, _i64, _f, _f, _u64, _vp]),
    "gaib_csr2csc": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gaib_relu": (_i, [_vp, _i64, _vp, _vp]),
    "gaib_d_relu": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "gaib_dropout": (_i, [_vp, _i64, _f, _f, _u64, _vp, _vp, _vp]),
    "gaib_d_dropout": (_i, [_vp, _i64, _f, _vp, _vp, _vp]),
    "gaib_softmax_xent": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gaib_d_softmax_xent": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gaib_sigmoid_xent": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gaib_d_sigmoid_xent": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gaib_masked_f1_micro": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, C.POINTER(_f), C.POINTER(_i64)]),
    "gaib_masked_avg_loss": (_i, [_vp, _i64, _i64, _vp, _vp, C.POINTER(_f)]),
    "gaib_masked_accuracy_single": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, C.POINTER(_f)]),
    "gaib_l2norm": (_i, [_vp, _i64, _i, _vp, _vp]),
    "gaib_d_l2norm": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "gaib_adam_step": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f]),
    "gaib_gather_rows": (_i, [_vp, _i64, _vp, _i, _vp, _vp]),
    "gaib_prof_enable": (_i, [_vp, _i]),
    "gaib_prof_reset": (_i, [_vp]),
    "gaib_prof_get": (_i, [_vp, C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "gaib_prof_get_work": (_i, [_vp, C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gaib_prof_table": (_i, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gaib_gat_score_signs": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "gaib_gat_forward_fused_rect": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, C.c_float, _i, _vp, _vp, _i]),
    "gaib_gat_backward_rec": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "gaib_gat_backward_fused_rect": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _i]),
    "gaib_gather_scatter_rows": (_i, [_vp, _i64, _vp, _vp, _i, _vp, _vp]),
    "gaib_graph_reorder": (_i, [_vp, _vp, _i, _pp, _vp, _vp]),
    "gaib_graph_sort_rows": (_i, [_vp, _vp]),
    "gaib_graph_locality": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "gaib_graph_stats": (_i, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gaib_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "gaib_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "gaib_device_count": (_i, [C.POINTER(_i)]),
    "gaib_comm_transport_available": (_i, [_i]),
    "gaib_comm_unique_id": (_i, [_i, _vp]),
    "gaib_comm_init": (_i, [_vp, _i, _i, _vp, _i, _pp]),
    "gaib_comm_destroy": (_i, [_vp]),
    "gaib_comm_rank": (_i, [_vp]),
    "gaib_comm_size": (_i, [_vp]),
    "gaib_comm_barrier": (_i, [_vp]),
    "gaib_allreduce_f32": (_i, [_vp, _vp, _i64]),
    "gaib_allreduce_host_f64": (_i, [_vp, C.POINTER(C.c_double), _i]),
    "gaib_halo_create": (_i, [_vp, C.POINTER(_i64), _vp, _i, C.POINTER(_i64), _pp]),
    "gaib_halo_destroy": (_i, [_vp]),
    "gaib_halo_rows": (_i64, [_vp]),
    "gaib_halo_send_rows": (_i64, [_vp]),
    "gaib_halo_bytes_sent": (_i64, [_vp]),
    "gaib_halo_send_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)]),
    "gaib_halo_link_rows": (_i64, [_vp]),
    "gaib_halo_exchange_begin": (_i, [_vp, _i, _vp]),
    "gaib_halo_exchange_end": (_i, [_vp, _pp]),
    "gaib_halo_set_pieces": (_i, [_vp, _i]),
    "gaib_halo_pieces": (_i, [_vp]),
    "gaib_halo_default_pieces": (_i, [_i64, _i]),
    "gaib_halo_piece_slice": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "gaib_halo_piece_ranges": (_i, [_vp, _i, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)]),
    "gaib_halo_exchange_wait_piece": (_i, [_vp, _i, _pp]),
    "gaib_halo_reduce": (_i, [_vp, _i, _vp, _vp]),
    "gaib_probe_stream_copy": (_i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
    "gaib_probe_peer_copy": (_i, [_i, _i, C.c_size_t, _i, _i, C.POINTER(C.c_double)]),
}


class GaibError(RuntimeError):
    pass


COMM_RCCL, COMM_IPC = 0, 1
ORDER_DEGREE, ORDER_BFS, ORDER_CM = 1, 2, 3
COMM_ID_BYTES = 128


def parse_prof_table(text: str) -> dict:
    """lines "key count total_ms alg_bytes flops roof_ms" (gaib_prof_table; the trainer prints them after "[gaib prof]")"""
    out = {}
    for line in text.splitlines():
        f = line.split()
        if len(f) == 6 and f[0] not in ("epochs", "epoch_seconds") and f[1].isdigit():
            out[f[0]] = dict(count=int(f[1]), ms=float(f[2]), bytes=float(f[3]), flops=float(f[4]), roof_ms=float(f[5]))
    return out


def comm_transport_available(transport: int = COMM_RCCL) -> bool:
    """local and cheap: can this process use the transport (RCCL: librccl loads with every entry point)"""
    return load().gaib_comm_transport_available(transport) == 0


def comm_unique_id(transport: int = COMM_RCCL) -> bytes:
    """called by ONE rank; the bytes travel to the others by whatever the launcher has (a file, torch's store ...)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().gaib_comm_unique_id(transport, buf), "gaib_comm_unique_id")
    return buf.raw


class Comm:
    """gaib_comm: the collectives of the partitioned path behind the C ABI (RCCL, or peer-to-peer pull over hipIpc)"""

    def __init__(self, ctx: "Context", rank: int, nranks: int, unique_id: bytes, transport: int = COMM_RCCL):
        self.lib = load()
        self.ctx, self.rank, self.nranks, self.transport = ctx, rank, nranks, transport
        h = C.c_void_p()
        _check(self.lib.gaib_comm_init(ctx.h, rank, nranks, C.c_char_p(unique_id), transport, C.byref(h)), "gaib_comm_init")
        self.h = h

    @property
    def size(self) -> int:
        """ranks of the communicator as the transport itself reports them (RCCL: ncclCommCount)"""
        return int(self.lib.gaib_comm_size(self.h))

    def barrier(self):
        _check(self.lib.gaib_comm_barrier(self.h), "gaib_comm_barrier")

    def allreduce(self, t):
        """in-place sum of a float32 device tensor"""
        _check(self.lib.gaib_allreduce_f32(self.h, t.data_ptr(), t.numel()), "gaib_allreduce_f32")

    def allreduce_host(self, values):
        arr = (C.c_double * len(values))(*values)
        _check(self.lib.gaib_allreduce_host_f64(self.h, arr, len(values)), "gaib_allreduce_host_f64")
        return list(arr)

    def halo(self, send_counts, send_idx, recv_counts) -> "Halo":
        return Halo(self, send_counts, send_idx, recv_counts)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gaib_comm_destroy(self.h)
            self.h = None


class Halo:
    """gaib_halo: one exchange plan (which rows go to / come from which rank)"""

    def __init__(self, comm: Comm, send_counts, send_idx, recv_counts):
        self.comm, self.lib = comm, comm.lib
        n = comm.nranks
        sc = (C.c_int64 * n)(*[int(v) for v in send_counts])
        rc = (C.c_int64 * n)(*[int(v) for v in recv_counts])
        on_dev = int(hasattr(send_idx, "is_cuda") and send_idx.is_cuda)
        if hasattr(send_idx, "data_ptr"):
            ptr = send_idx.data_ptr() if send_idx.numel() else None
            self._keep = send_idx
        else:
            import numpy as np
            self._keep = np.ascontiguousarray(send_idx, dtype=np.int64)
            ptr = self._keep.ctypes.data if len(self._keep) else None
        h = C.c_void_p()
        _check(self.lib.gaib_halo_create(comm.h, sc, ptr, on_dev, rc, C.byref(h)), "gaib_halo_create")
        self.h = h
        self.rows = int(self.lib.gaib_halo_rows(h))

    def begin(self, rows, length: int):
        _check(self.lib.gaib_halo_exchange_begin(self.h, length, _ptr(rows)), "gaib_halo_exchange_begin")

    def end(self) -> int:
        p = C.c_void_p()
        _check(self.lib.gaib_halo_exchange_end(self.h, C.byref(p)), "gaib_halo_exchange_end")
        return p.value or 0

    def set_pieces(self, n_pieces: int):
        """the exchange in n_pieces time slices (gaib_halo_set_pieces; the same on every rank)"""
        _check(self.lib.gaib_halo_set_pieces(self.h, int(n_pieces)), "gaib_halo_set_pieces")

    @property
    def pieces(self) -> int:
        return int(self.lib.gaib_halo_pieces(self.h))

    def piece_ranges(self, piece: int):
        """[(begin, end), ...] rows of the halo table that slice `piece` fills (gaib_halo_piece_ranges)"""
        cap = self.comm.nranks
        b, e, n = (C.c_int64 * cap)(), (C.c_int64 * cap)(), _i()
        _check(self.lib.gaib_halo_piece_ranges(self.h, piece, cap, b, e, C.byref(n)), "gaib_halo_piece_ranges")
        return [(int(b[j]), int(e[j])) for j in range(n.value)]

    def wait_piece(self, piece: int) -> int:
        p = C.c_void_p()
        _check(self.lib.gaib_halo_exchange_wait_piece(self.h, piece, C.byref(p)), "gaib_halo_exchange_wait_piece")
        return p.value or 0

    def reduce(self, halo_rows, rows, length: int):
        """rows[send_idx[k]] += what the peers hold for this rank's vertices (the reverse exchange)"""
        _check(self.lib.gaib_halo_reduce(self.h, length, _ptr(halo_rows), _ptr(rows)), "gaib_halo_reduce")

    @property
    def bytes_sent(self) -> int:
        return int(self.lib.gaib_halo_bytes_sent(self.h))

    def send_stats(self) -> dict:
        """dict(packs, direct_sends, direct_peers): gaib_halo_send_stats"""
        a, b, n = _i64(), _i64(), _i()
        _check(self.lib.gaib_halo_send_stats(self.h, C.byref(a), C.byref(b), C.byref(n)), "gaib_halo_send_stats")
        return dict(packs=a.value, direct_sends=b.value, direct_peers=n.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gaib_halo_destroy(self.h)
            self.h = None


def probe_peer_copy(src_dev: int, dst_dev: int, nbytes: int = 1 << 28, iters: int = 10, bidir: bool = False) -> float:
    """GB/s per direction of hipMemcpyPeerAsync between two visible devices (the xGMI link probe)"""
    v = C.c_double(0.0)
    _check(load().gaib_probe_peer_copy(src_dev, dst_dev, nbytes, iters, int(bidir), C.byref(v)), "gaib_probe_peer_copy")
    return v.value


_lib = None


def load() -> C.CDLL:
    """dlopen libgaib_hip.so and attach prototypes.  Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise GaibError(
            f"{LIB_PATH} is missing: build it with `python -m graphaibench_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gaib_last_error().decode(errors="replace")
        raise GaibError(f"{what} failed (status {rc}): {msg}")


def _ptr(x) -> int | None:
    """torch tensor / int / None -> raw pointer."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


class Exec:
    """a recorded call sequence (gaib_capture_begin / end)"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    @property
    def nodes(self) -> int:
        return self.ctx.lib.gaib_exec_nodes(self.h)

    def launch(self):
        _check(self.ctx.lib.gaib_exec_launch(self.ctx.h, self.h), "gaib_exec_launch")

    def elapsed_ms(self) -> float:
        """device time of the last launch (call after ctx.sync())"""
        v = C.c_float()
        _check(self.ctx.lib.gaib_exec_elapsed_ms(self.h, C.byref(v)), "gaib_exec_elapsed_ms")
        return v.value

    def close(self):
        if self.h:
            self.ctx.lib.gaib_exec_destroy(self.h)
            self.h = None


class Context:
    """gaib_ctx bound to one device and one HIP stream (default: torch's current stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        import torch

        self.lib = load()
        self.device = device
        torch.cuda.set_device(device)
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        _check(self.lib.gaib_ctx_create(device, C.c_void_p(stream), C.byref(h)), "gaib_ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.gaib_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self.lib.gaib_sync(self.h), "gaib_sync")

    def side_begin(self):
        _check(self.lib.gaib_side_begin(self.h), "gaib_side_begin")

    def side_end(self):
        _check(self.lib.gaib_side_end(self.h), "gaib_side_end")

    def side_wait(self):
        _check(self.lib.gaib_side_wait(self.h), "gaib_side_wait")

    # ---- HIP graphs: record a call sequence, replay it with one launch ---------------------------
    def own_stream(self):
        """switch to a stream created by the context (the null stream cannot be recorded)"""
        _check(self.lib.gaib_ctx_own_stream(self.h), "gaib_ctx_own_stream")

    def capture_begin(self):
        _check(self.lib.gaib_capture_begin(self.h), "gaib_capture_begin")

    def capture_end(self) -> "Exec":
        x = C.c_void_p()
        _check(self.lib.gaib_capture_end(self.h, C.byref(x)), "gaib_capture_end")
        return Exec(self, x)

    def capture_abort(self):
        _check(self.lib.gaib_capture_abort(self.h), "gaib_capture_abort")

    def set_option(self, key: str, value: int):
        _check(self.lib.gaib_set_option(self.h, key.encode(), int(value)), f"gaib_set_option({key})")

    def get_option(self, key: str) -> int:
        v = _i64()
        _check(self.lib.gaib_get_option(self.h, key.encode(), C.byref(v)), f"gaib_get_option({key})")
        return v.value

    # ---- in-stream kernel timing ----------------------------------------------------------
    def prof_enable(self, on: bool = True):
        _check(self.lib.gaib_prof_enable(self.h, int(on)), "gaib_prof_enable")

    def prof_reset(self):
        _check(self.lib.gaib_prof_reset(self.h), "gaib_prof_reset")

    def prof_get(self, key: str):
        n, ms = _i64(), C.c_double()
        _check(self.lib.gaib_prof_get(self.h, key.encode(), C.byref(n), C.byref(ms)), "gaib_prof_get")
        return n.value, ms.value

    def prof_table(self) -> dict:
        """{key: dict(count, ms, bytes, flops, roof_ms)} of every key that has records (gaib_prof_table)"""
        need = C.c_size_t(0)
        _check(self.lib.gaib_prof_table(self.h, None, 0, C.byref(need)), "gaib_prof_table")
        buf = C.create_string_buffer(need.value + 16)
        _check(self.lib.gaib_prof_table(self.h, buf, len(buf), C.byref(need)), "gaib_prof_table")
        return parse_prof_table(buf.value.decode())

    def probe_stream_copy(self, nbytes: int = 1 << 30, iters: int = 20) -> float:
        """GB/s (read + written bytes) of a 16-B-per-lane copy kernel on this context's stream"""
        v = C.c_double(0.0)
        _check(load().gaib_probe_stream_copy(self.h, nbytes, iters, C.byref(v)), "gaib_probe_stream_copy")
        return v.value

    def graph_stats(self, g: "Graph"):
        a, b, c = _i64(), _i64(), _i64()
        _check(self.lib.gaib_graph_stats(self.h, g.h, C.byref(a), C.byref(b), C.byref(c)), "gaib_graph_stats")
        return dict(n_heavy=a.value, heavy_edges=b.value, max_degree=c.value)

    # ---- graph ------------------------------------------------------------------------
    def graph(self, rowptr, colidx, ncols: int | None = None) -> "Graph":
        return Graph(self, rowptr, colidx, ncols)

    # ---- aggregation --------------------------------------------------------------------
    def spmm(self, g: "Graph", kind: int, x, out, edge_w=None, accumulate: bool = False, relu: bool = False,
             heads: int = 1):
        assert x.is_contiguous() and out.is_contiguous() and x.dim() == 2
        flags = (1 if accumulate else 0) | (2 if relu else 0)
        _check(self.lib.gaib_spmm_mh(self.h, g.h, kind, _ptr(edge_w), heads, x.shape[1], _ptr(x), _ptr(out), flags),
               "gaib_spmm")
        return out

    def spmm_gemm(self, g: "Graph", kind: int, x, agg, W, out, transW: bool = False, relu: bool = False,
                  agg_scratch: bool = False, edge_w=None, accumulate: bool = False, rows2=None, W2=None):
        """agg = A.x ; out = act(agg . op(W)) (gaib_spmm_gemm: fused on the matrix cores when the shape allows)"""
        assert x.is_contiguous() and agg.is_contiguous() and W.is_contiguous() and out.is_contiguous()
        len_in, len_out = agg.shape[1], out.shape[1]
        assert tuple(W.shape) == ((len_out, len_in) if transW else (len_in, len_out))
        flags = (2 if relu else 0) | (4 if agg_scratch else 0) | (1 if accumulate else 0)
        if rows2 is not None:  # + rows2 . op(W2) in the same store (gaib_spmm_gemm2)
            assert rows2.is_contiguous() and W2.is_contiguous() and W2.shape == W.shape and rows2.shape[1] == len_in
            _check(self.lib.gaib_spmm_gemm2(self.h, g.h, kind, _ptr(edge_w), len_in, _ptr(x), _ptr(agg), _ptr(W),
                                            1 if transW else 0, _ptr(rows2), _ptr(W2), len_out, _ptr(out), flags),
                   "gaib_spmm_gemm2")
            return out
        _check(self.lib.gaib_spmm_gemm(self.h, g.h, kind, _ptr(edge_w), len_in, _ptr(x), _ptr(agg), _ptr(W),
                                       1 if transW else 0, len_out, _ptr(out), flags), "gaib_spmm_gemm")
        return out

    def spmm_2t(self, g: "Graph", kind: int, x, x2, n_first: int, out, edge_w=None, accumulate: bool = False,
                relu: bool = False):
        """aggregation over two feature tables: column ids >= n_first index x2 (gaib_spmm_2t)"""
        assert x.is_contiguous() and out.is_contiguous() and (x2 is None or x2.is_contiguous())
        flags = (1 if accumulate else 0) | (2 if relu else 0)
        _check(self.lib.gaib_spmm_2t(self.h, g.h, kind, _ptr(edge_w), x.shape[1], _ptr(x), _ptr(x2), n_first, _ptr(out),
                                     flags), "gaib_spmm_2t")
        return out

    def spmm_gemm_2t(self, g: "Graph", kind: int, x, x2, n_first: int, agg, W, out, transW: bool = False,
                     relu: bool = False, agg_scratch: bool = False, edge_w=None, accumulate: bool = False, rows2=None,
                     W2=None):
        """gaib_spmm_gemm_2t: the fused aggregation + product over two feature tables"""
        len_in, len_out = agg.shape[1], out.shape[1]
        flags = (2 if relu else 0) | (4 if agg_scratch else 0) | (1 if accumulate else 0)
        _check(self.lib.gaib_spmm_gemm_2t(self.h, g.h, kind, _ptr(edge_w), len_in, _ptr(x), _ptr(x2), n_first, _ptr(agg),
                                          _ptr(W), 1 if transW else 0, _ptr(rows2), _ptr(W2), len_out, _ptr(out), flags),
               "gaib_spmm_gemm_2t")
        return out

    def spmm_gemm_fusable(self, kind: int, len_in: int, len_out: int, dual: bool = False) -> bool:
        return bool(self.lib.gaib_spmm_gemm_fusable(self.h, kind, len_in, len_out, int(dual)))

    def split_classes(self, g_own: "Graph", g_halo: "Graph", interior=True, bnd_own=True, bnd_halo=True, bnd_full=True,
                      all_boundary: bool = False):
        """gaib_graph_split_classes -> dict(interior=, bnd_own=, bnd_halo=, bnd_full= Graph or None, n_boundary=,
        boundary_edges=)"""
        hs = [C.c_void_p() for _ in range(4)]
        want = [interior, bnd_own, bnd_halo, bnd_full]
        nb, be = _i64(), _i64()
        _check(self.lib.gaib_graph_split_classes(self.h, g_own.h, g_halo.h, *[C.byref(h) if w else None for h, w in zip(hs, want)],
                                                 C.byref(nb), C.byref(be), int(all_boundary)), "gaib_graph_split_classes")
        out = {k: (Graph(self, _handle=h) if w else None)
               for k, h, w in zip(("interior", "bnd_own", "bnd_halo", "bnd_full"), hs, want)}
        out["n_boundary"], out["boundary_edges"] = nb.value, be.value
        return out

    def split_pieces(self, g: "Graph", n_pieces: int, ranges):
        """gaib_graph_split_pieces: ranges = [(begin, end, piece), ...] over g's column space -> [Graph] * n_pieces"""
        nr = len(ranges)
        b = (C.c_int64 * max(nr, 1))(*[int(r[0]) for r in ranges])
        e = (C.c_int64 * max(nr, 1))(*[int(r[1]) for r in ranges])
        pc = (C.c_int * max(nr, 1))(*[int(r[2]) for r in ranges])
        hs = (C.c_void_p * n_pieces)()
        _check(self.lib.gaib_graph_split_pieces(self.h, g.h, n_pieces, nr, b, e, pc, hs), "gaib_graph_split_pieces")
        return [Graph(self, _handle=C.c_void_p(hs[k])) for k in range(n_pieces)]

    def gat_scores(self, g, h, alpha_l, alpha_r, temp, scores, norm, eps: float = 0.2, heads: int = 1):
        _check(self.lib.gaib_gat_scores_mh(self.h, g.h, h.shape[1], heads, _ptr(h), _ptr(alpha_l), _ptr(alpha_r),
                                           eps, _ptr(temp), _ptr(scores), _ptr(norm)), "gaib_gat_scores")

    def gat_score_signs(self, g, h, alpha_l, alpha_r, heads: int = 1):
        """[ne x heads] uint8: (pre-activation score > 0) exactly as the one-sweep kernels form it (test / diagnostic)"""
        import torch

        out = torch.empty(g.ne, heads, dtype=torch.uint8, device=h.device)
        _check(self.lib.gaib_gat_score_signs(self.h, g.h, h.shape[1], heads, _ptr(h), _ptr(alpha_l), _ptr(alpha_r), _ptr(out)),
               "gaib_gat_score_signs")
        return out

    def sddmm(self, g, grad, feat, out_e, heads: int = 1):
        _check(self.lib.gaib_sddmm_mh(self.h, g.h, grad.shape[1], heads, _ptr(grad), _ptr(feat), _ptr(out_e)),
               "gaib_sddmm")

    def gat_softmax_bwd_alpha(self, g, feat, norm, norm_grad, temp, scores, lgrad, rgrad, eps: float = 0.2,
                              heads: int = 1, grad_rows=None, fwd_out_rows=None, norm_t=None, alpha=None):
        """scores may be None; grad_rows + fwd_out_rows select the one-pass form (gaib_gat_softmax_bwd_alpha_ex);
        temp=None with alpha=(alpha_l, alpha_r): the form without the temp array (gaib_gat_softmax_bwd_alpha_re)"""
        if temp is None:
            _check(self.lib.gaib_gat_softmax_bwd_alpha_re(self.h, g.h, feat.shape[1], heads, _ptr(feat), _ptr(alpha[0]),
                                                          _ptr(alpha[1]), _ptr(norm), _ptr(norm_grad), eps, _ptr(scores),
                                                          _ptr(lgrad), _ptr(rgrad), _ptr(grad_rows),
                                                          _ptr(fwd_out_rows), _ptr(norm_t)), "gaib_gat_softmax_bwd_alpha_re")
            return
        _check(self.lib.gaib_gat_softmax_bwd_alpha_ex(self.h, g.h, feat.shape[1], heads, _ptr(feat), _ptr(norm),
                                                      _ptr(norm_grad), _ptr(temp), eps, _ptr(scores),
                                                      _ptr(lgrad), _ptr(rgrad), _ptr(grad_rows),
                                                      _ptr(fwd_out_rows), _ptr(norm_t)), "gaib_gat_softmax_bwd_alpha")

    def gat_forward_fused(self, g, h, alpha_l, alpha_r, out, row_stats, eps: float = 0.2, heads: int = 1,
                          relu: bool = False) -> bool:
        """one-sweep forward; row_stats [nv, heads, 2] receives (row max, 1 / row sum).  False = not applicable"""
        rc = self.lib.gaib_gat_forward_fused(self.h, g.h, h.shape[1], heads, _ptr(h), _ptr(alpha_l), _ptr(alpha_r), eps,
                                             int(relu), _ptr(out), _ptr(row_stats))
        if rc == -5:
            return False
        _check(rc, "gaib_gat_forward_fused")
        return True

    def gat_backward_fused(self, g, feat, grad, fwd_out, alpha_l, alpha_r, norm, grad_out, lgrad, rgrad, eps: float = 0.2,
                           heads: int = 1, row_stats=None) -> bool:
        """False when the fused path does not apply (GAIB_ERR_UNSUPPORTED, nothing touched).  row_stats (from
        gat_forward_fused): the attention is formed again, norm may be None"""
        rc = self.lib.gaib_gat_backward_fused(self.h, g.h, feat.shape[1], heads, _ptr(feat), _ptr(grad), _ptr(fwd_out),
                                              _ptr(alpha_l), _ptr(alpha_r), _ptr(norm), _ptr(row_stats), eps,
                                              _ptr(grad_out), _ptr(lgrad), _ptr(rgrad))
        if rc == -5:
            return False
        _check(rc, "gaib_gat_backward_fused")
        return True

    def edge_transpose(self, g, in_e, out_e, heads: int = 1):
        _check(self.lib.gaib_edge_transpose_mh(self.h, g.h, heads, _ptr(in_e), _ptr(out_e)), "gaib_edge_transpose")

    # ---- dense ----------------------------------------------------------------------------
    def sgemm_drelu(self, A, G, mask, Cm, accum=False):
        """G <- G * (mask > 0) in place; Cm (+)= A^T . G   (A [K x M], G / mask [K x N])"""
        K, M = A.shape
        assert G.shape == mask.shape and G.shape[0] == K and tuple(Cm.shape) == (M, G.shape[1])
        _check(self.lib.gaib_sgemm_drelu(self.h, M, G.shape[1], K, _ptr(A), _ptr(G), _ptr(mask), 1 if accum else 0,
                                         _ptr(Cm)), "gaib_sgemm_drelu")

    def sgemm(self, A, B, Cm, transA=False, transB=False, accum=False, relu=False):
        """row-major C[M x N] (=|+=) op(A) . op(B); shapes follow the reference's matmul()."""
        M, N = Cm.shape
        K = A.shape[0] if transA else A.shape[1]
        flags = (1 if accum else 0) | (2 if relu else 0)
        _check(self.lib.gaib_sgemm_ex(self.h, int(transA), int(transB), M, N, K, _ptr(A), _ptr(B),
                                      flags, _ptr(Cm)), "gaib_sgemm")
        return Cm

    # ---- elementwise / loss / optimizer -----------------------------------------------------
    def bias_add(self, x, b):
        _check(self.lib.gaib_bias_add(self.h, x.shape[0], x.shape[1], _ptr(x), _ptr(b)), "gaib_bias_add")

    def colsum(self, x, a):
        _check(self.lib.gaib_colsum(self.h, x.shape[0], x.shape[1], _ptr(x), _ptr(a)), "gaib_colsum")

    def rng_uniform(self, out, a: float = 0.0, b: float = 1.0, seed: int = 1):
        _check(self.lib.gaib_rng_uniform(self.h, out.numel(), a, b, seed, _ptr(out)), "gaib_rng_uniform")

    def csr2csc(self, nrows, ncols, values, rowptr, colidx, valuesT, rowptrT, colidxT):
        _check(self.lib.gaib_csr2csc(self.h, nrows, ncols, colidx.numel(), _ptr(values), _ptr(rowptr), _ptr(colidx),
                                     _ptr(valuesT), _ptr(rowptrT), _ptr(colidxT)), "gaib_csr2csc")

    def relu(self, x, out):
        _check(self.lib.gaib_relu(self.h, x.numel(), _ptr(x), _ptr(out)), "gaib_relu")

    def d_relu(self, grad, data, out):
        _check(self.lib.gaib_d_relu(self.h, grad.numel(), _ptr(grad), _ptr(data), _ptr(out)), "gaib_d_relu")

    def dropout(self, x, masks, out, rate: float, seed: int):
        scale = 1.0 / (1.0 - rate)
        _check(self.lib.gaib_dropout(self.h, x.numel(), scale, rate, seed, _ptr(x), _ptr(masks), _ptr(out)),
               "gaib_dropout")

    def d_dropout(self, x, masks, out, rate: float):
        scale = 1.0 / (1.0 - rate)
        _check(self.lib.gaib_d_dropout(self.h, x.numel(), scale, _ptr(x), _ptr(masks), _ptr(out)),
               "gaib_d_dropout")

    def softmax_xent(self, logits, labels, loss, probs, begin, end, masks=None):
        _check(self.lib.gaib_softmax_xent(self.h, logits.shape[1], begin, end, _ptr(logits), _ptr(masks),
                                          _ptr(labels), _ptr(loss), _ptr(probs)), "gaib_softmax_xent")

    def d_softmax_xent(self, probs, labels, diff, begin, end, masks=None):
        _check(self.lib.gaib_d_softmax_xent(self.h, probs.shape[1], begin, end, _ptr(masks), _ptr(labels),
                                            _ptr(probs), _ptr(diff)), "gaib_d_softmax_xent")

    def sigmoid_xent(self, logits, labels, loss, probs, begin, end, masks=None):
        _check(self.lib.gaib_sigmoid_xent(self.h, logits.shape[1], begin, end, _ptr(logits), _ptr(masks),
                                          _ptr(labels), _ptr(loss), _ptr(probs)), "gaib_sigmoid_xent")

    def d_sigmoid_xent(self, probs, labels, diff, begin, end, masks=None):
        _check(self.lib.gaib_d_sigmoid_xent(self.h, probs.shape[1], begin, end, _ptr(masks), _ptr(labels),
                                            _ptr(probs), _ptr(diff)), "gaib_d_sigmoid_xent")

    def masked_f1_micro(self, preds, labels, begin, end, masks=None):
        """-> (f1_micro, (tp, fp, fn))"""
        r = _f(0.0)
        cnt = (_i64 * 3)()
        _check(self.lib.gaib_masked_f1_micro(self.h, begin, end, preds.shape[1], _ptr(masks), _ptr(preds),
                                             _ptr(labels), C.byref(r), cnt), "gaib_masked_f1_micro")
        return float(r.value), tuple(int(x) for x in cnt)

    def masked_avg_loss(self, loss, begin, end, masks=None) -> float:
        r = C.c_float()
        _check(self.lib.gaib_masked_avg_loss(self.h, begin, end, _ptr(masks), _ptr(loss), C.byref(r)),
               "gaib_masked_avg_loss")
        return r.value

    def masked_accuracy_single(self, preds, labels, begin, end, masks=None) -> float:
        r = C.c_float()
        _check(self.lib.gaib_masked_accuracy_single(self.h, begin, end, preds.shape[1], _ptr(masks),
                                                    _ptr(preds), _ptr(labels), C.byref(r)),
               "gaib_masked_accuracy_single")
        return r.value

    def l2norm(self, x, out):
        _check(self.lib.gaib_l2norm(self.h, x.shape[0], x.shape[1], _ptr(x), _ptr(out)), "gaib_l2norm")

    def d_l2norm(self, feat, grad, out):
        _check(self.lib.gaib_d_l2norm(self.h, feat.shape[0], feat.shape[1], _ptr(feat), _ptr(grad), _ptr(out)),
               "gaib_d_l2norm")

    def adam_step(self, dW, W, m, v, alpha, b1_t, b2_t, b1=0.9, b2=0.999, eps=1e-8):
        _check(self.lib.gaib_adam_step(self.h, W.numel(), _ptr(dW), _ptr(W), _ptr(m), _ptr(v), alpha, b1, b2,
                                       b1_t, b2_t, eps), "gaib_adam_step")

    def adam_step_dev(self, dW, W, m, v, alpha, d_pow, b1=0.9, b2=0.999, eps=1e-8):
        """beta powers in device memory (d_pow = [b1^t, b2^t]), advanced there: the recordable form"""
        _check(self.lib.gaib_adam_step_dev(self.h, W.numel(), _ptr(dW), _ptr(W), _ptr(m), _ptr(v), alpha, b1, b2, eps,
                                           _ptr(d_pow)), "gaib_adam_step_dev")

    def masked_avg_loss_dev(self, loss, begin, end, d_result, masks=None):
        _check(self.lib.gaib_masked_avg_loss_dev(self.h, begin, end, _ptr(masks), _ptr(loss), _ptr(d_result)),
               "gaib_masked_avg_loss_dev")

    def masked_accuracy_single_dev(self, preds, labels, begin, end, d_result, masks=None):
        _check(self.lib.gaib_masked_accuracy_single_dev(self.h, begin, end, preds.shape[1], _ptr(masks), _ptr(preds),
                                                        _ptr(labels), _ptr(d_result)), "gaib_masked_accuracy_single_dev")

    def graph_locality(self, g) -> float:
        v = C.c_float()
        _check(self.lib.gaib_graph_locality(self.h, g.h, C.byref(v)), "gaib_graph_locality")
        return float(v.value)

    def gather_scatter_rows(self, src_idx, dst_idx, x, out):
        _check(self.lib.gaib_gather_scatter_rows(self.h, src_idx.numel(), _ptr(src_idx), _ptr(dst_idx), x.shape[1], _ptr(x),
                                                 _ptr(out)), "gaib_gather_scatter_rows")

    def gather_rows(self, idx, x, out):
        _check(self.lib.gaib_gather_rows(self.h, idx.numel(), _ptr(idx), x.shape[1], _ptr(x), _ptr(out)),
               "gaib_gather_rows")


class Graph:
    """gaib_graph: CSR resident in HBM.  rowptr: int64 or int32/uint32 tensor/array [nv+1];
    colidx: int32/uint32 [ne].  Host (numpy / cpu tensor) or device (cuda tensor) sources."""

    def __init__(self, ctx: Context, rowptr=None, colidx=None, ncols: int | None = None, _handle=None):
        import numpy as np
        import torch

        self.ctx = ctx
        self.lib = ctx.lib
        if _handle is not None:
            self.h = _handle
            return
        if isinstance(rowptr, np.ndarray):
            rowptr = torch.from_numpy(np.ascontiguousarray(rowptr))
        if isinstance(colidx, np.ndarray):
            colidx = torch.from_numpy(np.ascontiguousarray(colidx.view(np.int32) if colidx.dtype == np.uint32 else colidx))
        assert rowptr.dtype in (torch.int64, torch.int32), rowptr.dtype
        assert colidx.dtype == torch.int32, colidx.dtype
        assert rowptr.is_cuda == colidx.is_cuda or colidx.numel() == 0
        rowptr = rowptr.contiguous()
        colidx = colidx.contiguous()
        nv = rowptr.numel() - 1
        ne = colidx.numel()
        bits = 64 if rowptr.dtype == torch.int64 else 32
        h = C.c_void_p()
        if ncols is None:
            rc = self.lib.gaib_graph_create(ctx.h, nv, ne, _ptr(rowptr), bits, _ptr(colidx),
                                            int(rowptr.is_cuda), C.byref(h))
        else:
            rc = self.lib.gaib_graph_create_rect(ctx.h, nv, ncols, ne, _ptr(rowptr), bits, _ptr(colidx),
                                                 int(rowptr.is_cuda), C.byref(h))
        _check(rc, "gaib_graph_create")
        self.h = h

    @property
    def nv(self) -> int:
        return self.lib.gaib_graph_nv(self.h)

    @property
    def ne(self) -> int:
        return self.lib.gaib_graph_ne(self.h)

    @property
    def nc(self) -> int:
        return self.lib.gaib_graph_nc(self.h)

    def add_selfloop(self) -> "Graph":
        h = C.c_void_p()
        _check(self.lib.gaib_graph_add_selfloop(self.ctx.h, self.h, C.byref(h)), "gaib_graph_add_selfloop")
        return Graph(self.ctx, _handle=h)

    def reorder(self, method: int = ORDER_BFS):
        """(relabelled graph, new_of_old, old_of_new) -- gaib_graph_reorder: rows keep their edge order"""
        import torch

        h = C.c_void_p()
        new_of_old = torch.empty(self.nv, dtype=torch.int64, device=f"cuda:{self.ctx.device}")
        old_of_new = torch.empty_like(new_of_old)
        _check(self.lib.gaib_graph_reorder(self.ctx.h, self.h, method, C.byref(h), new_of_old.data_ptr(), old_of_new.data_ptr()),
               "gaib_graph_reorder")
        return Graph(self.ctx, _handle=h), new_of_old, old_of_new

    def sort_rows(self):
        """sort every row's column ids (a relabelled graph keeps its rows' edge order: GAT backward needs sorted rows)"""
        _check(self.lib.gaib_graph_sort_rows(self.ctx.h, self.h), "gaib_graph_sort_rows")

    def compute_vertex_data(self):
        _check(self.lib.gaib_graph_compute_vertex_data(self.ctx.h, self.h), "gaib_graph_compute_vertex_data")

    def compute_edge_data(self):
        _check(self.lib.gaib_graph_compute_edge_data(self.ctx.h, self.h), "gaib_graph_compute_edge_data")

    def set_vertex_norm(self, row_vdata, col_vdata, col_inv_deg, row_inv_deg=None):
        """rectangular (partitioned) graphs: normalisers come from the GLOBAL degrees."""
        _check(self.lib.gaib_graph_set_vertex_norm(self.ctx.h, self.h, _ptr(row_vdata), _ptr(row_inv_deg),
                                                   _ptr(col_vdata), _ptr(col_inv_deg)),
               "gaib_graph_set_vertex_norm")

    def _dev_tensor(self, ptr, n, dtype):
        """copy a device array owned by the graph into a fresh torch tensor"""
        import torch

        out = torch.empty(n, dtype=dtype, device=f"cuda:{self.ctx.device}")
        if n:
            _check(self.lib.gaib_memcpy_d2d(self.ctx.h, _ptr(out), ptr, out.numel() * out.element_size()),
                   "gaib_memcpy_d2d")
        self.ctx.sync()
        return out

    def rowptr(self):
        import torch
        return self._dev_tensor(self.lib.gaib_graph_rowptr(self.h), self.nv + 1, torch.int64)

    def colidx(self):
        import torch
        return self._dev_tensor(self.lib.gaib_graph_colidx(self.h), self.ne, torch.int32)

    def vertex_data(self):
        import torch
        return self._dev_tensor(self.lib.gaib_graph_vertex_data(self.h), self.nv, torch.float32)

    def edge_data(self):
        import torch
        return self._dev_tensor(self.lib.gaib_graph_edge_data(self.h), self.ne, torch.float32)

    def row_map(self):
        """[nv] int64 copy of the row map of a class graph (gaib_graph_split_classes), or None"""
        import torch
        p = self.lib.gaib_graph_row_map(self.h)
        if not p:
            return None
        return self._dev_tensor(p, self.nv, torch.int32).to(torch.int64)

    def device_bytes(self) -> int:
        return self.lib.gaib_graph_device_bytes(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gaib_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
