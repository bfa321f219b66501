"""Vertex-range partitioning + halo exchange for the GNN layer path on 1..8 MI355X.

One process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend); CPU tensors + "gloo" work
too (tests).  The reference has no multi-GPU GNN; the scheme restates
PartitionedGraph::edgecut_induced_partition1D (src/partitioner/graph_partition.cc:128-178) for
LearningGraph-style CSR: rank p owns a contiguous vertex range, its local CSR has the owned rows
and column ids over [owned | halo] vertices, with a local -> global id map.

Per aggregation (the path's one real exchange step):
  1. pack the owned rows other ranks list as halo          (gaib_gather_rows)
  2. all-to-all(v) of halo rows, every pair on its own xGMI link (all_to_all_single)
  3. local SpMM over [owned | halo]                          (gaib_spmm on the rectangular graph)
Weight gradients are summed with one all-reduce per layer (<= 64 KB); weights and Adam state are
replicated.  Degrees/normalisers of halo columns come from their owners (a halo vertex's local
degree is truncated).
"""
from __future__ import annotations

import os
import socket
import time
from dataclasses import dataclass, field

import torch
import torch.distributed as dist


def partition_bounds(n: int, world: int):
    per = -(-n // world)
    return [min(p * per, n) for p in range(world + 1)]


_A2A_OK = {}  # (backend, device type) -> every rank's all_to_all_single works (decided once, by all ranks together)


def _a2a_supported(sample: torch.Tensor, group=None) -> bool:
    """Does the backend have all_to_all_single for this kind of tensor?  Probed ONCE per (backend, device type) with a
    one-element exchange, and the answer is the MINIMUM over ranks, so every rank takes the same path afterwards --
    never a per-call, per-rank exception handler (ranks would end up in different collectives and hang)."""
    key = (dist.get_backend(group), sample.device.type)
    if key not in _A2A_OK:
        world = dist.get_world_size(group)
        ok = 1
        try:
            a = torch.zeros(world, dtype=torch.float32, device=sample.device)
            b = torch.empty_like(a)
            dist.all_to_all_single(b, a, group=group)
        except (RuntimeError, NotImplementedError):
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=sample.device if key[0] == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        _A2A_OK[key] = bool(int(flag.item()))
    return _A2A_OK[key]


def _all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_counts, in_counts, group=None):
    """all-to-all of row blocks; pairwise exchange where the backend lacks it (sends and receives posted together with
    batch_isend_irecv: on a one-stream NCCL communicator a send queued ahead of the matching receive would deadlock).
    Device tensors over a gloo group (2 processes sharing one GPU in the tests) go through host."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        o_h = torch.empty(out.shape, dtype=out.dtype)
        _all_to_all_rows(o_h, inp.cpu(), out_counts, in_counts, group)
        out.copy_(o_h)
        return
    if _a2a_supported(out, group):
        dist.all_to_all_single(out, inp, output_split_sizes=list(out_counts), input_split_sizes=list(in_counts),
                               group=group)
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    oo = [0]
    for c in out_counts:
        oo.append(oo[-1] + c)
    io = [0]
    for c in in_counts:
        io.append(io[-1] + c)
    out[oo[rank]:oo[rank + 1]] = inp[io[rank]:io[rank + 1]]
    ops, bufs = [], []
    for q in range(world):
        if q == rank:
            continue
        if in_counts[q]:
            ops.append(dist.P2POp(dist.isend, inp[io[q]:io[q + 1]].contiguous(), q, group))
        if out_counts[q]:
            buf = torch.empty_like(out[oo[q]:oo[q + 1]])
            bufs.append((q, buf))
            ops.append(dist.P2POp(dist.irecv, buf, q, group))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    for q, buf in bufs:
        out[oo[q]:oo[q + 1]] = buf


@dataclass
class Partition:
    """one rank's share of a vertex-range partitioned graph.  The rows' edges are split by column
    owner: `own` (columns in [0, n_own): this rank's vertices) and `halo` (columns in [0, n_halo):
    index into halo_gids), so the owned part can be aggregated while the halo rows are in flight."""
    rank: int
    world: int
    n_global: int
    lo: int
    hi: int
    rowptr_own: torch.Tensor    # int64 [n_own+1]
    colidx_own: torch.Tensor    # int32, local ids in [0, n_own)
    rowptr_halo: torch.Tensor   # int64 [n_own+1]
    colidx_halo: torch.Tensor   # int32, ids in [0, n_halo)
    degree: torch.Tensor        # int64 [n_own] full (global) degree of every owned row
    halo_gids: torch.Tensor     # int64 [n_halo] global ids, ascending (hence grouped by owner)
    recv_counts: list           # halo rows owned by rank q (contiguous segments of halo_gids)
    send_idx: torch.Tensor      # int64 [total_send] local row ids to ship, grouped by destination
    send_counts: list
    group: object = None

    @property
    def n_own(self) -> int:
        return self.hi - self.lo

    @property
    def n_halo(self) -> int:
        return int(self.halo_gids.numel())

    @property
    def ne(self) -> int:
        return int(self.colidx_own.numel() + self.colidx_halo.numel())


def complete_halo_threshold(world: int) -> float:
    """From which share of a peer's rows on the halo is taken whole (split_by_owner; the C++ builder computes the same figure,
    host/partition.cpp).  Taking a range whole puts the (1 - share) rows nobody reads on the wire, spread over world - 1 links,
    and saves the sender the pack of the share that is read (a read and a write of every row at ~5 TB/s of HBM): it pays while
    (1 - share) / ((world - 1) * link) < 2 * share / 5 TB/s, i.e. share > 1 - 2 (world - 1) link / 5000 GB/s, with the link rate
    the partition rule uses (GAIB_LINK_GBS: measured by the run where it can be, else 100).  N = 2: 0.96, N = 4: 0.88, N = 8:
    0.72 at 100 GB/s; never below 0.5.  GAIB_COMPLETE_HALO overrides (0 = never)."""
    ov = os.environ.get("GAIB_COMPLETE_HALO")
    if ov is not None and ov != "":
        return float(ov)
    if world < 2:
        return 0.0
    link = float(os.environ.get("GAIB_LINK_GBS", "100") or "100")
    return max(0.5, 1.0 - 2.0 * (world - 1) * link / 5000.0)


def split_by_owner(rowptr_local: torch.Tensor, colidx_global: torch.Tensor, lo: int, hi: int, bounds=None, complete: float = 0.0):
    """the local (no communication) half of the partition: rows [lo, hi) of the global CSR -> an owned-column CSR
    (column ids relative to lo), a halo-column CSR (column ids index `halo`), the sorted global ids of the halo
    vertices and the rows' full degrees.  owned + halo == the vertex set of the reference's induced subgraph
    (graph_partition.cc:150-166), pinned against it in tests/test_dist_cpu.py.
    bounds + complete > 0 (round 5): a peer range of which this rank needs at least that share of the rows is taken WHOLE --
    the halo grows by the few rows nobody here reads, and the peer's send list becomes its full row range, one run of
    consecutive rows, which the RCCL transport sends straight from the peer's matrix without packing (the N-way cut of a
    graph on a random numbering needs 99.9 % of every peer's rows at N = 8)."""
    device = colidx_global.device
    n_own = hi - lo
    assert rowptr_local.numel() == n_own + 1
    rowptr_local = rowptr_local.to(torch.int64)
    cols = colidx_global.to(torch.int64)
    deg = rowptr_local[1:] - rowptr_local[:-1]
    rows = torch.repeat_interleave(torch.arange(n_own, device=device), deg)
    own = (cols >= lo) & (cols < hi)
    halo = torch.unique(cols[~own])  # sorted
    if bounds is not None and complete > 0.0 and halo.numel():
        bt = torch.tensor(bounds, dtype=torch.int64, device=device)
        owner = torch.searchsorted(bt, halo, right=True) - 1
        cnt = torch.bincount(owner, minlength=len(bounds) - 1).tolist()
        segs, grew = [], False
        for q in range(len(bounds) - 1):
            size_q = bounds[q + 1] - bounds[q]
            if bounds[q] == lo and bounds[q + 1] == hi:
                continue  # this rank's own range
            if size_q > 0 and cnt[q] >= complete * size_q and cnt[q] < size_q:
                segs.append(torch.arange(bounds[q], bounds[q + 1], dtype=torch.int64, device=device))
                grew = True
            elif cnt[q]:
                segs.append(halo[owner == q])
        if grew:
            halo = torch.cat(segs)  # (still ascending: the ranges are)

    def csr_of(mask, ids):
        cnt = torch.bincount(rows[mask], minlength=n_own)
        rp = torch.zeros(n_own + 1, dtype=torch.int64, device=device)
        torch.cumsum(cnt, 0, out=rp[1:])
        return rp, ids.to(torch.int32).contiguous()  # edge order inside a row is preserved

    rp_own, ci_own = csr_of(own, cols[own] - lo)
    rp_halo, ci_halo = csr_of(~own, torch.searchsorted(halo, cols[~own]))
    return rp_own, ci_own, rp_halo, ci_halo, halo, deg


def build_partition(rowptr_local: torch.Tensor, colidx_global: torch.Tensor, n_global: int, rank: int, world: int,
                    group=None) -> Partition:
    """rowptr_local/colidx_global: this rank's rows [lo,hi) of the global CSR (global column ids),
    on the compute device.  Collective: every rank calls it."""
    device = colidx_global.device
    bounds = partition_bounds(n_global, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_own = hi - lo
    complete = complete_halo_threshold(world)  # see split_by_owner
    rp_own, ci_own, rp_halo, ci_halo, halo, deg = split_by_owner(rowptr_local, colidx_global, lo, hi, bounds, complete)
    # owner of each halo vertex -> how many rows we receive from each rank
    bt = torch.tensor(bounds, dtype=torch.int64, device=device)
    owner = torch.searchsorted(bt, halo, right=True) - 1
    recv_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    # tell every owner which of its rows we need
    send_counts = torch.empty_like(recv_counts)
    _all_to_all_rows(send_counts, recv_counts, [1] * world, [1] * world, group)
    rc, sc = [int(v) for v in recv_counts.tolist()], [int(v) for v in send_counts.tolist()]
    want = torch.empty(sum(sc), dtype=torch.int64, device=device)
    _all_to_all_rows(want, halo.contiguous(), sc, rc, group)  # ids requested FROM us, grouped by requester
    send_idx = want - lo
    assert send_idx.numel() == 0 or (int(send_idx.min()) >= 0 and int(send_idx.max()) < n_own)
    return Partition(rank, world, n_global, lo, hi, rp_own, ci_own, rp_halo, ci_halo, deg, halo, rc,
                     send_idx.contiguous(), sc, group)


class HaloExchanger:
    """moves the feature rows of halo vertices from their owners into a [n_halo x D] table.
    start() packs and launches the all-to-all (asynchronously on NCCL/RCCL), finish() waits."""

    def __init__(self, part: Partition, gather_rows=None):
        self.p = part
        # gather_rows(idx[int64 tensor], src (tensor or raw device pointer), D, out tensor)
        self.gather_rows = gather_rows or (lambda idx, src, D, out: torch.index_select(src, 0, idx, out=out))
        self.bytes_sent = 0
        self._buf = {}
        self._pending = None

    def _bufs(self, D, dtype, device):
        key = (D, dtype, str(device))
        if key not in self._buf:
            p = self.p
            self._buf[key] = (torch.empty(max(p.send_idx.numel(), 1), D, dtype=dtype, device=device),
                              torch.empty(max(p.n_halo, 1), D, dtype=dtype, device=device))
        s, r = self._buf[key]
        return s[:self.p.send_idx.numel()], r[:self.p.n_halo]

    def start(self, src, D: int, dtype=torch.float32, device=None):
        p = self.p
        device = device if device is not None else src.device
        sendbuf, recvbuf = self._bufs(D, dtype, device)
        if p.send_idx.numel():
            self.gather_rows(p.send_idx, src, D, sendbuf)
        self.bytes_sent += sendbuf.numel() * sendbuf.element_size()
        work = None
        if p.world > 1:
            # asynchronous all-to-all(v) on RCCL's stream where every rank has it (probed once, MIN over ranks);
            # otherwise the synchronous pairwise exchange: no overlap, still correct, the same path on every rank
            if sendbuf.is_cuda and dist.get_backend(p.group) != "gloo" and _a2a_supported(sendbuf, p.group):
                work = dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=list(p.recv_counts),
                                              input_split_sizes=list(p.send_counts), group=p.group, async_op=True)
            else:
                _all_to_all_rows(recvbuf, sendbuf, p.recv_counts, p.send_counts, p.group)
        self._pending = (work, recvbuf)

    def finish(self) -> torch.Tensor:
        work, recvbuf = self._pending
        self._pending = None
        if work is not None:
            work.wait()  # the compute stream now waits for the exchange
        return recvbuf

    def exchange(self, src, D: int, dtype=torch.float32, device=None) -> torch.Tensor:
        self.start(src, D, dtype, device)
        return self.finish()


class AbiHaloExchanger:
    """the same exchange behind the C ABI (gaib_halo_*: RCCL send/recv groups, or peer-to-peer pull over hipIpc) --
    torch.distributed is not on the data path.  Same interface as HaloExchanger; with it the C++ aggregators call
    gaib_halo_exchange_begin/end themselves (LearningGraph::set_halo_plan), no Python in the loop."""

    def __init__(self, ctx, comm, part: Partition):
        self.ctx, self.comm, self.p = ctx, comm, part
        self.halo = comm.halo(part.send_counts, part.send_idx, part.recv_counts)
        assert self.halo.rows == part.n_halo
        # the exchange in time slices where the ranges are large enough (gaib_halo_default_pieces: the same on every rank)
        self.halo.set_pieces(ctx.lib.gaib_halo_default_pieces(part.n_global, part.world))
        self._base = 0

    @property
    def bytes_sent(self) -> int:
        return self.halo.bytes_sent - self._base

    @bytes_sent.setter
    def bytes_sent(self, v):
        self._base = self.halo.bytes_sent - int(v)

    def exchange(self, src, D: int, dtype=torch.float32, device=None) -> torch.Tensor:
        """[n_halo x D] copy of the exchanged rows (set-up time use: normalisers, diagnostics)"""
        assert dtype == torch.float32
        self.halo.begin(src, D)
        ptr = self.halo.end()
        out = torch.empty(max(self.p.n_halo, 1), D, dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        if self.p.n_halo:
            from . import capi
            capi._check(self.ctx.lib.gaib_memcpy_d2d(self.ctx.h, out.data_ptr(), ptr, self.p.n_halo * D * 4),
                        "gaib_memcpy_d2d")
            self.ctx.sync()
        return out[:self.p.n_halo]

    def pack_only(self, src, D: int):
        """diagnostics: the pack kernel alone"""
        from . import capi
        if self.p.send_idx.numel():
            tmp = torch.empty(self.p.send_idx.numel(), D, dtype=torch.float32, device=f"cuda:{self.ctx.device}")
            capi._check(self.ctx.lib.gaib_gather_rows(self.ctx.h, self.p.send_idx.numel(), self.p.send_idx.data_ptr(), D,
                                                      src.data_ptr(), tmp.data_ptr()), "gaib_gather_rows")


def global_normalisers(part: Partition, ex: HaloExchanger):
    """deg^-1/2 and 1/deg from the GLOBAL degrees: (vd_own, inv_own, vd_halo, inv_halo).
    Owned rows are complete in a row partition, so their local degree is the global one; halo
    vertices' values come from their owners."""
    deg = part.degree.to(torch.float32)
    # deg^-1/2 with 0 for isolated vertices (lgraph.cpp:22-34); 1/deg in double then narrowed
    # (sage_aggregator.cpp:18).  float64 arithmetic reproduces both roundings.
    d64 = deg.to(torch.float64)
    s = torch.sqrt(deg).to(torch.float64)
    vd = torch.where(s == 0, torch.zeros_like(s), 1.0 / s).to(torch.float32).reshape(-1, 1).contiguous()
    inv = (1.0 / d64).to(torch.float32).reshape(-1, 1).contiguous()
    vd_h = ex.exchange(vd, 1).clone()
    inv_h = ex.exchange(inv, 1).clone()
    return vd[:, 0].contiguous(), inv[:, 0].contiguous(), vd_h[:, 0].contiguous(), inv_h[:, 0].contiguous()


# ---- GPU layer driver ----------------------------------------------------------------------------
class DistLayerGraph:
    """A LearningGraph over this rank's owned-column CSR plus a halo-column CSR
    (LearningGraph::set_halo): every aggregation packs + starts the halo all-to-all, sums the
    owned-column edges meanwhile, then adds the halo-column edges.  The C++ layer code is the
    single-GPU one."""

    def __init__(self, ctx, part: Partition, comm=None):
        """comm: a capi.Comm -> the exchange runs behind the C ABI; None -> torch.distributed (HaloExchanger)"""
        from . import capi, layers as L

        self.ctx, self.part, self.comm = ctx, part, comm
        self._capi = capi
        dev = f"cuda:{ctx.device}"

        def gather(idx, src, D, out):
            ptr = src if isinstance(src, int) else src.data_ptr()
            capi._check(ctx.lib.gaib_gather_rows(ctx.h, idx.numel(), idx.data_ptr(), D, ptr, out.data_ptr()),
                        "gaib_gather_rows")

        self.ex = AbiHaloExchanger(ctx, comm, part) if comm is not None else HaloExchanger(part, gather_rows=gather)
        vd, inv, vd_h, inv_h = global_normalisers(part, self.ex)
        g_own = ctx.graph(part.rowptr_own, part.colidx_own)
        g_own.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
        self.lgraph = L.LGraph.adopt(g_own)
        self.g_halo = None
        if part.world > 1:
            # every rank takes part in every exchange, also one without halo rows of its own
            nh = max(part.n_halo, 1)
            pad = lambda t: t if part.n_halo > 0 else torch.zeros(1, dtype=torch.float32, device=dev)
            self.g_halo = ctx.graph(part.rowptr_halo, part.colidx_halo, ncols=nh)
            self.g_halo.set_vertex_norm(vd, pad(vd_h), pad(inv_h), row_inv_deg=inv)
            if comm is not None:
                self.lgraph.set_halo_plan(self.g_halo, self.ex.halo)
            else:
                self.lgraph.set_halo(self.g_halo, self._begin, self._end)
                # (the partition-mode rule prices the exchange by the most rows one peer pair moves: the plan form knows)
                self.lgraph.set_halo_link_rows(max(max(part.send_counts), max(part.recv_counts)))
        self._dev = dev

    def _check_stream(self) -> None:
        """torch.distributed path only: the pack kernel and the halo SpMM run on the context's stream, the collective
        syncs against torch's CURRENT stream -- they must be the same stream (L.init binds the context to the stream
        that was current then), or the halo table / send buffer would be touched while still in flight."""
        want = getattr(self.ctx, "stream_ptr", None)
        if want is not None:
            cur = torch.cuda.current_stream().cuda_stream
            assert cur == want, (f"forward/backward called under torch stream {cur:#x}, the gaib context is bound to "
                                 f"{want:#x}: call L.init(device, stream) / gaib_ctx_set_stream with the stream you use")

    def _begin(self, length: int, d_in: int) -> None:
        self._check_stream()
        self.ex.start(d_in, length, torch.float32, self._dev)

    def _end(self, length: int) -> int:
        return self.ex.finish().data_ptr()

    def stats(self):
        """heavy-row split of both halves (roofline accounting)"""
        st = self.ctx.graph_stats(self.lgraph.device_graph())
        if self.g_halo is not None:
            sh = self.ctx.graph_stats(self.g_halo)
            st = {k: st[k] + sh[k] if k != "max_degree" else max(st[k], sh[k]) for k in st}
        return st


def allreduce_layer_grads(ctx, layer, which_list, shape, group=None, comm=None):
    """sum the weight gradients of one layer over ranks (one fused all-reduce)."""
    from . import capi

    n = shape[0] * shape[1]
    if comm is not None:  # behind the C ABI, in place on the layer's own gradient buffers
        for w in which_list:
            capi._check(ctx.lib.gaib_allreduce_f32(comm.h, layer.ptr(w), n), "gaib_allreduce_f32")
        return
    buf = torch.empty(len(which_list) * n, dtype=torch.float32, device=f"cuda:{ctx.device}")
    for i, w in enumerate(which_list):
        capi._check(ctx.lib.gaib_memcpy_d2d(ctx.h, buf[i * n:].data_ptr(), layer.ptr(w), n * 4), "gaib_memcpy_d2d")
    if dist.get_backend(group) == "gloo":  # tests: 2 processes on one GPU
        h = buf.cpu()
        dist.all_reduce(h, group=group)
        buf.copy_(h)
    else:
        dist.all_reduce(buf, group=group)
    for i, w in enumerate(which_list):
        capi._check(ctx.lib.gaib_memcpy_d2d(ctx.h, layer.ptr(w), buf[i * n:].data_ptr(), n * 4), "gaib_memcpy_d2d")


# what the record must say about the communicator's set-up (config.comm_init_timed_out): a transport whose set-up ran into its
# deadline on THIS rank is still inside a C call in a helper thread while the run goes on over the next transport -- the figure
# is then not a clean one, and the record names it
COMM_SETUP = {"timed_out": []}


def comm_attempt(ctx, rank: int, world: int, transport: int):
    """one communicator of one transport, created by all ranks together: every rank tries, the outcome is agreed on by all of
    them (all-reduce(MIN) of the success flag) -> (capi.Comm | None, error text).  Collective."""
    import threading

    from . import capi

    ok, comm, err = 1, None, ""
    try:
        uid = [capi.comm_unique_id(transport) if rank == 0 else None]
    except capi.GaibError as e:
        uid, ok, err = [None], 0, str(e)
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is None:
        ok = 0
    if ok:
        # ncclCommInitRank is collective and has no deadline of its own: if a peer fails before it (or inside it), this
        # rank would wait there for good and take the record with it.  The communicator is therefore created in a helper
        # thread with a deadline (GAIB_COMM_INIT_TIMEOUT_S, 90 s); a rank that runs into it reports failure -- the call is
        # left behind in its daemon thread, and a communicator that still comes out of it is closed at once -- and all ranks
        # move on to the next transport together; the record names the transport (config.comm_init_timed_out).  (The same
        # path creates the peer-to-peer communicator, so every N > 1 run exercises it.)
        box = {}
        hand_over = threading.Lock()  # the helper's "abandoned? else store" and the main thread's "abandon, close what is stored" are
                                      # each one step: a communicator that finishes exactly at the deadline is used or closed, never lost

        def create():
            try:
                c = capi.Comm(ctx, rank, world, uid[0], transport)
            except Exception as e:  # noqa: BLE001
                box["err"] = str(e)
                return
            with hand_over:
                if not box.get("abandoned"):
                    box["comm"] = c
                    return
            try:  # the deadline passed while the call was inside: nobody will use this communicator -- it goes at once instead of
                c.close()  # living on beside the one the run fell back to
            except Exception:  # noqa: BLE001
                pass

        limit = float(os.environ.get("GAIB_COMM_INIT_TIMEOUT_S", "90"))
        th = threading.Thread(target=create, daemon=True, name="gaib-comm-init")
        th.start()
        th.join(limit)
        with hand_over:  # (a communicator stored in the instant between join() and this lock counts as in time)
            if "comm" not in box and "err" not in box:
                box["abandoned"] = True
        if box.get("abandoned"):
            ok, err = 0, f"communicator set-up did not return within {limit:.0f} s (left behind in its thread)"
            COMM_SETUP["timed_out"].append("ipc" if transport == capi.COMM_IPC else "rccl")
        elif "comm" in box:
            comm = box["comm"]
        else:
            ok, err = 0, box.get("err", "communicator set-up failed")
    flag = torch.tensor([ok], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if comm is not None:
            comm.close()
        return None, err or "a peer failed"
    return comm, ""


def rccl_possible(ctx, world: int):
    """the cheap preconditions of ncclCommInitRank, agreed by all ranks BEFORE anybody enters it (the call is collective and
    has no deadline: a rank that cannot load RCCL, or two ranks on one device, would leave the others waiting inside it):
    librccl loads on every rank, and every rank has a device of its own -> (bool, reason).  Collective."""
    from . import capi

    mine = [ctx.device, 1 if capi.comm_transport_available(capi.COMM_RCCL) else 0]
    every = [None] * world
    dist.all_gather_object(every, mine)
    if not all(e[1] for e in every):
        return False, "librccl.so.1 does not load on rank(s) " + str([r for r, e in enumerate(every) if not e[1]])
    if len({e[0] for e in every}) != world:
        return False, f"ranks share devices ({[e[0] for e in every]}): RCCL wants one GPU per rank"
    return True, ""


def make_comm(ctx, rank: int, world: int, log):
    """The data-path communicator of bench.py's N > 1 leg, behind the C ABI.  GAIB_DIST_BACKEND:
         rccl (default)  gaib_comm over RCCL -- one GPU per rank
         ipc             gaib_comm over hipIpc peer-to-peer pull (also several ranks on one GPU: tests)
         nccl | gloo     no gaib_comm: torch.distributed moves the halo rows (the round-1 path)
    torch.distributed (the launcher's process group) only carries the 128-byte id, barriers and the timing reductions.
    The choice is made ONCE here, by all ranks together: if any rank fails to create its communicator, every rank
    falls back to the next transport, last to torch.distributed (an all-reduce(MIN) of the success flag) -- never per call."""
    from . import capi

    backend = os.environ.get("GAIB_DIST_BACKEND", "rccl")
    if backend in ("nccl", "gloo"):
        return None, f"torch.distributed/{dist.get_backend()}"
    # rccl (default): if it cannot be set up on some rank, the peer-to-peer pull transport (hipIpc handles, device-to-
    # device copies over xGMI) is tried before torch.distributed carries the rows
    order = [capi.COMM_IPC] if backend == "ipc" else [capi.COMM_RCCL, capi.COMM_IPC]
    failed = []
    for transport in order:
        name = "ipc" if transport == capi.COMM_IPC else "rccl"
        if transport == capi.COMM_RCCL:
            ok, err = rccl_possible(ctx, world)
            if not ok:
                log(f"[bench r{rank}] gaib_comm(rccl) not attempted: {err}")
                failed.append("rccl")
                continue
        comm, err = comm_attempt(ctx, rank, world, transport)
        if comm is not None:
            note = f" (after {', '.join(failed)} failed at set-up)" if failed else ""
            return comm, f"gaib_comm/{name}{note}"
        log(f"[bench r{rank}] gaib_comm({name}) unavailable on some rank: {err}")
        failed.append(name)
    log(f"[bench r{rank}] torch.distributed/{dist.get_backend()} carries the halo")
    return None, f"torch.distributed/{dist.get_backend()} (gaib_comm {', '.join(failed)} failed at set-up)"


class Budget:
    """wall-clock budget of an N > 1 bench run, counted from the rank's start.  The headline case always runs; a further
    sub-case starts only if EVERY rank still has its estimated time (one all-reduce(MIN) per decision: all ranks run the
    case or none does).  reduce_min(flag: int) -> int is the collective (torch.distributed in bench.py, a stub in tests)."""

    def __init__(self, total_s: float, t_start: float, reduce_min, clock=time.time):
        self.total_s, self.t_start, self.reduce_min, self.clock = float(total_s), float(t_start), reduce_min, clock

    def elapsed(self) -> float:
        return self.clock() - self.t_start

    def left(self) -> float:
        return self.total_s - self.elapsed()

    def agree(self, need_s: float) -> bool:
        return bool(self.reduce_min(1 if self.left() >= need_s else 0))

    def skipped(self, need_s: float) -> dict:
        return {"skipped": "budget", "elapsed_s": round(self.elapsed(), 1), "budget_s": self.total_s,
                "needed_s_estimate": round(need_s, 1)}


class BenchCase:
    """one partitioned GCN layer of bench.py's N > 1 leg: partition `rows`, build the graph (row classes by the rule) and the
    layer; time steps; diagnostics (one exchange on its own, the pack alone); A/B legs on the live objects; close.
    inputs(rank, nv) -> (x, grad_in) host arrays: what a comparison with the oracle will want the layer to run on."""

    def __init__(self, ctx, comm, args, rank, world, D, log, rows, label, inputs=None):
        from . import layers as L

        self.ctx, self.comm, self.args, self.rank, self.world, self.D, self.log, self.label = ctx, comm, args, rank, world, D, log, label
        self.L = L
        t0 = time.time()
        self.part = part = build_partition(rows.rowptr, rows.colidx_global, rows.n_global, rank, world)
        self.dg = dg = DistLayerGraph(ctx, part, comm)
        # decided (and the class graphs built) before the timed steps
        self.mode_used, self.n_bnd, self.bnd_edges = dg.lgraph.partition_mode(D)
        torch.cuda.synchronize()
        log(f"[bench r{rank}] {label}: rows [{part.lo},{part.hi}) ne={part.ne} (own-column {part.colidx_own.numel()}) "
            f"halo rows={part.n_halo} send rows={part.send_idx.numel()} mode {L.LGraph.PART_NAMES[self.mode_used]} "
            f"boundary rows={self.n_bnd} setup {time.time()-t0:.1f}s")
        self.nv = nv = part.n_own
        torch.manual_seed(43 + rank)
        self.layer = layer = L.Layer(L.GCN, 1, nv, D, D, dg.lgraph, act=True, lr=0.01)
        if inputs is not None:
            x_h, gin_h = inputs(rank, nv)
            layer.write(L.FEAT_IN, torch.from_numpy(x_h).cuda())
            layer.write(L.GRAD_IN, torch.from_numpy(gin_h).cuda())
            del x_h, gin_h
        else:
            layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda"))
            layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
        self.feat_out = torch.empty(nv, D, device="cuda")
        self.grad_out = torch.empty(nv, D, device="cuda")
        self.rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def step(self):
        L = self.L
        self.layer.forward(self.feat_out)
        self.layer.backward(self.feat_out, self.grad_out)
        allreduce_layer_grads(self.ctx, self.layer, [L.W_NEIGH_GRAD], (self.D, self.D), comm=self.comm)

    def time_steps(self, steps: int, warmup: int) -> float:
        """seconds of `steps` steps between barriers (this rank's clock; the caller takes the MAX over ranks).  Collective."""
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter() - t0

    def max_over_ranks(self, *vals):
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=self.rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def exchange_standalone_ms(self, ex=None, reps: int = 3) -> float:
        """one halo exchange of the layer's [nv x D] input on its own (pack + transfer + wait), this rank's clock.  Collective."""
        ex = ex or self.dg.ex
        feat = self.layer.tensor(self.L.FEAT_IN, (self.nv, self.D))
        ex.exchange(feat, self.D)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.exchange(feat, self.D)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def measure(self) -> dict:
        ctx, args, part, dg, L, D, nv = self.ctx, self.args, self.part, self.dg, self.L, self.D, self.nv
        for _ in range(args.warmup):
            self.step()
        torch.cuda.synchronize()
        ctx.prof_reset()
        ctx.prof_enable(True)
        dg.ex.bytes_sent = 0
        elapsed = self.time_steps(args.steps, 0)
        ctx.prof_enable(False)
        n_light, ms_light = ctx.prof_get("spmm_light")
        n_fused, ms_fused = ctx.prof_get("spmm_gemm_fused")
        n_heavy, ms_heavy = ctx.prof_get("spmm_heavy")
        n_gemm, ms_gemm = ctx.prof_get("sgemm")
        n_pack, ms_pack = ctx.prof_get("gather_rows")
        part_ms = {k: ctx.prof_get(k)[1] / args.steps for k in ("part_fused", "part_fused_acc", "part_fused_2t", "part_light",
                                                                "part_light_acc", "part_light_2t")}
        ctx.prof_reset()
        # diagnostics outside the timed region (collective: every rank runs them): one halo exchange of a [nv x D]
        # matrix on its own (pack + all-to-all + wait) and the pack alone -- what the owned-edge SpMM has to hide
        bytes_timed = dg.ex.bytes_sent
        exch_ms = self.exchange_standalone_ms()
        feat = self.layer.tensor(L.FEAT_IN, (nv, D))
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            if isinstance(dg.ex, AbiHaloExchanger):
                dg.ex.pack_only(feat, D)
            elif part.send_idx.numel():
                sendbuf, _ = dg.ex._bufs(D, torch.float32, feat.device)
                dg.ex.gather_rows(part.send_idx, feat, D, sendbuf)
        torch.cuda.synchronize()
        pack_ms = (time.perf_counter() - t0) / reps * 1e3
        # max time over ranks, total edges over ranks
        packed = True
        if isinstance(dg.ex, AbiHaloExchanger) and dg.ex.halo.send_stats()["packs"] == 0:
            # every peer of this plan is sent straight from the matrix (complete halos over RCCL): the timed steps ran no pack
            # kernel, so the stand-alone figure above is not a part of them
            pack_ms, packed = 0.0, False
        t = self.max_over_ranks(elapsed, exch_ms, pack_ms)
        e = torch.tensor([float(part.ne), float(part.n_halo), float(bytes_timed), float(part.colidx_halo.numel())],
                         dtype=torch.float64, device=self.rdev)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        # dominant kernel (rank 0's view) = the pass over the owned-column edges of each aggregation: with halo
        # edges that is spmm_w64_kernel (the halo half then carries the dense product), without them the fused
        # kernel does everything, as in the single-GPU bench
        mode_used, n_bnd, bnd_edges = self.mode_used, self.n_bnd, self.bnd_edges
        st_own = ctx.graph_stats(dg.lgraph.device_graph())
        e_light = part.colidx_own.numel() - st_own["heavy_edges"]
        if mode_used != L.LGraph.PART_SPLIT:
            # row classes: the dominant kernel is the fused pass over the larger class -- the interior rows (one table) or the
            # boundary rows over [owned | halo] (two tables); bytes as in the single-GPU record, per edge of that class
            by_int = part_ms["part_fused"] >= max(part_ms["part_fused_2t"], part_ms["part_light"])
            e_int = part.ne - bnd_edges
            if by_int:
                kernel_name = "spmm_gemm_kernel<VEC=2,edge-weights,U=16,PART> over the interior rows (aggregation + MFMA product, rank 0)"
                e_k, r_k, ms_dom = e_int, nv - n_bnd, part_ms["part_fused"] * args.steps
            elif part_ms["part_fused_2t"] > 0:
                kernel_name = ("spmm_gemm_kernel<VEC=2,edge-weights,U=16,PART> over the boundary rows, one pass over "
                               "[owned | halo] (two feature tables, rank 0)")
                e_k, r_k, ms_dom = bnd_edges, n_bnd, part_ms["part_fused_2t"] * args.steps
            else:
                kernel_name = "spmm_w64_kernel<VEC=2,PART> over the boundary rows' owned-column edges (rank 0)"
                e_k, r_k, ms_dom = bnd_edges - part.colidx_halo.numel(), n_bnd, part_ms["part_light"] * args.steps
            alg_bytes = e_k * (4 * D + 8) + int(1.5 * r_k * 4 * D) + (r_k + 1) * 8
            n_dom = 2 * args.steps
        elif part.colidx_halo.numel() > 0 and dg.g_halo is not None and dg.lgraph.halo_pieces(D) > 1 and ms_light >= ms_fused:
            # round 6, the column split with the halo-column half consumed in K' pieces: the row kernel runs the owned-column pass
            # AND the K' - 1 accumulate passes over the pieces that are not the last (the last one is the fused kernel with the
            # product): K' launches per aggregation under one key.  Bytes of an average launch: the owned-column gathers and
            # stores, the pieces' gathers ((K' - 1) / K' of the halo-column edges), one read + write of the rows per piece pass
            kc = dg.lgraph.halo_pieces(D)
            st_h = ctx.graph_stats(dg.g_halo)
            e_h = part.colidx_halo.numel() - st_h["heavy_edges"]
            kernel_name = (f"spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer>: the owned-column pass and the {kc - 1} accumulate "
                           f"pass(es) over the halo-column pieces that are not the last (rank 0; {kc} launches per aggregation, averaged)")
            total = (e_light * (4 * D + 8) + (nv - st_own["n_heavy"]) * 4 * D + (nv + 1) * 8
                     + (kc - 1) / kc * e_h * (4 * D + 8) + (kc - 1) * (2 * nv * 4 * D + (nv + 1) * 8))
            alg_bytes = int(total / kc)
            n_dom, ms_dom = n_light, ms_light
        elif part.colidx_halo.numel() > 0 and ms_fused > ms_light and dg.g_halo is not None:
            # the column split on a partition whose edges mostly cross ranges (the strong case on a random order at N >= 4):
            # the halo-column half -- the fused kernel continuing the owned-column partial sums, the dense product riding on
            # it -- is the larger launch.  Bytes: its gathers, the partial sums read, the rows it stores (A.X forward; 1.5 on average)
            st_h = ctx.graph_stats(dg.g_halo)
            e_h = part.colidx_halo.numel() - st_h["heavy_edges"]
            kernel_name = ("spmm_gemm_kernel<VEC=2,edge-weights,U=16,buffer> over the halo-column edges (continues the owned-column "
                           "sums, aggregation + MFMA dense product, rank 0)")
            alg_bytes = e_h * (4 * D + 8) + nv * 4 * D + int(1.5 * nv * 4 * D) + (nv + 1) * 8
            n_dom, ms_dom = n_fused, ms_fused
        elif part.colidx_halo.numel() > 0:
            kernel_name = "spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer> over the owned-column edges (rank 0)"
            alg_bytes = e_light * (4 * D + 8) + (nv - st_own["n_heavy"]) * 4 * D + (nv + 1) * 8
            n_dom, ms_dom = n_light, ms_light
        else:
            kernel_name = "spmm_gemm_kernel<VEC=2,edge-weights,U=16,buffer> (aggregation + MFMA dense product, rank 0)"
            alg_bytes = e_light * (4 * D + 8) + int(1.5 * nv * 4 * D) + (nv + 1) * 8
            n_dom, ms_dom = n_fused, ms_fused
        avg_ms = ms_dom / max(n_dom, 1)
        link_env = os.environ.get("GAIB_LINK_GBS")
        # how rank 0's rows left: exchanges that packed, sends straight from the matrix (complete halos over RCCL), and the share
        # of a peer's rows from which its range is taken whole on this run
        send_stats = dict(dg.ex.halo.send_stats(), complete_halo_from_share=round(complete_halo_threshold(self.world), 3)) \
            if isinstance(dg.ex, AbiHaloExchanger) else None
        return dict(elapsed=t[0], exch_ms=t[1], pack_ms=t[2], total_edges=float(e[0]),
                    halo_rows_total=int(e[1]), halo_bytes_per_step_total=float(e[2]) / args.steps, nv=nv,
                    cut_fraction_measured=float(e[3]) / max(float(e[0]), 1.0),
                    owned_edge_spmm_ms_per_step=ms_light / args.steps, kernel_name=kernel_name, alg_bytes=alg_bytes,
                    halo_send_stats=send_stats, halo_pack_in_timed_steps=packed,
                    # time slices the exchanges of the timed steps travelled in / the halo-column half was consumed in
                    halo_pieces=dict(plan=dg.ex.halo.pieces, consumed=dg.lgraph.halo_pieces(D)) if isinstance(dg.ex, AbiHaloExchanger) else None,
                    avg_ms=avg_ms, launches=n_dom, parity=None,
                    # rank 0's kernels per step: the owned-column pass, the halo-column half (fused with the dense product),
                    # heavy rows, the weight gradient, the pack of the rows on the send lists
                    breakdown=dict(owned_edge_spmm_ms=ms_light / args.steps, halo_half_ms=ms_fused / args.steps,
                                   heavy_rows_ms=ms_heavy / args.steps, sgemm_ms=ms_gemm / args.steps,
                                   pack_ms=ms_pack / args.steps, **{k + "_ms": v for k, v in part_ms.items() if v}),
                    # how rank 0 aggregates on this partition (LearningGraph::partition_mode) and what decided it
                    partition_mode=dict(mode=L.LGraph.PART_NAMES[mode_used], boundary_rows=n_bnd, boundary_row_share=n_bnd / max(nv, 1),
                                        boundary_edges=bnd_edges, link_gbs_assumed=float(link_env or "100"),
                                        link_gbs_source=LINK_GBS_SOURCE.get("source", "the library's default (100 GB/s per peer pair)")),
                    value=2 * float(e[0]) * args.steps / t[0], ms_per_step=t[0] / args.steps * 1e3)

    def cu_reserve_ab(self, reserves=(0, 32, 64), steps: int = 6) -> dict:
        """the step with the persistent fused aggregation leaving 0 / 32 / 64 CUs to the transport while an exchange is in
        flight (option comm_reserve_cus; GAIB_OVERLAPS_TRANSFER).  The option only reaches launches of the fused kernel that
        overlap an exchange: the interior pass of the class modes -- in the column split the overlapping pass is the
        non-persistent spmm_w64_kernel, whose workgroups retire and leave room by themselves (the record says which mode ran).
        Collective; max over ranks; the option returns to what it was."""
        ctx = self.ctx
        raw = ctx.get_option("comm_reserve_cus_raw")
        out = {"mode": self.L.LGraph.PART_NAMES[self.mode_used], "steps_each": steps, "ms_per_step": {},
               "fused_launches_overlap_an_exchange": self.mode_used != self.L.LGraph.PART_SPLIT}
        try:
            for r in reserves:
                ctx.set_option("comm_reserve_cus", r)
                el = self.time_steps(steps, 1)
                out["ms_per_step"][str(r)] = self.max_over_ranks(el)[0] / steps * 1e3
        finally:
            ctx.set_option("comm_reserve_cus", raw)
        out["in_effect_for_the_timed_steps"] = ctx.get_option("comm_reserve_cus")
        return out

    def halo_pipeline_ab(self, pieces=(1, 2, 4, 8), steps: int = 6) -> dict:
        """the step with the exchange travelling in 1 / 2 / 4 / 8 time slices (gaib_halo_set_pieces on the live plan, the same
        on every rank): one slice = the whole exchange awaited before the halo-column half, K slices = that half aggregated
        piece by piece as they land.  Per K: ms per step (max over ranks) and the outputs' distance from
        the one-slice run's (inf norm over this rank's rows, max over ranks: the same terms added piece-major -- 0 on rows
        below the heavy threshold where every rank has ONE peer).  Collective; the plan returns to what it was."""
        L, plan = self.L, getattr(self.dg.ex, "halo", None)
        out = {"mode": L.LGraph.PART_NAMES[self.mode_used], "steps_each": steps, "ms_per_step": {}, "inf_vs_one_slice": {},
               "slices_consumed": {}}
        if plan is None:
            return {"skipped": "torch.distributed carries the rows: no plan to cut into slices"}
        if self.mode_used not in (L.LGraph.PART_SPLIT, L.LGraph.PART_CLASSES):
            return {"skipped": f"mode {out['mode']}: no halo-column half to consume piece by piece", "mode": out["mode"]}
        k0 = plan.pieces
        ref = None
        try:
            for k in pieces:
                plan.set_pieces(k)
                self.dg.lgraph.set_halo_consumption(k)  # (forced: k slices on the wire, consumed in k pieces)
                el = self.time_steps(steps, 2)  # (the first step after a change cuts the piece graphs)
                out["ms_per_step"][str(k)] = self.max_over_ranks(el)[0] / steps * 1e3
                out["slices_consumed"][str(k)] = self.dg.lgraph.halo_pieces(self.D)
                cur = (self.feat_out.clone(), self.grad_out.clone())
                if ref is None:
                    ref = cur
                else:
                    d = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(cur, ref))
                    out["inf_vs_one_slice"][str(k)] = self.max_over_ranks(d)[0]
        finally:
            plan.set_pieces(k0)
            self.dg.lgraph.set_halo_consumption(-1)  # back to the rule
        # what the timed steps of the record ran with: k0 slices on the wire, consumed in as many pieces as this rank's rule chose
        out["in_effect_for_the_timed_steps"] = {"plan": k0, "consumed_rank0": self.dg.lgraph.halo_pieces(self.D)}
        return out

    def transport_ab(self, transports) -> dict:
        """the SAME exchange plan (this case's send / receive lists) timed on its own over every transport in `transports`:
        {name: capi.Comm | (None, reason)} -> {name: ms (max over ranks) | {"skipped": reason}}.  Collective."""
        out = {}
        for name, c in transports.items():
            if not hasattr(c, "halo"):
                out[name] = {"skipped": c[1]}
                continue
            plan_ex = None
            try:
                if c is self.comm:
                    ms = self.exchange_standalone_ms()
                else:
                    plan_ex = AbiHaloExchanger(self.ctx, c, self.part)
                    ms = self.exchange_standalone_ms(plan_ex)
                out[name] = {"exchange_standalone_ms": self.max_over_ranks(ms)[0]}
            finally:
                if plan_ex is not None:
                    torch.cuda.synchronize()
                    plan_ex.halo.close()
        return out

    def close(self):
        # the case's objects go in an order: the layer, the graph (which points at the plan), then the plan -- collectively (every
        # rank is here): its send buffer and halo table return to the communicator's pool for the next case, and a default run
        # never comes near the communicator's limit of plans alive at once
        plan = getattr(self.dg.ex, "halo", None)
        self.layer.close()
        self.dg.lgraph.close()
        self.layer = self.feat_out = self.grad_out = self.dg = None
        torch.cuda.synchronize()
        if plan is not None:
            plan.close()
        torch.cuda.empty_cache()


# where the link rate of the partition-mode rule came from (set once per run by bench_gcn_layer; the record names it)
LINK_GBS_SOURCE = {}


def _bench_case(ctx, comm, args, rank, world, D, log, rows, label, check=None, on_measured=None, extras=None):
    """one timed case.  check(part, layer, feat_out, grad_out): bench.py's comparison with the oracle's GLOBAL run (test
    infrastructure stays outside this package); the layer then runs on check.inputs(rank, nv) -> (x, grad_in) host arrays.
    on_measured(res): called as soon as the timed steps and the diagnostics are in -- BEFORE the comparison and the A/B legs
    (the caller holds the record from there on).  extras(case, res): further legs on the live objects (A/B), before close."""
    case = BenchCase(ctx, comm, args, rank, world, D, log, rows, label, inputs=check.inputs if check is not None else None)
    res = case.measure()
    if on_measured is not None:
        on_measured(res)
    if check is not None:
        if hasattr(check, "exchanger"):
            check.exchanger = case.dg.ex  # (diagnostics of a failing comparison: the plan the layer itself uses)
        res["parity"] = check(case.part, case.layer, case.feat_out, case.grad_out)
    if extras is not None:
        extras(case, res)
    case.close()
    return res


def bench_gcn_layer(ctx, args, rank: int, world: int, D: int, log, make_check=None, t_start=None, hold=None, cpu_leg=None,
                    parity_check=None, traffic_of=None, strong_check=None):
    """bench.py's N > 1 leg.  GCN hidden layer D -> D forward + backward per step, halo exchange before each of the 2
    SpMM, one all-reduce of dW per step.

    --scaling strong (default since round 5): north_star's curve -- the SAME 2.45 M-vertex products-shaped graph of the
    single-GPU bench, partitioned N ways by vertex range (its vertex order is random, so the cut is (N-1)/N), the reference's
    scheme for one fixed graph on N devices (src/partitioner/graph_partition.cc:128-178, src/triangle/multigpu_induced.cu:31-84).
    That case is `value`; its sub-record `config.strong_products` adds the one-rank timing of the same graph taken in the run
    (`speedup_vs_n1`).  The weak case -- every rank owns a products-shaped vertex range of one global block Chung-Lu graph
    (synth.block_rows), cut 0.1 -- runs second as `config.weak_products_range`; the share of a range's edges that cross ranges
    stands for the partitioner's quality, so the other end, (N-1)/N, is `config.random_order`, and the clustered-boundary
    generator `config.clustered_boundary`.
    --scaling weak: the two swap places (`value` = the weak case at --cut-fraction, `config.strong_products` second).
    --workload gcn-papers (BASELINE config 5): the same layer on the ogbn-papers100M-shaped graph, one vertex range of
    1/8 of it per rank -- at N = 8 the whole 111 M-vertex / 3.2 G-edge graph -- at both ends of the partition-quality
    axis (weak by construction).  make_check(shape, cut) -> bench.py's oracle comparison for one case (--check-oracle), or None.
    The record cannot be lost: the HEADLINE case runs first and rank 0 hands the record to hold() as soon as it is measured
    (bench.py prints it if anything ends the run early); every further leg -- the comparison with the oracle, the other
    scaling mode, the A/B legs of the run's own constants (config.cu_reserve_ab, config.transport_ab), the clustered-boundary
    generator, the random vertex order, config 5 -- starts only if all ranks agree that its estimated time fits args.budget_s
    counted from t_start (Budget), else its slot says {"skipped": "budget", ...}.
    strong_check(comm, bounds) -> bench.py's comparison of the strong case with the oracle's run on the bench graph (the run
    the N = 1 bench makes: also the record's cpu_baseline).  parity_check(shape, cut, comm, scale) -> the comparison at another
    scale for the weak generator: the same partitioned layer over the same transport on a graph whose GLOBAL size the oracle
    runs in seconds, every rank's rows element-wise against the oracle's global run."""
    import socket

    from . import capi, layers as L, synth

    comm, transport = make_comm(ctx, rank, world, log)
    rdev0 = "cuda" if dist.get_backend() == "nccl" else "cpu"
    every = [None] * world
    dist.all_gather_object(every, [list(COMM_SETUP["timed_out"]), socket.gethostname(), ctx.device])
    comm_timed_out = sorted({t for e in every for t in e[0]})
    share = len({(e[1], e[2]) for e in every}) < world  # two ranks on one device: timings are not evidence of anything

    def reduce_min(flag: int) -> int:
        t = torch.tensor([flag], dtype=torch.int32, device=rdev0)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    budget = Budget(getattr(args, "budget_s", 1e9), t_start if t_start is not None else time.time(), reduce_min)
    papers = getattr(args, "workload", "gcn-products") == "gcn-papers"
    shape = "ogbn-papers100M/8" if papers else "ogbn-products"
    mk = (lambda c: make_check(shape, c, comm)) if make_check is not None else (lambda c: None)
    # ---- the xGMI link, measured FIRST (rank 0 while the others wait): the figure replaces the spec constant in the record and
    # -- before any partitioned graph exists -- the constant in the partition-mode rule (LearningGraph::partition_mode prices the
    # exchange per peer pair: GAIB_LINK_GBS, read when a graph's mode is decided)
    ndev = torch.cuda.device_count()
    link = {"skipped": f"{ndev} visible device(s): no peer to copy to"}
    if world > 1 and share:
        link = {"skipped": "ranks share a device: a peer copy would not cross a link"}
    elif rank == 0 and ndev >= 2:
        try:
            t0 = time.time()
            link = {"unidirectional_gbs": capi.probe_peer_copy(0, 1, 1 << 28, 10, False),
                    "bidirectional_gbs_per_direction": capi.probe_peer_copy(0, 1, 1 << 28, 10, True), "bytes": 1 << 28,
                    "pair": [0, 1]}
            # every peer of device 0, one direction (the 7 links of a node are not promised to be alike), bounded to ~10 s
            per_peer = {}
            for d in range(2, min(ndev, 8)):
                if time.time() - t0 > 10.0:
                    break
                per_peer[str(d)] = capi.probe_peer_copy(0, d, 1 << 27, 5, False)
            if per_peer:
                link["unidirectional_gbs_from_0_to"] = {"1": link["unidirectional_gbs"], **per_peer}
            link["seconds"] = round(time.time() - t0, 2)
        except capi.GaibError as e:
            link = {"error": str(e)[:200]}
    got = [link.get("unidirectional_gbs") if rank == 0 else None]
    dist.broadcast_object_list(got, src=0)
    if "GAIB_LINK_GBS" in os.environ:
        LINK_GBS_SOURCE["source"] = "GAIB_LINK_GBS from the environment"
    elif got[0]:
        # (the slowest link seen decides: the rule prices the pair that moves the most rows)
        slow = min([got[0]] + (list(link.get("unidirectional_gbs_from_0_to", {}).values()) if rank == 0 else []))
        slow_b = [slow]
        dist.broadcast_object_list(slow_b, src=0)
        os.environ["GAIB_LINK_GBS"] = f"{float(slow_b[0]):.1f}"
        LINK_GBS_SOURCE["source"] = "config.xgmi_link_probe of this run (slowest link from device 0, one direction), set before the first partition was built"
    else:
        LINK_GBS_SOURCE["source"] = "the library's default (100 GB/s per peer pair): no link was probed in this run"
    dist.barrier()
    strong = getattr(args, "scaling", "strong") == "strong" and not papers
    cut = 0.1 if args.cut_fraction is None else args.cut_fraction
    default_run = args.cut_fraction is None and world > 1 and not papers
    extra = clustered = config5 = cpu_rec = parity_scaled = None
    other = None          # the sub-record of the scaling mode that is not the headline
    strong_extra = {}     # one-rank timing of the same graph, speedup (strong case)
    ab = {"cu_reserve_ab": None, "transport_ab": None, "halo_pipeline_ab": None}
    state = {"main": None, "headline_s": 0.0}

    def sub_record(r, **more):
        return {"value": r["value"], "ms_per_step": r["ms_per_step"], "halo_rows_total": r["halo_rows_total"],
                "halo_bytes_per_step_total": r["halo_bytes_per_step_total"], "halo_exchange_standalone_ms": r["exch_ms"],
                "halo_pack_ms": r["pack_ms"], "owned_edge_spmm_ms_per_step": r["owned_edge_spmm_ms_per_step"],
                "cut_fraction_measured": r["cut_fraction_measured"], "halo_send_stats_rank0": r["halo_send_stats"],
                "halo_pieces_rank0": r["halo_pieces"],
                "breakdown_ms_per_step_rank0": r["breakdown"], "partition_mode_rank0": r["partition_mode"], **more}

    strong_workload = (f"the single-GPU bench's ogbn-products-shaped graph (seed 42, random vertex order) partitioned into "
                       f"{world} vertex ranges, GCN hidden layer 128->128 fwd+bwd, halo exchange before each SpMM + dW all-reduce")
    if papers:
        weak_workload = (f"BASELINE config 5: block Chung-Lu graph of {world} vertex range(s), each 1/8 of the ogbn-papers100M "
                         f"shape (seed 42; N = 8: 111 M vertices, 3.2 G edges), GCN hidden layer 128->128 fwd+bwd, halo "
                         f"exchange before each SpMM + dW all-reduce")
    else:
        weak_workload = ("block Chung-Lu graph, one ogbn-products-shaped vertex range per GPU (seed 42), "
                         "GCN hidden layer 128->128 fwd+bwd, halo exchange before each SpMM + dW all-reduce")

    def strong_rows(one_rank=True):
        """this rank's rows of the single-GPU bench graph (every rank generates the whole graph: seeded, identical) and, on
        rank 0, the one-rank timing of the same layer on the whole graph (the denominator of speedup_vs_n1) while the full
        CSR is on the device anyway"""
        sg = synth.make("ogbn-products", seed=42, device="cuda", scale=args.scale)
        g0 = ctx.graph(sg.rowptr, sg.colidx)
        g1 = g0.add_selfloop()  # GCN aggregates over A + I (net.cpp:96)
        g0.close()
        del sg
        rp_all, ci_all = g1.rowptr(), g1.colidx()  # (copies)
        n_global = g1.nv
        b = partition_bounds(n_global, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = int(rp_all[lo]), int(rp_all[hi])
        rows = synth.BlockRows((rp_all[lo:hi + 1] - e0).contiguous(), ci_all[e0:e1].to(torch.int64).contiguous(), n_global, hi - lo)
        del rp_all, ci_all
        torch.cuda.empty_cache()
        one = None
        if one_rank and rank == 0 and os.environ.get("GAIB_BENCH_ONE_RANK", "1") != "0":
            try:
                one = _one_rank_timing(ctx, g1, args, D)  # (takes the graph over and closes it)
            except Exception as e:  # noqa: BLE001 -- a side measurement must not cost the headline record
                one = {"error": f"{type(e).__name__}: {e}"[:200]}
            torch.cuda.empty_cache()
        else:
            g1.close()
        dist.barrier()  # (the others wait here for rank 0's one-rank timing: the cases start together)
        return rows, b, one

    ab_on = os.environ.get("GAIB_BENCH_AB", "1") != "0" and default_run
    ab_force = os.environ.get("GAIB_BENCH_AB", "") == "force"  # rehearsals / tests: run the bookkeeping on a shared device too

    def cu_reserve_leg(case, res):
        """the run measures its own constants, (b): the CUs left to the transport, on the LIVE headline case --
        the same communicator and plan, only the option changes -- budgeted, {"skipped": reason} where it would not measure anything"""
        if not ab_on:
            return
        step_s = max(res["ms_per_step"] * 1e-3, 1e-3)
        need = min(20.0, 3 * 8 * step_s + 3)
        if share and not ab_force:
            ab["cu_reserve_ab"] = {"skipped": "ranks share a device: the transport and the fused kernel would compete for the same CUs whatever is reserved"}
        elif comm is None:
            ab["cu_reserve_ab"] = {"skipped": "torch.distributed carries the rows: no GAIB_OVERLAPS_TRANSFER launches"}
        elif not budget.agree(need):
            ab["cu_reserve_ab"] = budget.skipped(need)
        else:
            ab["cu_reserve_ab"] = case.cu_reserve_ab(steps=max(3, min(8, int(5.0 / step_s))))
            if share:
                ab["cu_reserve_ab"]["ranks_share_device"] = True
        if hold is not None and rank == 0:
            hold(assemble())

    def halo_pipeline_leg(case, res):
        """(a'): the exchange in 1 / 2 / 4 / 8 time slices on the LIVE headline case (round 6): whether consuming the halo-column
        half piece by piece hides the wire, measured where there is a wire"""
        if not ab_on:
            return
        step_s = max(res["ms_per_step"] * 1e-3, 1e-3)
        n_steps = max(3, min(8, int(4.0 / step_s)))
        need = min(30.0, 4 * (n_steps + 2) * step_s * 1.5 + 4)
        if share and not ab_force:
            ab["halo_pipeline_ab"] = {"skipped": "ranks share a device: no wire to hide, the slices' kernels only take turns"}
        elif comm is None:
            ab["halo_pipeline_ab"] = {"skipped": "torch.distributed carries the rows: no plan to cut into slices"}
        elif not budget.agree(need):
            ab["halo_pipeline_ab"] = budget.skipped(need)
        else:
            ab["halo_pipeline_ab"] = case.halo_pipeline_ab(steps=n_steps)
            if share:
                ab["halo_pipeline_ab"]["ranks_share_device"] = True
        if hold is not None and rank == 0:
            hold(assemble())

    def headline_legs(case, res):
        halo_pipeline_leg(case, res)
        cu_reserve_leg(case, res)

    def transport_leg():
        """(c): the strong case's exchange plan timed on its own over RCCL and over the peer-to-peer pull.  The LAST leg of the
        run: it creates a second communicator, the one step of this file that has never met real hardware with N > 1 -- if
        it fails on some rank only, the ranks part ways and the deadline ends the run; every other figure is held by then."""
        if not ab_on:
            return
        need = 25.0
        if comm is None:
            ab["transport_ab"] = {"skipped": "torch.distributed carries the rows: no gaib_comm plan to re-time"}
            return
        if not budget.agree(need):
            ab["transport_ab"] = budget.skipped(need)
            return
        mine = "rccl" if transport.startswith("gaib_comm/rccl") else "ipc"
        tr = {mine: comm}
        made = None
        other_t, other_name = (capi.COMM_IPC, "ipc") if mine == "rccl" else (capi.COMM_RCCL, "rccl")
        if other_name == "rccl":
            ok, why = rccl_possible(ctx, world)
            if not ok:
                tr["rccl"] = (None, why)
            elif "rccl" in transport:  # "gaib_comm/ipc (after rccl failed at set-up)": do not walk into the same failure twice
                tr["rccl"] = (None, "RCCL failed at set-up in this run")
        if other_name not in tr:
            made, err = comm_attempt(ctx, rank, world, other_t)
            tr[other_name] = made if made is not None else (None, f"set-up failed: {err}"[:200])
        rows, _, _ = strong_rows(one_rank=False)
        a2 = type(args)(**{**vars(args), "steps": 1, "warmup": 0})
        case = BenchCase(ctx, comm, a2, rank, world, D, log, rows, "transport A/B: the strong case's plan")
        del rows
        try:
            ab["transport_ab"] = {"plan": "the strong case's send / receive lists, [nv x 128] fp32, pack + transfer + wait",
                                  "carried_the_run": mine, **case.transport_ab(tr)}
            if share:
                ab["transport_ab"]["ranks_share_device"] = True
        finally:
            case.close()
            if made is not None:
                torch.cuda.synchronize()
                made.close()

    def run_strong(on_measured, check, extras=None):
        rows, bounds, one = strong_rows()
        r = _bench_case(ctx, comm, args, rank, world, D, log, rows, "strong scaling, products graph",
                        check=check(bounds) if check is not None else None, on_measured=on_measured, extras=extras)
        if one is not None:
            strong_extra["one_rank_same_graph"] = one
            if "value" in one:
                strong_extra["speedup_vs_n1"] = r["value"] / one["value"]
                strong_extra["speedup_vs_n1_of"] = ("this case's value over the one-rank timing of the same layer on the whole graph, "
                                                    "taken on rank 0's device in this run before the graph was partitioned")
        return r

    def run_weak(on_measured, check, extras=None):
        rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=args.scale,
                                selfloops=True)  # GCN aggregates over A + I (net.cpp:96)
        return _bench_case(ctx, comm, args, rank, world, D, log, rows, f"cut {cut:.3f}", check=check, on_measured=on_measured,
                           extras=extras)

    def assemble():
        main = state["main"]
        achieved = main["alg_bytes"] / (main["avg_ms"] * 1e-3) / 1e9 if main["avg_ms"] > 0 else 0.0
        rccl_ranks = comm.size if (comm is not None and transport.startswith("gaib_comm/rccl")) else 0
        parity = main["parity"] if main["parity"] is not None else (None if strong else parity_scaled)
        if strong and parity_scaled is not None:
            parity = {**(parity or {"ok": None, "reason": "the comparison of the strong case did not run"}),
                      "weak_generator_scaled": parity_scaled}
            if parity_scaled.get("ok") is False:
                parity["ok"] = False
        if isinstance(extra, dict) and "value" in extra:
            rp = extra.get("parity")
            if parity is not None and rp is not None:
                parity = {**parity, "random_order": rp, "ok": (None if parity.get("ok") is None else bool(parity["ok"] and rp.get("ok")))}
        rec = {"parity": parity} if parity is not None else {}
        # L2 -> fabric bytes per launch of the dominant kernel, where a PMC pass of exactly this kernel on exactly this shard
        # exists (bench.py: profiles/hbm_traffic.json, verified by source hashes): the owned-column pass of the split on the
        # products-shaped range at cut 0.1, uniform generator -- the same 113.6 M own-column edges at every N
        traffic, traffic_src = None, "no PMC pass of this kernel on this shard"
        if (traffic_of is not None and not strong and not papers and args.scale == 1.0 and abs(cut - 0.1) < 1e-9
                and main["partition_mode"]["mode"] == "split" and main["kernel_name"].startswith("spmm_w64_kernel")):
            traffic, traffic_src = traffic_of("spmm_w64_kernel_owned_pass_bytes_per_launch", "partitioned_products_uniform")
        # ... and of the strong headline's dominant kernel at N = 8 (the halo-column half on rank 0's eighth of the bench graph)
        if (traffic_of is not None and strong and world == 8 and args.scale == 1.0 and main["partition_mode"]["mode"] == "split"
                and "halo-column edges" in main["kernel_name"]):
            traffic, traffic_src = traffic_of("spmm_gemm_kernel_halo_half_bytes_per_launch", "partitioned_products_strong")
        strong_rec = ({**sub_record(main, workload=strong_workload), **strong_extra} if strong else other)
        weak_rec = (other if strong else sub_record(main, workload=weak_workload, cut_fraction=cut))
        rec.update({
            "metric": "GCN-layer fwd+bwd aggregated edges/sec",
            "value": main["value"],
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main["ms_per_step"],
            "higher_is_better": True,
            # strong: ONE graph (the N = 1 bench's), partitioned N ways -- north_star's "edges/sec reported at 1/2/4/8 MI355X" on
            # ogbn-products; total work fixed as N grows.  weak (--scaling weak, and config 5): a new products-shaped range per GPU
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": strong_workload if strong else weak_workload,
                "nv_per_gpu": main["nv"], "ne_total_with_selfloops": int(main["total_edges"]), "D": D, "scale": args.scale,
                "cut_fraction": (world - 1) / world if strong else cut,
                "cut_fraction_measured": main["cut_fraction_measured"],
                "boundary": "uniform (a cut edge may end at any vertex of the two ranges)",
                "halo_rows_total": main["halo_rows_total"],
                "halo_bytes_per_step_total": main["halo_bytes_per_step_total"],
                # slowest rank, measured after the timed region: one exchange on its own (pack + all-to-all + wait), the
                # pack alone, and the owned-edge aggregation kernels of one step that run while the two exchanges fly
                "halo_exchange_standalone_ms": main["exch_ms"], "halo_pack_ms": main["pack_ms"],
                "owned_edge_spmm_ms_per_step": main["owned_edge_spmm_ms_per_step"],
                "breakdown_ms_per_step_rank0": main["breakdown"],
                "partition_mode_rank0": main["partition_mode"],
                "halo_send_stats_rank0": main["halo_send_stats"],
                "parallelism": f"vertex-range x{world}",
                # rccl_ranks: what ncclCommCount reports for the communicator that carried the halo rows (0: RCCL not used)
                "transport": transport, "rccl_ranks": rccl_ranks,
                # transports whose set-up hit its deadline on some rank (the call is then left behind in a helper thread and the
                # run continued over the next transport: not a clean figure); [] in a clean run
                "comm_init_timed_out": comm_timed_out,
                # several ranks on one device (a one-GPU box rehearsing the N-rank code): every timing of this record is then
                # ranks taking turns on one chip -- NOT evidence of scaling, of a roofline fraction or of anything else
                "ranks_share_device": share,
                # CUs the persistent fused aggregation leaves free while an exchange is in flight (the effective figure: option,
                # GAIB_COMM_RESERVE_CUS or the communicator's default -- 32 under RCCL with more than one rank, 0 on the peer-to-peer pull)
                "cu_reserve_for_transport": ctx.get_option("comm_reserve_cus"),
                # north_star's curve: the N = 1 bench graph partitioned N ways (the headline itself under --scaling strong)
                "strong_products": strong_rec,
                # one products-shaped vertex range per GPU at cut 0.1 (the headline itself under --scaling weak)
                "weak_products_range": weak_rec,
                # the run's own constants, measured on the live headline case
                "cu_reserve_ab": ab["cu_reserve_ab"], "transport_ab": ab["transport_ab"],
                # the exchange in 1 / 2 / 4 / 8 time slices, the halo-column half consumed piece by piece (round 6)
                "halo_pipeline_ab": ab["halo_pipeline_ab"], "halo_pieces_rank0": main["halo_pieces"],
                # the weak case's cut with the cut edges on a boundary band (what a METIS / breadth-first partition looks like)
                "clustered_boundary": clustered,
                # the other end of the weak case's partition-quality axis, same invocation
                "random_order": {k: v for k, v in extra.items() if k != "parity"} if isinstance(extra, dict) else extra,
                "config5_papers100M": config5,
                "xgmi_link_probe": link,
                "budget": {"budget_s": budget.total_s, "elapsed_s": round(budget.elapsed(), 1), "headline_case_s": round(state["headline_s"], 1)},
            },
            "roofline": {
                "bound": "hbm", "kernel": main["kernel_name"],
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": main["alg_bytes"], "avg_launch_ms": main["avg_ms"], "launches": main["launches"],
                **({"ranks_share_device": True, "note": "ranks took turns on ONE device: this launch time includes waiting for the "
                    "other ranks' kernels and is not a roofline measurement"} if share else {}),
            },
            # the N = 1 workload's CPU baseline (named as such), timed on rank 0's host cores in this run
            "cpu_baseline": cpu_rec if cpu_rec is not None else ({"skipped": "--no-cpu-baseline"} if cpu_leg is None else None),
        })
        return rec

    def on_headline(res):
        state["main"] = res
        state["headline_s"] = time.time() - t_case
        if hold is not None and rank == 0:
            hold(assemble())  # from here on the headline value cannot be lost

    # the comparison of the strong case with the oracle's run on the bench graph: budgeted like every other leg (rank 0 spends
    # ~30 s of host time at full size: inputs of all ranks, the oracle's layer, the scatter of its rows), and its timing of the
    # oracle IS the N = 1 workload's CPU baseline
    def strong_check_budgeted(bounds):
        if strong_check is None:
            return None
        chk = strong_check(comm, bounds)
        need = 25.0 + 60.0 * min(args.scale, 1.0)

        class Budgeted:
            inputs = staticmethod(chk.inputs)
            exchanger = None

            def __call__(self_, part, layer, feat_out, grad_out):
                if not budget.agree(need):
                    return budget.skipped(need) if rank == 0 else None
                chk.exchanger = self_.exchanger
                return chk(part, layer, feat_out, grad_out)

        return Budgeted()

    t_case = time.time()
    if strong:
        main = run_strong(on_headline, strong_check_budgeted if strong_check is not None else None, extras=headline_legs)
        # the comparison's run of the oracle on the whole bench graph IS the N = 1 workload's CPU baseline (rank 0 has it; every
        # rank must know whether it exists: the bounded-sample leg below is collective)
        if rank == 0 and isinstance(main.get("parity"), dict) and main["parity"].get("cpu_baseline"):
            cb = main["parity"].pop("cpu_baseline")
            if cpu_leg is not None:  # (--no-cpu-baseline: the slot says skipped)
                cpu_rec = cb
        have = [cpu_rec is not None]
        dist.broadcast_object_list(have, src=0)
        if have[0] and rank != 0:
            cpu_rec = {"on": "rank 0"}
    else:
        main = run_weak(on_headline, mk(cut), extras=headline_legs)
        torch.cuda.empty_cache()
    state["main"] = main
    headline_s = state["headline_s"]
    if hold is not None and rank == 0:
        hold(assemble())
    # ---- what follows only adds to the record; every sub-case is budgeted --------------------------------------------
    if cpu_leg is not None and cpu_rec is None:
        need = 45.0
        if budget.agree(need):
            if rank == 0:
                try:
                    cpu_rec = cpu_leg(12.0)
                except Exception as e:  # noqa: BLE001 -- a failing baseline must not cost the scaling record
                    cpu_rec = {"error": f"{type(e).__name__}: {e}"[:300]}
            dist.barrier()
        else:
            cpu_rec = budget.skipped(need)
        if hold is not None and rank == 0:
            hold(assemble())
    if os.environ.get("GAIB_BENCH_FAIL_AFTER_HEADLINE") == str(rank):  # test hook: this rank fails in a sub-case (tests/test_gpu_dist.py)
        raise RuntimeError("GAIB_BENCH_FAIL_AFTER_HEADLINE: injected failure after the headline case")
    # ---- second: the other scaling mode (a default run) --------------------------------------------------
    if default_run and os.environ.get("GAIB_BENCH_OTHER_SCALING", "1") != "0":
        need = 1.3 * headline_s + 15
        if budget.agree(need):
            if strong:
                r = run_weak(None, None)
                other = sub_record(r, workload=weak_workload, cut_fraction=cut, nv_per_gpu=r["nv"],
                                   ne_total_with_selfloops=int(r["total_edges"]))
                weak_headline_s = time.time() - t_case - headline_s
            else:
                r = run_strong(None, None)
                other = {**sub_record(r, workload=strong_workload, cut_fraction=(world - 1) / world, nv_per_gpu=r["nv"],
                                      ne_total_with_selfloops=int(r["total_edges"])), **strong_extra}
            torch.cuda.empty_cache()
        else:
            other = budget.skipped(need)
        if hold is not None and rank == 0:
            hold(assemble())
    # what the weak sub-cases below cost: a case of the weak generator's size (the headline's under --scaling weak)
    weak_s = headline_s if not strong else (weak_headline_s if (isinstance(other, dict) and "value" in other) else 2.0 * headline_s + 5)
    weak_default = args.cut_fraction is None and world > 1  # (config 5's workload included: it has the same sub-cases)
    weak_parity = parity_check is not None and world > 1 and (strong or main["parity"] is None)
    if weak_parity:
        # element-wise against the oracle's GLOBAL run, at a global size the oracle finishes in seconds: 4.9 M vertices in all
        import argparse

        nv_weak = max(int(synth.SHAPES[shape][0] * args.scale), 16)
        scale_p = min(args.scale, 4.9e6 * args.scale / max(world * nv_weak, 1))
        need = 75.0
        if budget.agree(need):
            a2 = argparse.Namespace(**{**vars(args), "scale": scale_p, "steps": 2, "warmup": 1})
            rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=scale_p, selfloops=True)
            r = _bench_case(ctx, comm, a2, rank, world, D, log, rows, f"parity leg, scale {scale_p:.3f}, cut {cut:.3f}",
                            check=parity_check(shape, cut, comm, scale_p))
            if rank == 0:
                parity_scaled = {**(r["parity"] or {"error": "no record", "ok": None}), "scale": scale_p,
                                 "partition_mode_rank0": r["partition_mode"]["mode"],
                                 "of": f"the partitioned layer of this run ({world} ranks, {transport}) on the weak generator at scale "
                                       f"{scale_p:.3f} ({r['nv']} vertices per rank), not part of `value`"}
            del rows
            torch.cuda.empty_cache()
            # ... and on the clustered-boundary generator, where the rule takes the row classes (one pass over [owned | halo])
            if weak_default and budget.agree(need):
                rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=scale_p, selfloops=True,
                                        boundary="clustered", band=0.2)
                r = _bench_case(ctx, comm, a2, rank, world, D, log, rows, f"parity leg (clustered boundary), scale {scale_p:.3f}",
                                check=parity_check(shape, cut, comm, scale_p, "clustered"))
                if rank == 0:
                    pc = {**(r["parity"] or {"error": "no record", "ok": None}), "partition_mode_rank0": r["partition_mode"]["mode"]}
                    parity_scaled["clustered_boundary"] = pc
                    if pc.get("ok") is False:
                        parity_scaled["ok"] = False
                del rows
                torch.cuda.empty_cache()
            elif weak_default and rank == 0:
                parity_scaled["clustered_boundary"] = budget.skipped(need)
        elif rank == 0:
            parity_scaled = budget.skipped(need)
        if hold is not None and rank == 0:
            hold(assemble())
    if weak_default and os.environ.get("GAIB_BENCH_CLUSTERED", "1") != "0":
        need = 1.2 * weak_s + 10
        if budget.agree(need):
            rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=args.scale,
                                    selfloops=True, boundary="clustered", band=0.2)
            r = _bench_case(ctx, comm, args, rank, world, D, log, rows, f"clustered boundary, cut {cut:.3f}")
            clustered = sub_record(r, cut_fraction=cut, boundary="clustered: the cut edges land on a boundary band, the first 20 % "
                                   "of every range's ids, one slice facing each peer (synth.block_rows)")
            del rows
            torch.cuda.empty_cache()
        else:
            clustered = budget.skipped(need)
        if hold is not None and rank == 0:
            hold(assemble())
    if weak_default and os.environ.get("GAIB_BENCH_RANDOM_ORDER", "1") != "0":
        rcut = (world - 1) / world
        need = 1.6 * weak_s + 10  # (the halo of a random order is several times the headline's: longer set-up and exchanges)
        if budget.agree(need):
            rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=rcut, device="cuda",
                                    scale=args.scale, selfloops=True)
            r = _bench_case(ctx, comm, args, rank, world, D, log, rows, f"random order, cut {rcut:.3f}", check=mk(rcut))
            extra = sub_record(r, cut_fraction=rcut, parity=r["parity"])
            del rows
            torch.cuda.empty_cache()
        else:
            extra = budget.skipped(need)
        if hold is not None and rank == 0:
            hold(assemble())
    # BASELINE config 5 inside the default 8-GPU run: at N = 8 one vertex range of 1/8 of the papers100M shape per rank IS
    # the papers100M-shaped graph, so the scaling run that measures the metric's "1/2/4/8" half also yields config 5's number
    # (locality-preserving end of the partition axis; `--workload gcn-papers` gives both ends).  GAIB_BENCH_CONFIG5=0 skips.
    c5 = os.environ.get("GAIB_BENCH_CONFIG5", "1")  # "force": also at other N / scales (the one-GPU test of this branch)
    if not papers and args.cut_fraction is None and ((world == 8 and args.scale == 1.0 and c5 != "0") or c5 == "force"):
        need = 4.5 * weak_s + 20  # 5.7 x the rows, 3.3 x the edges of a products-shaped range
        if not budget.agree(need):
            config5 = budget.skipped(need)
        else:
            ok = 1
            try:
                t0 = time.time()
                rows = synth.block_rows("ogbn-papers100M/8", rank, world, seed=42, cut_fraction=0.1, device="cuda",
                                        scale=args.scale, selfloops=True)
            except Exception as e:  # noqa: BLE001 -- an allocation failure here must not cost the scaling record
                log(f"[bench r{rank}] config 5 graph generation failed: {type(e).__name__}: {e}")
                rows, ok = None, 0
            if reduce_min(ok):  # every rank runs the case, or none does
                r = _bench_case(ctx, comm, args, rank, world, D, log, rows, "config 5: papers100M shape, cut 0.100")
                config5 = sub_record(r, workload="BASELINE config 5: ogbn-papers100M-shaped graph (111 M vertices, 3.2 G edges incl. "
                                     "self loops) in 8 vertex ranges, GCN hidden layer 128->128 fwd+bwd, cut 0.1", unit="edges/s",
                                     nv_per_gpu=r["nv"], ne_total_with_selfloops=int(r["total_edges"]),
                                     set_up_and_run_s=time.time() - t0)
            else:
                config5 = {"error": "graph generation failed on some rank (see stderr)"}
            del rows
            torch.cuda.empty_cache()
    transport_leg()
    if comm is not None:
        comm.barrier()
    return assemble()


def _one_rank_timing(ctx, g1, args, D) -> dict:
    """the single-GPU layer (the N = 1 bench's step) on the whole graph `g1`, args.steps steps after args.warmup: what
    `speedup_vs_n1` of the strong case divides by.  The graph moves into the layer's LearningGraph and goes with it."""
    from . import layers as L

    nv, ne = g1.nv, g1.ne
    lg = L.LGraph.adopt(g1)
    layer = L.Layer(L.GCN, 1, nv, D, D, lg, act=True, lr=0.01)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(43)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda", generator=gen))
    layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda", generator=gen))
    fo, go = torch.empty(nv, D, device="cuda"), torch.empty(nv, D, device="cuda")
    for _ in range(args.warmup):
        layer.forward(fo)
        layer.backward(fo, go)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        layer.forward(fo)
        layer.backward(fo, go)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    layer.close()
    lg.close()
    del fo, go
    torch.cuda.empty_cache()
    return {"value": 2 * ne * args.steps / el, "ms_per_step": el / args.steps * 1e3, "steps": args.steps,
            "of": "the single-GPU GCN layer step on the whole graph, rank 0's device, while the other ranks wait"}


def bench_gat_layer(ctx, args, rank: int, world: int, log, hold=None) -> dict:
    """`bench.py --gpus N --workload gat-reddit` (BASELINE config 4's layer across ranks; strong scaling): the reddit-shaped graph
    (seed 7, self loops: net.cpp:96) in N vertex ranges, the hidden GAT layer 64 -> 64 with 8 heads forward + backward per step on
    every rank.  The h rows (scores, forward aggregation) and the gradient rows (transposed aggregation) of the halo vertices
    travel through the halo exchange behind the C ABI; the weight and attention-vector gradients are summed over the ranks.
    `value` = 2 E steps / (max over ranks of the time of `steps` steps between barriers).  In the run: the ONE-rank step of the
    same layer on the whole graph (rank 0's device, the others wait) -- its time, and its outputs on rank 0's rows as the parity
    reference: the one-GPU layer is the one held against the oracle (tests/test_gpu_layers.py, bench.py --workload gat-reddit);
    backward is compared on the reference's forward output, i.e. on identical relu masks (as every parity block does)."""
    import numpy as np

    from . import capi, layers as L, synth

    d, H = 64, 8
    comm, transport = make_comm(ctx, rank, world, log)
    if comm is None:
        raise RuntimeError("the partitioned GAT layer runs over gaib_comm (GAIB_DIST_BACKEND = rccl | ipc), not over torch.distributed")
    L.set_comm(comm)
    rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t0 = time.time()
    sg = synth.make("reddit", seed=7, device="cuda", scale=args.scale)
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g1 = g0.add_selfloop()
    g0.close()
    del sg
    n, ne = g1.nv, g1.ne
    rp = g1.rowptr().cpu().numpy()
    ci = g1.colidx().cpu().numpy().view(np.uint32)
    if rank != 0:
        g1.close()
    torch.cuda.empty_cache()
    part = L.HostPartition(rp, ci, rank, world, gat=True)
    del rp, ci
    lo, hi = part.lo, part.hi
    lg = part.make_graph(comm)
    nv = hi - lo
    n_halo = int(len(part.halo_gids))
    log(f"[bench r{rank}] gat-reddit: rows [{lo},{hi}) of {n}, halo rows {n_halo}, set-up {time.time() - t0:.1f}s over {transport}")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(43)  # the SAME global inputs on every rank; a rank keeps its rows
    x_all = torch.randn(n, d, device="cuda", generator=gen)
    gin_all = torch.randn(n, d, device="cuda", generator=gen)
    layer = L.Layer(L.GAT, 1, nv, d, d, lg, act=True, lr=0.01)
    layer.set_heads(H)
    layer.write(L.FEAT_IN, x_all[lo:hi].contiguous())
    gin = gin_all[lo:hi].contiguous()
    layer.write(L.GRAD_IN, gin)
    out, go = torch.empty(nv, d, device="cuda"), torch.empty(nv, d, device="cuda")
    if rank != 0:
        del x_all, gin_all
    grads = ((L.W_NEIGH_GRAD, (d, d)), (L.ALPHA_LGRAD, (d, 1)), (L.ALPHA_RGRAD, (d, 1)))

    def step():
        layer.forward(out)
        layer.backward(out, go)
        for which, shape in grads:
            allreduce_layer_grads(ctx, layer, [which], shape, comm=comm)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    mine = time.perf_counter() - t1
    ctx.prof_enable(False)
    tmax = torch.tensor([mine], dtype=torch.float64, device=rdev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0])
    prof = {k: v["ms"] / args.steps for k, v in ctx.prof_table().items()}
    ctx.prof_reset()
    halo_rows = torch.tensor([n_halo], dtype=torch.float64, device=rdev)
    dist.all_reduce(halo_rows)
    devs = [None] * world
    dist.all_gather_object(devs, (socket.gethostname(), ctx.device))
    result = None
    if rank == 0:
        result = {
            "metric": "GAT-layer fwd+bwd aggregated edges/sec", "value": 2 * ne * args.steps / elapsed, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"reddit-shaped Chung-Lu graph (seed 7) partitioned into {world} vertex ranges, GAT hidden layer "
                                   "64->64 with 8 heads fwd+bwd per rank (h and gradient rows of the halo vertices exchanged, weight "
                                   "and attention-vector gradients all-reduced)",
                       "nv": n, "ne_with_selfloops": ne, "D": d, "heads": H, "scale": args.scale, "parallelism": f"{world} vertex ranges",
                       "transport": transport, "ranks_share_device": len(set(devs)) < world, "halo_rows_total": int(halo_rows[0]),
                       "halo_rows_rank0": n_halo, "rows_rank0": nv, "comm_init_timed_out": list(COMM_SETUP["timed_out"])},
            "breakdown_ms_per_step_rank0": prof,
            "roofline": {"bound": "infinity-cache gather", "kernel": max(prof, key=prof.get) if prof else None,
                         "achieved": None, "peak": 8600.0, "unit": "GB/s", "frac": None, "traffic": None,
                         "note": "no PMC pass of the N-rank GAT step exists; the one-GPU record (--workload gat-reddit) prices the "
                                 "sweep kernels against the cache-resident gather rate"},
            "cpu_baseline": {"skipped": "the N = 1 record of this workload carries the CPU baseline (rank 0 at N = 1 only)"},
        }
        if hold:
            hold(result)
    # the one-rank step of the same layer on the whole graph + parity of rank 0's rows against it (the others wait at the barrier)
    ref_out = torch.empty(nv, d, device="cuda")
    if rank == 0:
        lg1 = L.LGraph.adopt(g1)
        l1 = L.Layer(L.GAT, 1, n, d, d, lg1, act=True, lr=0.01)
        l1.set_heads(H)
        l1.write(L.FEAT_IN, x_all)
        l1.write(L.GRAD_IN, gin_all)
        o1, g1o = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
        for _ in range(max(1, args.warmup)):
            l1.forward(o1)
            l1.backward(o1, g1o)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            l1.forward(o1)
            l1.backward(o1, g1o)
        torch.cuda.synchronize()
        one_ms = (time.perf_counter() - t2) / args.steps * 1e3
        l1.write(L.GRAD_IN, gin_all)
        l1.forward(o1)
        l1.backward(o1, g1o)
        torch.cuda.synchronize()
        ref_out.copy_(o1[lo:hi])
        result["config"]["one_rank_same_graph"] = {"ms_per_step": one_ms, "value": 2 * ne / (one_ms * 1e-3)}
        result["config"]["speedup_vs_n1"] = one_ms / result["ms_per_step"]
    dist.barrier()
    # every rank: backward once more on the forward output it would have had with the reference's masks.  Ranks other than 0 have
    # no reference: they run the same collective step on their own output.
    layer.write(L.GRAD_IN, gin)
    layer.forward(out)
    fwd_mine = out.clone()
    layer.write(L.GRAD_IN, gin)
    layer.backward(ref_out if rank == 0 else out, go)
    for which, shape in grads:
        allreduce_layer_grads(ctx, layer, [which], shape, comm=comm)
    barrier()
    if rank == 0:
        def err(a, b):
            scale = float(b.abs().max())
            return {"inf": float((a - b).abs().max()) / scale if scale > 0 else 0.0,
                    "elem": float(((a - b).abs() / torch.clamp(b.abs(), min=1e-5 * scale)).max())}

        tol = 1e-4
        par = {"tol": tol, "against": "the one-rank run of the same layer on the whole graph, rank 0's rows (forward as is; backward on the "
                                      "reference's forward output: identical relu masks)",
               "forward": err(fwd_mine, o1[lo:hi]), "grad_out": err(go, g1o[lo:hi]),
               "relu_mask_flips": {"count": int(((fwd_mine > 0) != (o1[lo:hi] > 0)).sum().item()), "of": int(fwd_mine.numel())}}
        # (the all-reduced gradients of the parity pass: ranks other than 0 ran backward on their OWN masks -- a handful of flips
        # among 1e7 outputs moves a K = 233 k sum by less than the tolerance, and the comparison says so if it does not)
        for name, which, shape in (("W_grad", L.W_NEIGH_GRAD, (d, d)), ("alpha_l_grad", L.ALPHA_LGRAD, (d,)), ("alpha_r_grad", L.ALPHA_RGRAD, (d,))):
            a, b = layer.tensor(which, shape), l1.tensor(which, shape)
            sc = float(b.abs().max())
            par[name] = {"inf": float((a - b).abs().max()) / sc if sc > 0 else 0.0}
        par["ok"] = bool(par["forward"]["inf"] <= tol and par["grad_out"]["inf"] <= tol and par["W_grad"]["inf"] <= tol
                         and par["alpha_l_grad"]["inf"] <= 10 * tol and par["alpha_r_grad"]["inf"] <= 10 * tol)
        par["alpha_tolerance"] = ("1e-3: 64 sums over 9e8 (edge, head) terms with leaky_relu' jumping at 0 -- two correct fp32 evaluations "
                                  "of a score within rounding of zero take either slope (bench.py --workload gat-reddit holds each side "
                                  "to 1e-4 of fp64 on its own signs)")
        result["parity"] = par
        l1.close()
        lg1.close()
    layer.close()
    lg.close()
    part.close()
    L.set_comm(None)
    comm.close()
    return result
