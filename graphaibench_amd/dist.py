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


def split_by_owner(rowptr_local: torch.Tensor, colidx_global: torch.Tensor, lo: int, hi: int):
    """the local (no communication) half of the partition: rows [lo, hi) of the global CSR -> an owned-column CSR
    (column ids relative to lo), a halo-column CSR (column ids index `halo`), the sorted global ids of the halo
    vertices and the rows' full degrees.  owned + halo == the vertex set of the reference's induced subgraph
    (graph_partition.cc:150-166), pinned against it in tests/test_dist_cpu.py."""
    device = colidx_global.device
    n_own = hi - lo
    assert rowptr_local.numel() == n_own + 1
    rowptr_local = rowptr_local.to(torch.int64)
    cols = colidx_global.to(torch.int64)
    deg = rowptr_local[1:] - rowptr_local[:-1]
    rows = torch.repeat_interleave(torch.arange(n_own, device=device), deg)
    own = (cols >= lo) & (cols < hi)
    halo = torch.unique(cols[~own])  # sorted

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
    rp_own, ci_own, rp_halo, ci_halo, halo, deg = split_by_owner(rowptr_local, colidx_global, lo, hi)
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


def make_comm(ctx, rank: int, world: int, log):
    """The data-path communicator of bench.py's N > 1 leg, behind the C ABI.  GAIB_DIST_BACKEND:
         rccl (default)  gaib_comm over RCCL -- one GPU per rank
         ipc             gaib_comm over hipIpc peer-to-peer pull (also several ranks on one GPU: tests)
         nccl | gloo     no gaib_comm: torch.distributed moves the halo rows (the round-1 path)
    torch.distributed (the launcher's process group) only carries the 128-byte id, barriers and the timing reductions.
    The choice is made ONCE here, by all ranks together: if any rank fails to create its communicator, every rank
    falls back to torch.distributed (an all-reduce(MIN) of the success flag) -- never per call."""
    import os

    from . import capi

    backend = os.environ.get("GAIB_DIST_BACKEND", "rccl")
    if backend in ("nccl", "gloo"):
        return None, f"torch.distributed/{dist.get_backend()}"

    def attempt(transport):
        """every rank tries; the outcome is agreed on by all of them (all-reduce(MIN) of the success flag)"""
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
            # left behind in its daemon thread -- and all ranks move on to the next transport together.  (The same path
            # creates the peer-to-peer communicator, so every N > 1 run exercises it.)
            import threading

            box = {}

            def create():
                try:
                    box["comm"] = capi.Comm(ctx, rank, world, uid[0], transport)
                except Exception as e:  # noqa: BLE001
                    box["err"] = str(e)

            limit = float(os.environ.get("GAIB_COMM_INIT_TIMEOUT_S", "90"))
            th = threading.Thread(target=create, daemon=True, name="gaib-comm-init")
            th.start()
            th.join(limit)
            if th.is_alive():
                ok, err = 0, f"communicator set-up did not return within {limit:.0f} s (left behind in its thread)"
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

    def rccl_possible():
        """the cheap preconditions of ncclCommInitRank, agreed by all ranks BEFORE anybody enters it (the call is
        collective and has no deadline: a rank that cannot load RCCL, or two ranks on one device, would leave the
        others waiting inside it): librccl loads on every rank, and every rank has a device of its own"""
        mine = [ctx.device, 1 if capi.comm_transport_available(capi.COMM_RCCL) else 0]
        every = [None] * world
        dist.all_gather_object(every, mine)
        if not all(e[1] for e in every):
            return False, "librccl.so.1 does not load on rank(s) " + str([r for r, e in enumerate(every) if not e[1]])
        if len({e[0] for e in every}) != world:
            return False, f"ranks share devices ({[e[0] for e in every]}): RCCL wants one GPU per rank"
        return True, ""

    # rccl (default): if it cannot be set up on some rank, the peer-to-peer pull transport (hipIpc handles, device-to-
    # device copies over xGMI) is tried before torch.distributed carries the rows
    order = [capi.COMM_IPC] if backend == "ipc" else [capi.COMM_RCCL, capi.COMM_IPC]
    failed = []
    for transport in order:
        name = "ipc" if transport == capi.COMM_IPC else "rccl"
        if transport == capi.COMM_RCCL:
            ok, err = rccl_possible()
            if not ok:
                log(f"[bench r{rank}] gaib_comm(rccl) not attempted: {err}")
                failed.append("rccl")
                continue
        comm, err = attempt(transport)
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


def _bench_case(ctx, comm, args, rank, world, D, log, rows, label, check=None):
    """one timed case: partition `rows`, build the layer, warm up, time args.steps steps.
    check(part, layer, feat_out, grad_out, inputs): bench.py's comparison with the oracle's GLOBAL run (test
    infrastructure stays outside this package); the layer then runs on inputs(rank, nv) -> (x, grad_in) host arrays"""
    from . import layers as L

    t0 = time.time()
    part = build_partition(rows.rowptr, rows.colidx_global, rows.n_global, rank, world)
    dg = DistLayerGraph(ctx, part, comm)
    mode_used, n_bnd, bnd_edges = dg.lgraph.partition_mode(D)  # decided (and the class graphs built) before the timed steps
    torch.cuda.synchronize()
    log(f"[bench r{rank}] {label}: rows [{part.lo},{part.hi}) ne={part.ne} (own-column {part.colidx_own.numel()}) "
        f"halo rows={part.n_halo} send rows={part.send_idx.numel()} mode {L.LGraph.PART_NAMES[mode_used]} "
        f"boundary rows={n_bnd} setup {time.time()-t0:.1f}s")
    nv = part.n_own
    torch.manual_seed(43 + rank)
    layer = L.Layer(L.GCN, 1, nv, D, D, dg.lgraph, act=True, lr=0.01)
    if check is not None:
        x_h, gin_h = check.inputs(rank, nv)
        layer.write(L.FEAT_IN, torch.from_numpy(x_h).cuda())
        layer.write(L.GRAD_IN, torch.from_numpy(gin_h).cuda())
        del x_h, gin_h
    else:
        layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda"))
        layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    feat_out = torch.empty(nv, D, device="cuda")
    grad_out = torch.empty(nv, D, device="cuda")

    def step():
        layer.forward(feat_out)
        layer.backward(feat_out, grad_out)
        allreduce_layer_grads(ctx, layer, [L.W_NEIGH_GRAD], (D, D), comm=comm)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    dg.ex.bytes_sent = 0
    dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t_start
    ctx.prof_enable(False)
    n_light, ms_light = ctx.prof_get("spmm_light")
    n_fused, ms_fused = ctx.prof_get("spmm_gemm_fused")
    n_heavy, ms_heavy = ctx.prof_get("spmm_heavy")
    n_gemm, ms_gemm = ctx.prof_get("sgemm")
    n_pack, ms_pack = ctx.prof_get("gather_rows")
    part_ms = {k: ctx.prof_get(k)[1] / args.steps for k in ("part_fused", "part_fused_acc", "part_fused_2t", "part_light",
                                                            "part_light_acc", "part_light_2t")}
    ctx.prof_reset()
    if check is not None and hasattr(check, "exchanger"):
        check.exchanger = dg.ex  # (diagnostics of a failing comparison: the plan the layer itself uses)
    parity = check(part, layer, feat_out, grad_out) if check is not None else None
    # diagnostics outside the timed region (collective: every rank runs them): one halo exchange of a [nv x D]
    # matrix on its own (pack + all-to-all + wait) and the pack alone -- what the owned-edge SpMM has to hide
    bytes_timed = dg.ex.bytes_sent
    feat = layer.tensor(L.FEAT_IN, (nv, D))
    reps = 3
    dg.ex.exchange(feat, D)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dg.ex.exchange(feat, D)
    torch.cuda.synchronize()
    exch_ms = (time.perf_counter() - t0) / reps * 1e3
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
    rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed, exch_ms, pack_ms], dtype=torch.float64, device=rdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    e = torch.tensor([float(part.ne), float(part.n_halo), float(bytes_timed)], dtype=torch.float64, device=rdev)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    # dominant kernel (rank 0's view) = the pass over the owned-column edges of each aggregation: with halo
    # edges that is spmm_w64_kernel (the halo half then carries the dense product), without them the fused
    # kernel does everything, as in the single-GPU bench
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
    elif part.colidx_halo.numel() > 0:
        kernel_name = "spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer> over the owned-column edges (rank 0)"
        alg_bytes = e_light * (4 * D + 8) + (nv - st_own["n_heavy"]) * 4 * D + (nv + 1) * 8
        n_dom, ms_dom = n_light, ms_light
    else:
        kernel_name = "spmm_gemm_kernel<VEC=2,edge-weights,U=16,buffer> (aggregation + MFMA dense product, rank 0)"
        alg_bytes = e_light * (4 * D + 8) + int(1.5 * nv * 4 * D) + (nv + 1) * 8
        n_dom, ms_dom = n_fused, ms_fused
    avg_ms = ms_dom / max(n_dom, 1)
    res = dict(elapsed=float(t[0]), exch_ms=float(t[1]), pack_ms=float(t[2]), total_edges=float(e[0]),
               halo_rows_total=int(e[1]), halo_bytes_per_step_total=float(e[2]) / args.steps, nv=nv,
               owned_edge_spmm_ms_per_step=ms_light / args.steps, kernel_name=kernel_name, alg_bytes=alg_bytes,
               avg_ms=avg_ms, launches=n_dom, parity=parity,
               # rank 0's kernels per step: the owned-column pass, the halo-column half (fused with the dense product),
               # heavy rows, the weight gradient, the pack of the rows on the send lists
               breakdown=dict(owned_edge_spmm_ms=ms_light / args.steps, halo_half_ms=ms_fused / args.steps,
                              heavy_rows_ms=ms_heavy / args.steps, sgemm_ms=ms_gemm / args.steps,
                              pack_ms=ms_pack / args.steps, **{k + "_ms": v for k, v in part_ms.items() if v}),
               # how rank 0 aggregates on this partition (LearningGraph::partition_mode) and what decided it
               partition_mode=dict(mode=L.LGraph.PART_NAMES[mode_used], boundary_rows=n_bnd, boundary_row_share=n_bnd / max(nv, 1),
                                   boundary_edges=bnd_edges, link_gbs_assumed=float(os.environ.get("GAIB_LINK_GBS", "100"))),
               value=2 * float(e[0]) * args.steps / float(t[0]), ms_per_step=float(t[0]) / args.steps * 1e3)
    # the case's objects go now, in an order: the layer, the graph (which points at the plan), then the plan -- collectively (every
    # rank is here): its send buffer and halo table return to the communicator's pool for the next case, and a default run never
    # comes near the communicator's limit of plans alive at once
    plan = getattr(dg.ex, "halo", None)
    layer.close()
    dg.lgraph.close()
    del layer, feat_out, grad_out, dg
    torch.cuda.synchronize()
    if plan is not None:
        plan.close()
    torch.cuda.empty_cache()
    return res


def bench_gcn_layer(ctx, args, rank: int, world: int, D: int, log, make_check=None, t_start=None, hold=None, cpu_leg=None,
                    parity_check=None, traffic_of=None):
    """bench.py's N > 1 leg.  GCN hidden layer D -> D forward + backward per step, halo exchange before each of the 2
    SpMM, one all-reduce of dW per step.

    --scaling weak (default): every rank owns a products-shaped vertex range of one global block Chung-Lu graph
    (synth.block_rows).  The share of a range's edges that cross ranges stands for the partitioner's quality, so BOTH
    ends are measured in one invocation: `value` at --cut-fraction (default 0.1, a locality-preserving order) and
    `config.random_order` at (N-1)/N (a random vertex order: the adversarial end).
    --scaling strong: the SAME 2.45 M-vertex products-shaped graph of the single-GPU bench, partitioned N ways by
    vertex range (its vertex order is random, so the cut is (N-1)/N).
    --workload gcn-papers (BASELINE config 5): the same layer on the ogbn-papers100M-shaped graph, one vertex range of
    1/8 of it per rank -- at N = 8 the whole 111 M-vertex / 3.2 G-edge graph -- again at both ends of the partition-quality
    axis.  make_check(shape, cut) -> bench.py's oracle comparison for one case (--check-oracle), or None.
    The record cannot be lost: the HEADLINE case runs first and rank 0 hands the record to hold() as soon as it is measured
    (bench.py prints it if anything ends the run early); every further sub-case -- the N = 1 CPU baseline (cpu_leg), the
    clustered-boundary generator, the random vertex order, config 5 -- starts only if all ranks agree that its estimated
    time fits args.budget_s counted from t_start (Budget), else its slot says {"skipped": "budget", ...}.
    parity_check(shape, cut, comm, scale) -> bench.py's oracle comparison at another scale: the record's `parity` block of a
    default run -- the same partitioned layer over the same transport on a graph whose GLOBAL size the oracle runs in seconds
    (two products-shaped ranges in all), every rank's rows element-wise against the oracle's global run."""
    import os

    from . import capi, synth

    comm, transport = make_comm(ctx, rank, world, log)
    rdev0 = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def reduce_min(flag: int) -> int:
        t = torch.tensor([flag], dtype=torch.int32, device=rdev0)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    budget = Budget(getattr(args, "budget_s", 1e9), t_start if t_start is not None else time.time(), reduce_min)
    papers = getattr(args, "workload", "gcn-products") == "gcn-papers"
    shape = "ogbn-papers100M/8" if papers else "ogbn-products"
    mk = (lambda c: make_check(shape, c, comm)) if make_check is not None else (lambda c: None)
    # the xGMI link, measured (rank 0 while the others wait): replaces the 153 GB/s spec constant in the record
    link = None
    if rank == 0 and torch.cuda.device_count() >= 2:
        try:
            link = {"unidirectional_gbs": capi.probe_peer_copy(0, 1, 1 << 28, 10, False),
                    "bidirectional_gbs_per_direction": capi.probe_peer_copy(0, 1, 1 << 28, 10, True), "bytes": 1 << 28,
                    "pair": [0, 1]}
        except capi.GaibError as e:
            link = {"error": str(e)[:200]}
    # the partition-mode rule (LearningGraph::partition_mode) prices the exchange per peer pair: with the MEASURED link rate
    # where there is one (every rank gets rank 0's figure before any partitioned graph exists), else the library's default
    got = [link["unidirectional_gbs"] if (rank == 0 and isinstance(link, dict) and "unidirectional_gbs" in link) else None]
    dist.broadcast_object_list(got, src=0)
    if got[0] and "GAIB_LINK_GBS" not in os.environ:
        os.environ["GAIB_LINK_GBS"] = f"{float(got[0]):.1f}"
    dist.barrier()
    strong = getattr(args, "scaling", "weak") == "strong"
    cut = 0.1 if args.cut_fraction is None else args.cut_fraction
    extra = clustered = config5 = cpu_rec = parity_scaled = None
    t_case = time.time()
    if strong:
        sg = synth.make("ogbn-products", seed=42, device="cuda", scale=args.scale)
        g0 = ctx.graph(sg.rowptr, sg.colidx)
        g1 = g0.add_selfloop()  # GCN aggregates over A + I (net.cpp:96)
        g0.close()
        rp_all, ci_all = g1.rowptr(), g1.colidx().to(torch.int64)
        n_global = g1.nv
        g1.close()
        b = partition_bounds(n_global, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = int(rp_all[lo]), int(rp_all[hi])
        rows = synth.BlockRows((rp_all[lo:hi + 1] - e0).contiguous(), ci_all[e0:e1].contiguous(), n_global, hi - lo)
        del rp_all, ci_all, sg
        torch.cuda.empty_cache()
        assert not papers, "--scaling strong partitions the single-GPU products graph; gcn-papers is defined per range"
        main = _bench_case(ctx, comm, args, rank, world, D, log, rows, "strong scaling, products graph")
        workload = (f"the single-GPU bench's ogbn-products-shaped graph (seed 42, random vertex order) partitioned into "
                    f"{world} vertex ranges, GCN hidden layer 128->128 fwd+bwd, halo exchange before each SpMM + dW all-reduce")
        cut_main = (world - 1) / world
    else:
        rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=args.scale,
                                selfloops=True)  # GCN aggregates over A + I (net.cpp:96)
        main = _bench_case(ctx, comm, args, rank, world, D, log, rows, f"cut {cut:.3f}", check=mk(cut))
        del rows
        torch.cuda.empty_cache()
        cut_main = cut
        if papers:
            workload = (f"BASELINE config 5: block Chung-Lu graph of {world} vertex range(s), each 1/8 of the ogbn-papers100M "
                        f"shape (seed 42; N = 8: 111 M vertices, 3.2 G edges), GCN hidden layer 128->128 fwd+bwd, halo "
                        f"exchange before each SpMM + dW all-reduce")
        else:
            workload = ("block Chung-Lu graph, one ogbn-products-shaped vertex range per GPU (seed 42), "
                        "GCN hidden layer 128->128 fwd+bwd, halo exchange before each SpMM + dW all-reduce")
    headline_s = time.time() - t_case  # set-up + warm-up + timed steps + diagnostics of one case of this size

    def sub_record(r, **more):
        return {"value": r["value"], "ms_per_step": r["ms_per_step"], "halo_rows_total": r["halo_rows_total"],
                "halo_bytes_per_step_total": r["halo_bytes_per_step_total"], "halo_exchange_standalone_ms": r["exch_ms"],
                "halo_pack_ms": r["pack_ms"], "owned_edge_spmm_ms_per_step": r["owned_edge_spmm_ms_per_step"],
                "breakdown_ms_per_step_rank0": r["breakdown"], "partition_mode_rank0": r["partition_mode"], **more}

    def assemble():
        achieved = main["alg_bytes"] / (main["avg_ms"] * 1e-3) / 1e9 if main["avg_ms"] > 0 else 0.0
        rccl_ranks = comm.size if (comm is not None and transport.startswith("gaib_comm/rccl")) else 0
        parity = main["parity"] if main["parity"] is not None else parity_scaled
        if isinstance(extra, dict) and "value" in extra:
            rp = extra.get("parity")
            if parity is not None and rp is not None:
                parity = {**parity, "random_order": rp, "ok": bool(parity["ok"] and rp["ok"])}
        rec = {"parity": parity} if parity is not None else {}
        # L2 -> fabric bytes per launch of the dominant kernel, where a PMC pass of exactly this kernel on exactly this shard
        # exists (bench.py: profiles/hbm_traffic.json, verified by source hashes): the owned-column pass of the split on the
        # products-shaped range at cut 0.1, uniform generator -- the same 113.6 M own-column edges at every N
        traffic, traffic_src = None, "no PMC pass of this kernel on this shard"
        if (traffic_of is not None and not strong and not papers and args.scale == 1.0 and abs(cut_main - 0.1) < 1e-9
                and main["partition_mode"]["mode"] == "split" and main["kernel_name"].startswith("spmm_w64_kernel")):
            traffic, traffic_src = traffic_of("spmm_w64_kernel_owned_pass_bytes_per_launch", "partitioned_products_uniform")
        rec.update({
            "metric": "GCN-layer fwd+bwd aggregated edges/sec",
            "value": main["value"],
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "nv_per_gpu": main["nv"], "ne_total_with_selfloops": int(main["total_edges"]), "D": D, "scale": args.scale,
                "cut_fraction": cut_main, "boundary": "uniform (a cut edge may end at any vertex of the two ranges)",
                "halo_rows_total": main["halo_rows_total"],
                "halo_bytes_per_step_total": main["halo_bytes_per_step_total"],
                # slowest rank, measured after the timed region: one exchange on its own (pack + all-to-all + wait), the
                # pack alone, and the owned-edge aggregation kernels of one step that run while the two exchanges fly
                "halo_exchange_standalone_ms": main["exch_ms"], "halo_pack_ms": main["pack_ms"],
                "owned_edge_spmm_ms_per_step": main["owned_edge_spmm_ms_per_step"],
                "breakdown_ms_per_step_rank0": main["breakdown"],
                "partition_mode_rank0": main["partition_mode"],
                "parallelism": f"vertex-range x{world}",
                # rccl_ranks: what ncclCommCount reports for the communicator that carried the halo rows (0: RCCL not used)
                "transport": transport, "rccl_ranks": rccl_ranks,
                # CUs the persistent fused aggregation leaves free while an exchange is in flight (gaib_comm_init: 32 under
                # RCCL with more than one rank -- 1.5 % of that kernel, DESIGN 3.5 --, 0 on the peer-to-peer pull transport)
                "cu_reserve_for_transport": ctx.get_option("comm_reserve_cus"),
                # the same cut with the cut edges on a boundary band (what a METIS / breadth-first partition looks like)
                "clustered_boundary": clustered,
                # the other end of the partition-quality axis, same invocation (weak scaling only)
                "random_order": {k: v for k, v in extra.items() if k != "parity"} if isinstance(extra, dict) else extra,
                "config5_papers100M": config5,
                "xgmi_link_probe": link,
                "budget": {"budget_s": budget.total_s, "elapsed_s": round(budget.elapsed(), 1), "headline_case_s": round(headline_s, 1)},
            },
            "roofline": {
                "bound": "hbm", "kernel": main["kernel_name"],
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": main["alg_bytes"], "avg_launch_ms": main["avg_ms"], "launches": main["launches"],
            },
            # the N = 1 workload's CPU baseline (named as such), timed on rank 0's host cores in this run
            "cpu_baseline": cpu_rec,
        })
        return rec

    if hold is not None and rank == 0:
        hold(assemble())  # from here on the headline value cannot be lost
    # ---- what follows only adds to the record; every sub-case is budgeted --------------------------------------------
    if cpu_leg is not None:
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
    weak_default = not strong and args.cut_fraction is None and world > 1
    if os.environ.get("GAIB_BENCH_FAIL_AFTER_HEADLINE") == str(rank):  # test hook: this rank fails in a sub-case (tests/test_gpu_dist.py)
        raise RuntimeError("GAIB_BENCH_FAIL_AFTER_HEADLINE: injected failure after the headline case")
    if parity_check is not None and not strong and world > 1 and main["parity"] is None:
        # element-wise against the oracle's GLOBAL run, at a global size the oracle finishes in seconds: 4.9 M vertices in all
        import argparse

        scale_p = min(args.scale, 4.9e6 * args.scale / max(world * main["nv"], 1))
        need = 75.0
        if budget.agree(need):
            a2 = argparse.Namespace(**{**vars(args), "scale": scale_p, "steps": 2, "warmup": 1})
            rows = synth.block_rows(shape, rank, world, seed=42, cut_fraction=cut, device="cuda", scale=scale_p, selfloops=True)
            r = _bench_case(ctx, comm, a2, rank, world, D, log, rows, f"parity leg, scale {scale_p:.3f}, cut {cut:.3f}",
                            check=parity_check(shape, cut, comm, scale_p))
            if rank == 0:
                parity_scaled = {**(r["parity"] or {"error": "no record", "ok": None}), "scale": scale_p,
                                 "partition_mode_rank0": r["partition_mode"]["mode"],
                                 "of": f"the partitioned layer of this run ({world} ranks, {transport}) on the same generator at scale "
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
        need = 1.2 * headline_s + 10
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
        need = 1.6 * headline_s + 10  # (the halo of a random order is several times the headline's: longer set-up and exchanges)
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
    if not papers and not strong and args.cut_fraction is None and ((world == 8 and args.scale == 1.0 and c5 != "0") or c5 == "force"):
        need = 4.5 * headline_s + 20  # 5.7 x the rows, 3.3 x the edges of a products-shaped range
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
    if comm is not None:
        comm.barrier()
    return assemble()
