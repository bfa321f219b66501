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

import time
from dataclasses import dataclass, field

import torch
import torch.distributed as dist


def partition_bounds(n: int, world: int):
    per = -(-n // world)
    return [min(p * per, n) for p in range(world + 1)]


def _all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_counts, in_counts, group=None):
    """all-to-all of row blocks; falls back to pairwise send/recv where the backend lacks it.
    Device tensors over a gloo group (2 processes sharing one GPU in the tests) go through host."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        o_h = torch.empty(out.shape, dtype=out.dtype)
        _all_to_all_rows(o_h, inp.cpu(), out_counts, in_counts, group)
        out.copy_(o_h)
        return
    try:
        dist.all_to_all_single(out, inp, output_split_sizes=list(out_counts), input_split_sizes=list(in_counts),
                               group=group)
        return
    except (RuntimeError, NotImplementedError):
        pass
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    oo = [0]
    for c in out_counts:
        oo.append(oo[-1] + c)
    io = [0]
    for c in in_counts:
        io.append(io[-1] + c)
    out[oo[rank]:oo[rank + 1]] = inp[io[rank]:io[rank + 1]]
    reqs = []
    for q in range(world):
        if q == rank:
            continue
        if in_counts[q]:
            reqs.append(dist.isend(inp[io[q]:io[q + 1]].contiguous(), q, group=group))
    for q in range(world):
        if q == rank or not out_counts[q]:
            continue
        buf = torch.empty_like(out[oo[q]:oo[q + 1]])
        dist.recv(buf, q, group=group)
        out[oo[q]:oo[q + 1]] = buf
    for r in reqs:
        r.wait()


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
        self._sync_fallback = False

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
            if sendbuf.is_cuda and dist.get_backend(p.group) != "gloo" and not self._sync_fallback:
                try:
                    work = dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=list(p.recv_counts),
                                                  input_split_sizes=list(p.send_counts), group=p.group, async_op=True)
                except (RuntimeError, NotImplementedError):
                    # a backend without (async) all-to-all(v): pairwise exchange from here on (no overlap, still correct)
                    self._sync_fallback = True
                    _all_to_all_rows(recvbuf, sendbuf, p.recv_counts, p.send_counts, p.group)
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

    def __init__(self, ctx, part: Partition):
        from . import capi, layers as L

        self.ctx, self.part = ctx, part
        self._capi = capi
        dev = f"cuda:{ctx.device}"

        def gather(idx, src, D, out):
            ptr = src if isinstance(src, int) else src.data_ptr()
            capi._check(ctx.lib.gaib_gather_rows(ctx.h, idx.numel(), idx.data_ptr(), D, ptr, out.data_ptr()),
                        "gaib_gather_rows")

        self.ex = HaloExchanger(part, gather_rows=gather)
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
            self.lgraph.set_halo(self.g_halo, self._begin, self._end)
        self._dev = dev

    def _begin(self, length: int, d_in: int) -> None:
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


def allreduce_layer_grads(ctx, layer, which_list, shape, group=None):
    """sum the weight gradients of one layer over ranks (one fused all-reduce)."""
    from . import capi

    n = shape[0] * shape[1]
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


def bench_gcn_layer(ctx, args, rank: int, world: int, D: int, log):
    """bench.py's N > 1 leg: weak scaling, every rank owns a products-shaped vertex range of one
    global Chung-Lu graph (synth.block_rows); GCN hidden layer D -> D forward + backward per step,
    halo exchange before each of the 2 SpMM, one all-reduce of dW per step."""
    from . import layers as L, synth

    cut = 0.1 if args.cut_fraction is None else args.cut_fraction
    t0 = time.time()
    rows = synth.block_rows("ogbn-products", rank, world, seed=42, cut_fraction=cut, device="cuda", scale=args.scale,
                            selfloops=True)  # GCN aggregates over A + I (net.cpp:96)
    part = build_partition(rows.rowptr, rows.colidx_global, rows.n_global, rank, world)
    dg = DistLayerGraph(ctx, part)
    torch.cuda.synchronize()
    log(f"[bench r{rank}] rows [{part.lo},{part.hi}) ne={part.ne} (own-column {part.colidx_own.numel()}) "
        f"halo rows={part.n_halo} "
        f"send rows={part.send_idx.numel()} setup {time.time()-t0:.1f}s")
    nv = part.n_own
    torch.manual_seed(43 + rank)
    layer = L.Layer(L.GCN, 1, nv, D, D, dg.lgraph, act=True, lr=0.01)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda"))
    layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    feat_out = torch.empty(nv, D, device="cuda")
    grad_out = torch.empty(nv, D, device="cuda")

    def step():
        layer.forward(feat_out)
        layer.backward(feat_out, grad_out)
        allreduce_layer_grads(ctx, layer, [L.W_NEIGH_GRAD], (D, D))

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
    ctx.prof_reset()
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
    sendbuf, _ = dg.ex._bufs(D, torch.float32, feat.device)
    t0 = time.perf_counter()
    for _ in range(reps):
        if part.send_idx.numel():
            dg.ex.gather_rows(part.send_idx, feat, D, sendbuf)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) / reps * 1e3
    # max time over ranks, total edges over ranks
    rdev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    t = torch.tensor([elapsed, exch_ms, pack_ms], dtype=torch.float64, device=rdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    exch_ms, pack_ms = float(t[1]), float(t[2])
    e = torch.tensor([float(part.ne), float(part.n_halo), float(bytes_timed)], dtype=torch.float64, device=rdev)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    elapsed = float(t[0])
    total_edges = float(e[0])
    # dominant kernel (rank 0's view) = the pass over the owned-column edges of each aggregation: with halo
    # edges that is spmm_w64_kernel (the halo half then carries the dense product), without them the fused
    # kernel does everything, as in the single-GPU bench
    st_own = ctx.graph_stats(dg.lgraph.device_graph())
    e_light = part.colidx_own.numel() - st_own["heavy_edges"]
    if part.colidx_halo.numel() > 0:
        kernel_name = "spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer> over the owned-column edges (rank 0)"
        alg_bytes = e_light * (4 * D + 8) + (nv - st_own["n_heavy"]) * 4 * D + (nv + 1) * 8
        n_dom, ms_dom = n_light, ms_light
    else:
        kernel_name = "spmm_gemm_kernel<VEC=2,edge-weights,U=16,buffer> (aggregation + MFMA dense product, rank 0)"
        alg_bytes = e_light * (4 * D + 8) + int(1.5 * nv * 4 * D) + (nv + 1) * 8
        n_dom, ms_dom = n_fused, ms_fused
    avg_ms = ms_dom / max(n_dom, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "metric": "GCN-layer fwd+bwd aggregated edges/sec",
        "value": 2 * total_edges * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "block Chung-Lu graph, one ogbn-products-shaped vertex range per GPU (seed 42), "
                        "GCN hidden layer 128->128 fwd+bwd, halo all-to-all before each SpMM + dW all-reduce",
            "nv_per_gpu": nv, "ne_total_with_selfloops": int(total_edges), "D": D, "scale": args.scale,
            "cut_fraction": cut, "halo_rows_total": int(e[1]),
            "halo_bytes_per_step_total": float(e[2]) / args.steps,
            # slowest rank, measured after the timed region: one exchange on its own (pack + all-to-all + wait), the
            # pack alone, and the owned-edge aggregation kernels of one step that run while the two exchanges fly
            "halo_exchange_standalone_ms": exch_ms, "halo_pack_ms": pack_ms,
            "owned_edge_spmm_ms_per_step": ms_light / args.steps,
            "parallelism": f"vertex-range x{world}",
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel_name,
            "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
            "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches": n_dom,
        },
    }
