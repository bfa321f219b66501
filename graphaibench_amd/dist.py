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
    """one rank's share of a vertex-range partitioned graph"""
    rank: int
    world: int
    n_global: int
    lo: int
    hi: int
    rowptr: torch.Tensor        # int64 [n_own+1]
    colidx: torch.Tensor        # int32 [ne_local], LOCAL ids: owned -> [0,n_own), halo -> n_own + k
    halo_gids: torch.Tensor     # int64 [n_halo] global ids, ascending (hence grouped by owner)
    recv_counts: list           # halo rows owned by rank q (contiguous segments of halo_gids)
    send_idx: torch.Tensor      # int64 [total_send] local row ids to ship, grouped by destination
    send_counts: list
    group: object = None
    _sendbuf: dict = field(default_factory=dict)

    @property
    def n_own(self) -> int:
        return self.hi - self.lo

    @property
    def n_halo(self) -> int:
        return int(self.halo_gids.numel())

    @property
    def n_table(self) -> int:
        return self.n_own + self.n_halo

    @property
    def ne(self) -> int:
        return int(self.colidx.numel())


def build_partition(rowptr_local: torch.Tensor, colidx_global: torch.Tensor, n_global: int, rank: int, world: int,
                    group=None) -> Partition:
    """rowptr_local/colidx_global: this rank's rows [lo,hi) of the global CSR (global column ids),
    on the compute device.  Collective: every rank calls it."""
    device = colidx_global.device
    bounds = partition_bounds(n_global, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_own = hi - lo
    assert rowptr_local.numel() == n_own + 1
    cols = colidx_global.to(torch.int64)
    own = (cols >= lo) & (cols < hi)
    halo = torch.unique(cols[~own])  # sorted
    local = torch.where(own, cols - lo, n_own + torch.searchsorted(halo, cols))
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
    return Partition(rank, world, n_global, lo, hi, rowptr_local.to(torch.int64).contiguous(),
                     local.to(torch.int32).contiguous(), halo, rc, send_idx.contiguous(), sc, group)


class HaloExchanger:
    """fills rows [n_own, n_own+n_halo) of a feature table from the owners of those vertices."""

    def __init__(self, part: Partition, gather_rows=None):
        self.p = part
        # gather_rows(idx[int64], src[n_own x D], out[k x D]) -- HIP kernel on GPU, index_select on CPU
        self.gather_rows = gather_rows or (lambda idx, src, out: torch.index_select(src, 0, idx, out=out))
        self.bytes_sent = 0
        self.seconds = 0.0

    def exchange(self, table: torch.Tensor):
        """table: [n_own + n_halo, D] contiguous; rows [0, n_own) are this rank's current values."""
        p = self.p
        D = table.shape[1]
        key = (D, table.dtype, table.device)
        if key not in p._sendbuf:
            p._sendbuf[key] = torch.empty(max(p.send_idx.numel(), 1), D, dtype=table.dtype, device=table.device)
        sendbuf = p._sendbuf[key][:p.send_idx.numel()]
        if p.send_idx.numel():
            self.gather_rows(p.send_idx, table[:p.n_own], sendbuf)
        recv = table[p.n_own:]
        _all_to_all_rows(recv, sendbuf, p.recv_counts, p.send_counts, p.group)
        self.bytes_sent += sendbuf.numel() * sendbuf.element_size()
        return table


def global_normalisers(part: Partition, ex: HaloExchanger):
    """(vdata [n_table], inv_deg [n_table]) with the GLOBAL degree of every owned and halo vertex.
    Owned rows are complete in a row partition, so their local degree is the global one."""
    deg = (part.rowptr[1:] - part.rowptr[:-1]).to(torch.float32)
    dev = deg.device
    vd = torch.zeros(part.n_table, 1, dtype=torch.float32, device=dev)
    inv = torch.zeros(part.n_table, 1, dtype=torch.float32, device=dev)
    # deg^-1/2 with 0 for isolated vertices (lgraph.cpp:22-34); 1/deg in double then narrowed
    # (sage_aggregator.cpp:18).  float64 arithmetic reproduces both roundings.
    d64 = deg.to(torch.float64)
    s = torch.sqrt(deg).to(torch.float64)
    vd[:part.n_own, 0] = torch.where(s == 0, torch.zeros_like(s), 1.0 / s).to(torch.float32)
    inv[:part.n_own, 0] = (1.0 / d64).to(torch.float32)
    ex.exchange(vd)
    ex.exchange(inv)
    return vd[:, 0].contiguous(), inv[:, 0].contiguous()


# ---- GPU layer driver ----------------------------------------------------------------------------
class DistLayerGraph:
    """A LearningGraph over this rank's rectangular local CSR whose aggregations run the halo
    exchange first (LearningGraph::set_halo_hook).  The C++ layer code is the single-GPU one."""

    def __init__(self, ctx, part: Partition):
        from . import capi, layers as L

        self.ctx, self.part = ctx, part
        self.ex = HaloExchanger(part, gather_rows=lambda idx, src, out: ctx.gather_rows(idx, src, out))
        g = ctx.graph(part.rowptr, part.colidx, ncols=part.n_table)
        vd, inv = global_normalisers(part, self.ex)
        g.set_vertex_norm(vd[:part.n_own].contiguous(), vd, inv)
        self.lgraph = L.LGraph.adopt(g)
        self.tables = {}
        self._capi = capi
        self.lgraph.set_halo_hook(self._hook)

    def _hook(self, length: int, d_in: int) -> int:
        p = self.part
        t = self.tables.get(length)
        if t is None:
            t = torch.empty(p.n_table, length, dtype=torch.float32, device=f"cuda:{self.ctx.device}")
            self.tables[length] = t
        # owned rows into the head of the table, then the halo rows from their owners
        self._capi._check(self.ctx.lib.gaib_memcpy_d2d(self.ctx.h, t.data_ptr(), d_in, p.n_own * length * 4),
                          "gaib_memcpy_d2d")
        if p.world > 1:
            self.ex.exchange(t)
        return t.data_ptr()


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
    log(f"[bench r{rank}] rows [{part.lo},{part.hi}) ne={part.ne} halo rows={part.n_halo} "
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
    ctx.prof_reset()
    # max time over ranks, total edges over ranks
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    e = torch.tensor([float(part.ne), float(part.n_halo), float(dg.ex.bytes_sent)], dtype=torch.float64, device="cuda")
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    elapsed = float(t[0])
    total_edges = float(e[0])
    stats = ctx.graph_stats(dg.lgraph.device_graph())
    e_light = part.ne - stats["heavy_edges"]
    alg_bytes = e_light * (4 * D + 8) + (nv - stats["n_heavy"]) * 4 * D + (nv + 1) * 8
    avg_ms = ms_light / max(n_light, 1)
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
            "parallelism": f"vertex-range x{world}",
        },
        "roofline": {
            "bound": "hbm", "kernel": "spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer> (rank 0)",
            "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
            "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches": n_light,
        },
    }
