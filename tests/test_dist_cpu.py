"""CPU suite, world_size 2 and 3 over gloo: vertex-range partitioning + halo exchange
(graphaibench_amd/dist.py) reproduce the single-process result.  The local SpMM is done by the
oracle here (no GPU in this container); the GPU path uses the same Partition / HaloExchanger."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphaibench_amd import dist as gd
        from oracle import binding as orc
        from util import random_graph, rel_err

        rp, ci = random_graph(1500, 12, seed=77, power_law=True)
        g = orc.Graph(rp, ci).add_selfloop()
        n, D = g.nv, 24
        x = np.random.default_rng(5).standard_normal((n, D)).astype(np.float32)
        want_gcn = orc.gcn_aggregate(g, x)
        want_mean_t = orc.sage_d_aggregate(g, x)
        # this rank's rows of the global CSR
        b = gd.partition_bounds(n, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = g.rowptr[lo], g.rowptr[hi]
        rp_l = torch.from_numpy((g.rowptr[lo:hi + 1] - e0).astype(np.int64))
        ci_g = torch.from_numpy(g.colidx[e0:e1].astype(np.int64))
        part = gd.build_partition(rp_l, ci_g, n, rank, world)
        assert part.n_own == hi - lo and part.ne == len(ci_g)
        # owned-/halo-column split keeps every edge, in row order
        own_g = part.colidx_own.long() + lo
        halo_g = part.halo_gids[part.colidx_halo.long()]
        for r in (0, part.n_own // 2, part.n_own - 1):
            seg = ci_g[rp_l[r]:rp_l[r + 1]]
            mine = torch.cat([own_g[part.rowptr_own[r]:part.rowptr_own[r + 1]],
                              halo_g[part.rowptr_halo[r]:part.rowptr_halo[r + 1]]])
            assert torch.equal(torch.sort(mine).values, seg)
        ex = gd.HaloExchanger(part)
        xt = torch.from_numpy(x)
        halo_rows = ex.exchange(xt[lo:hi].contiguous(), D)
        assert torch.equal(halo_rows, xt[part.halo_gids]), "halo rows must equal their owners' rows"
        vd, inv, vd_h, inv_h = gd.global_normalisers(part, ex)
        vd_g = g.vertex_data()
        assert np.array_equal(vd.numpy(), vd_g[lo:hi]) and np.array_equal(vd_h.numpy(), vd_g[part.halo_gids.numpy()])
        # owned-column edges first, then the halo-column edges accumulated onto the same rows ==
        # the rows of the global result (same terms; order: owned then halo)
        import ctypes as C

        def spmm_edge(rp, ci, ew, table, out, nrows):
            orc.lib().orc_spmm_edge(C.c_int64(nrows), orc._p(rp), orc._p(ci), orc._p(ew), C.c_int(D), orc._p(table),
                                    orc._p(out))

        rpo, cio = part.rowptr_own.numpy(), part.colidx_own.numpy().astype(np.uint32)
        rph, cih = part.rowptr_halo.numpy(), part.colidx_halo.numpy().astype(np.uint32)
        xo = np.ascontiguousarray(x[lo:hi])
        xh = np.ascontiguousarray(halo_rows.numpy())
        for vrow, vown, vhalo, want in [(vd.numpy(), vd.numpy(), vd_h.numpy(), want_gcn),
                                        (np.ones(part.n_own, np.float32), inv.numpy(), inv_h.numpy(), want_mean_t)]:
            ew_o = (vrow.repeat(np.diff(rpo)) * vown[cio]).astype(np.float32)
            ew_h = (vrow.repeat(np.diff(rph)) * vhalo[cih]).astype(np.float32)
            a = np.empty((part.n_own, D), np.float32)
            b2 = np.empty((part.n_own, D), np.float32)
            spmm_edge(rpo, cio, ew_o, xo, a, part.n_own)
            spmm_edge(rph, cih, ew_h, xh, b2, part.n_own)
            assert rel_err(a + b2, want[lo:hi]) < 1e-6
        # weight-gradient reduction: sum of per-rank X_own^T G_own == global X^T G
        G = np.random.default_rng(6).standard_normal((n, D)).astype(np.float32)
        dW = torch.from_numpy(x[lo:hi].astype(np.float64).T @ G[lo:hi].astype(np.float64))
        dist.all_reduce(dW)
        assert rel_err(dW.numpy(), x.astype(np.float64).T @ G.astype(np.float64)) < 1e-12
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partition_and_halo_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _complete_worker(rank, world, port, q):
    """a DENSE random graph (every range needs nearly all rows of every other): the halo of a peer above the threshold is taken
    whole (dist.split_by_owner, round 5), so that peer's send list is its full row range -- one run of consecutive rows"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphaibench_amd import dist as gd
        from oracle import binding as orc
        from util import random_graph

        rp, ci = random_graph(600, 60, seed=5, power_law=False)
        g = orc.Graph(rp, ci).add_selfloop()
        n, D = g.nv, 8
        b = gd.partition_bounds(n, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = g.rowptr[lo], g.rowptr[hi]
        rp_l = torch.from_numpy((g.rowptr[lo:hi + 1] - e0).astype(np.int64))
        ci_g = torch.from_numpy(g.colidx[e0:e1].astype(np.int64))
        needed = torch.unique(ci_g[(ci_g < lo) | (ci_g >= hi)])
        part = gd.build_partition(rp_l, ci_g, n, rank, world)
        # every peer range is complete: the halo is ALL remote vertices (a superset of what the rows read), ascending
        assert part.n_halo == n - (hi - lo) and part.n_halo >= needed.numel() > 0.9 * part.n_halo
        assert torch.equal(part.halo_gids, torch.cat([torch.arange(0, lo), torch.arange(hi, n)]))
        off = 0
        for qk in range(world):  # the list for every peer: my rows 0 .. n_own - 1 in order
            cnt = part.send_counts[qk]
            assert cnt == (0 if qk == rank else hi - lo)
            assert torch.equal(part.send_idx[off:off + cnt], torch.arange(cnt))
            off += cnt
        x = torch.from_numpy(np.random.default_rng(5).standard_normal((n, D)).astype(np.float32))
        halo_rows = gd.HaloExchanger(part).exchange(x[lo:hi].contiguous(), D)
        assert torch.equal(halo_rows, x[part.halo_gids])
        # the column split still holds every edge
        own_g = part.colidx_own.long() + lo
        halo_g = part.halo_gids[part.colidx_halo.long()]
        assert torch.equal(torch.sort(torch.cat([own_g, halo_g])).values, torch.sort(ci_g).values)
        # ... and with the upgrade off the halo is exactly what the rows read
        os.environ["GAIB_COMPLETE_HALO"] = "0"
        part0 = gd.build_partition(rp_l, ci_g, n, rank, world)
        assert torch.equal(part0.halo_gids, needed)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_complete_halo_makes_send_lists_one_run_of_rows(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_complete_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_block_rows_are_globally_symmetric():
    """the multi-GPU bench generator: every rank's rows of one symmetric global graph"""
    from graphaibench_amd import synth

    world = 3
    blocks = [synth.block_rows("tiny", r, world, seed=3, cut_fraction=0.3, device="cpu", scale=0.05, selfloops=True)
              for r in range(world)]
    n = blocks[0].n_global
    pairs = set()
    for r, b in enumerate(blocks):
        rp, ci = b.rowptr.numpy(), b.colidx_global.numpy()
        rows = np.repeat(np.arange(b.n_local), np.diff(rp)) + r * b.n_local
        for u, v in zip(rows.tolist(), ci.tolist()):
            pairs.add((u, v))
        # rows sorted, no duplicates
        for i in range(b.n_local):
            seg = ci[rp[i]:rp[i + 1]]
            assert np.all(np.diff(seg) > 0)
    assert all((v, u) in pairs for (u, v) in pairs)
    assert all((i, i) in pairs for i in range(n))
    cross = sum(1 for (u, v) in pairs if u // blocks[0].n_local != v // blocks[0].n_local)
    assert 0.15 < cross / len(pairs) < 0.40


# ---- the partition itself against the REAL reference partitioner -------------------------------------------------
def _check_against_reference_parts(rp, ci, parts, ref_parts):
    """ref_parts[i]: owned range, local -> global id map and local CSR of PartitionedGraph::edgecut_induced_partition1D
    (src/partitioner/graph_partition.cc:128-178).  Ours: owned rows only, split by column owner."""
    from graphaibench_amd import dist as gd

    n = len(rp) - 1
    bounds = gd.partition_bounds(n, parts)
    assert len(ref_parts) == parts
    for p, r in enumerate(ref_parts):
        lo, hi = bounds[p], bounds[p + 1]
        assert (int(r["range"][0]), int(r["range"][1])) == (lo, hi)
        rpl = torch.from_numpy((rp[lo:hi + 1] - rp[lo]).astype(np.int64))
        cg = torch.from_numpy(ci[rp[lo]:rp[hi]].astype(np.int64))
        rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(rpl, cg, lo, hi)
        idx_map = r["idx_map"].astype(np.int64)
        # the subgraph's vertex set is exactly owned + halo, in ascending global order
        assert np.array_equal(np.union1d(np.arange(lo, hi), halo.numpy()), idx_map)
        assert np.array_equal(deg.numpy(), np.diff(rp[lo:hi + 1]))
        # every owned row has the reference's neighbours (the reference numbers them by position in idx_map)
        pos = np.searchsorted(idx_map, np.arange(lo, hi))
        for k in range(hi - lo):
            lr = pos[k]
            want = idx_map[r["colidx"][r["rowptr"][lr]:r["rowptr"][lr + 1]]]
            mine = np.concatenate([ci_own[rp_own[k]:rp_own[k + 1]].numpy().astype(np.int64) + lo,
                                   halo.numpy()[ci_halo[rp_halo[k]:rp_halo[k + 1]].numpy()]])
            assert np.array_equal(np.sort(mine), want)


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_partition_matches_reference_golden(parts):
    g = np.load(Path(__file__).resolve().parent / "golden" / f"partition_p{parts}.npz")
    nv, deg, seed = (int(v) for v in g["graph"])
    from util import random_graph
    rp, ci = random_graph(nv, deg, seed=seed, power_law=True)
    ref_parts = [dict(range=g[f"range{i}"], idx_map=g[f"idx_map{i}"], rowptr=g[f"rowptr{i}"], colidx=g[f"colidx{i}"])
                 for i in range(parts)]
    _check_against_reference_parts(rp, ci, parts, ref_parts)


@pytest.mark.parametrize("nv,parts", [(5003, 4), (4096, 8), (777, 5)])
def test_partition_matches_live_reference(nv, parts):
    from oracle import binding as orc
    from util import random_graph
    rp, ci = random_graph(nv, 9, seed=nv, power_law=True)
    ref = orc.ref_partition(rp, ci, parts)
    if ref is None:
        pytest.skip("oracle/_ref/libref_partition.so is only built where /root/reference exists")
    ref_parts = [dict(range=np.array([r["begin"], r["end"]]), idx_map=r["idx_map"], rowptr=r["rowptr"],
                      colidx=r["colidx"]) for r in ref]
    _check_against_reference_parts(rp, ci, parts, ref_parts)


# ---- the host C++ partition builder of the multi-rank trainer (include/gnn/partition.h) --------------------------
@pytest.mark.parametrize("complete", [0.0, 0.9, None])
@pytest.mark.parametrize("nv,world", [(1000, 2), (777, 3), (4096, 8), (50, 8)])
def test_cpp_partition_equals_dist_py(nv, world, complete, monkeypatch):
    """build_vertex_range_partition (host C++, no communication: every rank derives its share from the global CSR)
    == dist.py's split (pinned against the reference partitioner above), and the ranks' send / receive lists mirror
    each other.  complete = 0.9 (the default, round 5): a peer range needed to at least 90 % is taken whole, by both builders
    alike; 0: the plain induced halo; None: the threshold from the link rate and the number of ranks."""
    from graphaibench_amd import dist as gd, layers as L
    from util import random_graph

    if complete is None:  # the rule: 1 - 2 (world - 1) link / 5000, the same figure in both builders
        monkeypatch.delenv("GAIB_COMPLETE_HALO", raising=False)
        monkeypatch.setenv("GAIB_LINK_GBS", "120")
        assert gd.complete_halo_threshold(world) == pytest.approx(max(0.5, 1 - 2 * (world - 1) * 120 / 5000))
    else:
        monkeypatch.setenv("GAIB_COMPLETE_HALO", str(complete))
    rp, ci = random_graph(nv, 7, seed=nv + world, power_law=True)
    bounds = gd.partition_bounds(nv, world)
    parts = [L.HostPartition(rp, ci, r, world) for r in range(world)]
    for r, P in enumerate(parts):
        lo, hi = bounds[r], bounds[r + 1]
        assert (P.lo, P.hi) == (lo, hi)
        rpl = torch.from_numpy((rp[lo:hi + 1] - rp[lo]).astype(np.int64))
        cg = torch.from_numpy(ci[rp[lo]:rp[hi]].astype(np.int64))
        rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(rpl, cg, lo, hi, bounds, gd.complete_halo_threshold(world))
        if complete == 0.0:  # exactly the columns the rows read
            assert np.array_equal(halo.numpy(), np.unique(cg.numpy()[(cg.numpy() < lo) | (cg.numpy() >= hi)]))
        assert np.array_equal(P.rowptr_own, rp_own.numpy()) and np.array_equal(P.rowptr_halo, rp_halo.numpy())
        assert np.array_equal(P.colidx_own, ci_own.numpy().view(np.uint32))
        assert np.array_equal(P.colidx_halo, ci_halo.numpy().view(np.uint32))
        assert np.array_equal(P.halo_gids, halo.numpy()) and np.array_equal(P.degree, deg.numpy())
        assert np.array_equal(P.halo_degree, np.diff(rp)[P.halo_gids])
        owner = np.searchsorted(np.asarray(bounds), P.halo_gids, side="right") - 1
        assert np.array_equal(P.recv_counts, np.bincount(owner, minlength=world))
        assert P.recv_counts[r] == 0 and P.send_counts[r] == 0
    for p in range(world):
        off = np.concatenate([[0], np.cumsum(parts[p].send_counts)])
        for q in range(world):
            assert parts[p].send_counts[q] == parts[q].recv_counts[p]
            # what p packs for q is exactly the segment of q's halo list that p owns, in q's order
            mine = parts[p].send_idx[off[q]:off[q + 1]] + parts[p].lo
            theirs = parts[q].halo_gids[(parts[q].halo_gids >= parts[p].lo) & (parts[q].halo_gids < parts[p].hi)]
            assert np.array_equal(mine, theirs)


@pytest.mark.parametrize("nv,world", [(500, 3), (1200, 2)])
def test_cpp_partition_gat_structures(nv, world):
    """build_gat_structures: the rows over ONE [owned | halo] column space keep the global CSR's edge order, and the
    transposed structure + tperm is exactly its transpose (what replaces the reverse-edge permutation on a partition)"""
    from graphaibench_amd import layers as L
    from util import random_graph

    rp, ci = random_graph(nv, 6, seed=nv, power_law=True)
    for r in range(world):
        P = L.HostPartition(rp, ci, r, world, gat=True)
        n_own = P.hi - P.lo
        nc = n_own + len(P.halo_gids)
        glob = np.concatenate([np.arange(P.lo, P.hi), P.halo_gids])
        assert np.array_equal(np.diff(P.rowptr_full), P.degree)
        assert np.array_equal(glob[P.colidx_full], ci[rp[P.lo]:rp[P.hi]])  # same edges, same order, local ids
        rows = np.repeat(np.arange(n_own), np.diff(P.rowptr_full))
        assert len(P.rowptr_t) == nc + 1 and P.rowptr_t[-1] == len(P.colidx_full)
        assert sorted(P.tperm.tolist()) == list(range(len(P.colidx_full)))  # a permutation of the edges
        tv = np.repeat(np.arange(nc), np.diff(P.rowptr_t))
        assert np.array_equal(P.colidx_full[P.tperm], tv) and np.array_equal(rows[P.tperm], P.colidx_t)
        for v in range(nc):  # rows of a column ascending (stable counting sort)
            seg = P.colidx_t[P.rowptr_t[v]:P.rowptr_t[v + 1]]
            assert np.all(np.diff(seg.astype(np.int64)) > 0)


# ---- the graph generator of bench.py's N > 1 legs (synth.block_rows): every rank generates ITS rows, from seeds shared
# by the two owners of a cross block -- the global graph must come out symmetric whoever assembles it ---------------------
@pytest.mark.parametrize("shape,world,cut", [("ogbn-papers100M/8", 8, 0.1), ("ogbn-papers100M/8", 8, 0.875), ("ogbn-products", 4, 0.1),
                                             ("ogbn-products", 2, 0.5)])
def test_block_rows_ranks_agree_on_one_symmetric_graph(shape, world, cut):
    """BASELINE config 5's graph (8 vertex ranges of the papers100M shape) and the weak-scaling products graph at toy scale, on
    the CPU: the ranks' row blocks join into ONE structurally symmetric graph with sorted, duplicate-free rows and a self
    loop per vertex; the share of edges that leave a range is the requested cut; and the joined graph partitions, rank by
    rank, into exactly the rows each rank generated (dist.split_by_owner keeps every edge)."""
    from graphaibench_amd import dist as gd, synth

    scale = 3e-4 if shape.startswith("ogbn-papers") else 2e-3
    blocks = [synth.block_rows(shape, r, world, seed=42, cut_fraction=cut, device="cpu", scale=scale, selfloops=True)
              for r in range(world)]
    nv_p = blocks[0].n_local
    n = nv_p * world
    assert all(b.n_local == nv_p and b.n_global == n for b in blocks)
    rp = np.concatenate([[0]] + [b.rowptr[1:].numpy() + sum(int(c.rowptr[-1]) for c in blocks[:r]) for r, b in enumerate(blocks)])
    ci = np.concatenate([b.colidx_global.numpy() for b in blocks])
    rows = np.repeat(np.arange(n), np.diff(rp))
    key = rows * n + ci
    assert np.all(np.diff(key) > 0)                      # rows sorted, no duplicates
    assert np.array_equal(np.sort(ci * n + rows), key)   # symmetric: the transpose is the same edge set
    assert np.all(ci[rp[:-1] + np.searchsorted(key, np.arange(n) * n + np.arange(n)) - rp[:-1]] == np.arange(n))  # self loops
    cross = (rows // nv_p) != (ci // nv_p)
    share = cross.sum() / (len(ci) - n)
    assert abs(share - cut) < 0.25 * cut + 0.02, (share, cut)  # (duplicates removed inside a range shift it a little)
    for r, b in enumerate(blocks):
        lo, hi = r * nv_p, (r + 1) * nv_p
        rp_own, ci_own, rp_halo, ci_halo, halo, deg = gd.split_by_owner(b.rowptr, b.colidx_global, lo, hi)
        assert np.array_equal(deg.numpy(), np.diff(rp)[lo:hi])
        assert int(rp_own[-1]) + int(rp_halo[-1]) == int(b.rowptr[-1])
        assert np.all((halo.numpy() < lo) | (halo.numpy() >= hi)) and np.all(np.diff(halo.numpy()) > 0)
        owned_cols = ci[rp[lo]:rp[hi]]
        assert (ci_own.numel() == ((owned_cols >= lo) & (owned_cols < hi)).sum())


def _worker8(rank, world, port, q):
    """one of EIGHT ranks of config 5's set-up path over gloo: this rank's rows of the papers100M/8-shaped block graph,
    build_partition (the all-to-all of counts and halo ids), one halo exchange of feature rows, the normalisers"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphaibench_amd import dist as gd, synth

        torch.set_num_threads(1)
        rows = synth.block_rows("ogbn-papers100M/8", rank, world, seed=42, cut_fraction=0.875, device="cpu", scale=1e-4,
                                selfloops=True)
        part = gd.build_partition(rows.rowptr, rows.colidx_global, rows.n_global, rank, world)
        nv = part.n_own
        assert (part.lo, part.hi) == (rank * nv, (rank + 1) * nv) and part.recv_counts[rank] == 0 == part.send_counts[rank]
        # every rank's feature rows are a function of the GLOBAL row id: what arrives must be the owners' rows
        D = 8
        gid = torch.arange(part.lo, part.hi, dtype=torch.float32).reshape(-1, 1)
        x = (gid * 3.0 + torch.arange(D, dtype=torch.float32)).contiguous()
        ex = gd.HaloExchanger(part)
        got = ex.exchange(x, D)
        want = part.halo_gids.to(torch.float32).reshape(-1, 1) * 3.0 + torch.arange(D, dtype=torch.float32)
        assert torch.equal(got, want)
        vd, inv, vd_h, inv_h = gd.global_normalisers(part, ex)
        # a halo vertex's normaliser comes from its owner's FULL degree: gather all degrees and compare
        degs = [torch.empty(nv, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(degs, part.degree)
        deg_all = torch.cat(degs).to(torch.float32)
        assert torch.equal(inv_h, (1.0 / deg_all[part.halo_gids].double()).float())
        send_total = torch.tensor([float(sum(part.send_counts)), float(sum(part.recv_counts))])
        dist.all_reduce(send_total)
        assert send_total[0] == send_total[1]  # every row sent is a row received
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_set_up_config5_over_gloo():
    """a rehearsal of the 8-rank control plane (the GPU box has one GPU; the 8-GPU lease is the driver's): eight gloo ranks on
    the CPU build config 5's partition from their own rows and exchange halo rows, random-order end (every rank talks to
    every other)"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 700)
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
