"""GPU suite: the collectives behind the C ABI (include/gaib.h: gaib_comm_* / gaib_halo_* / gaib_allreduce_*) and the
host C++ partition builder + layers on top of them (SURVEY.md 8b "halo_exchange / allreduce", 8e).

One GPU is all this box has, so:
  * the IPC transport (peer-to-peer pull through hipIpc handles) runs for real with 2 and 3 PROCESSES sharing cuda:0:
    partition -> LearningGraph with a halo plan -> C++ GCN / SAGE layer forward + backward + optimizer step, against the
    oracle's GLOBAL result;
  * the RCCL transport runs on the system's library with one rank (init, all-reduce, an exchange without peers): RCCL
    refuses two ranks on one device;
  * the RCCL BRANCH of comm.hip (grouped ncclSend / ncclRecv with per-peer offsets and counts, the reverse exchange,
    the all-reduces on the communication stream) runs with 2 and 3 ranks bound to tests/fake_rccl -- a strict double
    that moves the same calls' bytes between processes sharing cuda:0 and refuses every count / datatype / peer /
    address-range mismatch the real library would hang or fault on.  It proves the call pattern, not RCCL or xGMI;
  * a rank whose peer never arrives returns GAIB_ERR_COMM within the deadline instead of hanging.
"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


FAKE_RCCL = ROOT / "tests" / "fake_rccl" / "librccl_fake.so"


def _transport(name, capi):
    """"ipc" | "rccl" (the system's library: one rank per device) | "fake-rccl": the RCCL branch of comm.hip bound to
    tests/fake_rccl's strict double, which carries the same ncclSend / ncclRecv / ncclAllReduce calls between processes
    that share one GPU and fails on any count, datatype, peer or address-range mismatch"""
    if name == "fake-rccl":
        assert FAKE_RCCL.exists(), f"{FAKE_RCCL} not built (python -m graphaibench_amd.build)"
        os.environ["GAIB_RCCL_LIB"] = str(FAKE_RCCL)
        return capi.COMM_RCCL
    return capi.COMM_IPC if name == "ipc" else capi.COMM_RCCL


def _id_via_file(path, rank, transport, capi):
    """rank 0 draws the id, the others read it (what the C++ trainer does with GAIB_COMM_ID_FILE)"""
    if rank == 0:
        uid = capi.comm_unique_id(transport)
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < 60:
        if os.path.exists(path) and os.path.getsize(path) == 128:
            return open(path, "rb").read()
        time.sleep(0.005)
    raise RuntimeError("no communicator id")


def _worker(rank, world, idfile, q, arch, transport_name):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    try:
        from graphaibench_amd import capi, layers as L
        from oracle import binding as orc
        from util import LONG_SUM_FLOOR, assert_close, random_graph

        transport = _transport(transport_name, capi)
        ctx = L.init(0)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        # raw collectives first
        t = torch.full((70000,), float(rank + 1), device="cuda")  # > one 64 K staging chunk
        comm.allreduce(t)
        assert torch.all(t == world * (world + 1) / 2)
        assert comm.allreduce_host([1.0, float(rank)]) == [float(world), world * (world - 1) / 2]
        L.set_comm(comm)

        rp, ci = random_graph(3000, 16, seed=13, power_law=True, hub_deg=1500)
        g = orc.Graph(rp, ci)
        if arch == "gcn":
            g = g.add_selfloop()  # (SAGE aggregates over A, net.cpp:96)
        n, D = g.nv, 128
        x = np.random.default_rng(5).standard_normal((n, D)).astype(np.float32)
        gin = np.random.default_rng(6).standard_normal((n, D)).astype(np.float32)
        lo_ = (orc.GCNLayer if arch == "gcn" else orc.SAGELayer)(1, g, D, D, True)
        want = lo_.forward(x)
        want_go = lo_.backward(gin.copy())
        part = L.HostPartition(g.rowptr, g.colidx, rank, world)
        lo, hi = part.lo, part.hi
        lg = part.make_graph(comm)
        layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, hi - lo, D, D, lg, True)
        W0 = layer.tensor(L.W_NEIGH, (D, D)).clone()
        layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
        out = torch.empty(hi - lo, D, device="cuda")
        layer.forward(out)
        assert_close(out.cpu().numpy(), want[lo:hi], "forward", floor=LONG_SUM_FLOOR)
        out.copy_(torch.from_numpy(want[lo:hi]).cuda())  # identical relu masks (see bench.py parity_record)
        layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
        grad_out = torch.empty(hi - lo, D, device="cuda")
        layer.backward(out, grad_out)
        assert_close(grad_out.cpu().numpy(), want_go[lo:hi], "grad_out", floor=LONG_SUM_FLOOR)
        # the optimizer step sums the gradient over the ranks first (gpu_context::set_comm) and then updates:
        # afterwards the gradient buffer holds the GLOBAL gradient and every rank the same new weights
        opt = L.adam(0.01)
        layer.update_weight(opt)
        want_wg = lo_.W_grad if arch == "gcn" else lo_.W_neigh_grad
        assert_close(layer.tensor(L.W_NEIGH_GRAD, (D, D)).cpu().numpy(), want_wg, "W_grad", floor=LONG_SUM_FLOOR)
        if arch == "sage":
            assert_close(layer.tensor(L.W_SELF_GRAD, (D, D)).cpu().numpy(), lo_.W_self_grad, "W_self_grad",
                         floor=LONG_SUM_FLOOR)
        o_opt = orc.Adam(0.01)
        W_want = W0.cpu().numpy().copy()
        o_opt.update("w", want_wg, W_want)
        W_new = layer.tensor(L.W_NEIGH, (D, D))
        # the first Adam step is lr * g / sqrt(g^2 + 1e-8): +-lr wherever |g| >> 1e-4, and ill-conditioned in g where a
        # gradient entry is within rounding of zero -- compare where the step is well defined, bound the rest by lr
        sure = np.abs(want_wg) > 1e-4 * np.abs(want_wg).max()
        assert sure.mean() > 0.99
        assert_close(W_new.cpu().numpy()[sure], W_want[sure], "W after Adam")
        assert np.abs(W_new.cpu().numpy() - W_want).max() <= 2.1 * 0.01
        # bit-identical replicas: compare a checksum of the new weights across ranks
        s = W_new.double().sum().item()
        tot = comm.allreduce_host([s])[0]
        assert abs(tot - world * s) <= 1e-9 * abs(tot), (tot, s)
        comm.barrier()
        # a partition's LearningGraph owns its halo graph and exchange plan: dealloc gives them back (the ranks share
        # the device, so every reading of the free memory sits between two barriers)
        def free_now():
            L.sync()
            comm.barrier()
            f = torch.cuda.mem_get_info()[0]
            comm.barrier()
            return f

        def build_use_close():
            lg2 = part.make_graph(comm)
            l2 = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, hi - lo, D, D, lg2, True)
            l2.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
            l2.forward(out)
            L.sync()
            l2.close()
            lg2.close()

        build_use_close()
        base = free_now()
        for _ in range(3):
            build_use_close()
        def lost_now():  # the same number on every rank (so that all of them leave the loop below together)
            return comm.allreduce_host([float(base - free_now())])[0] / world

        lost, tries = lost_now(), 0
        while lost > (4 << 20) * world and tries < 20:  # (the driver hands a peer's freed memory back a little later)
            time.sleep(0.25)
            lost, tries = lost_now(), tries + 1
        assert lost <= (4 << 20) * world, f"{lost / 2**20:.1f} MiB of the device not returned by {world} ranks after {tries} polls"
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def _mode_worker(rank, world, idfile, q, arch, mode, din, dout, transport_name="ipc"):
    """one GCN / SAGE layer din -> dout on a vertex-range partition in a GIVEN row-class mode (LearningGraph::
    partition_mode: split = round 3's column split over all rows, classes = interior rows in one pass + column split of the
    boundary rows, onepass = interior rows in one pass + boundary rows in one pass over [owned | halo]) against the GLOBAL
    oracle.  din == dout = 128 / 64: the product rides on the aggregation in every class; 200 -> 64: not a fused shape, the
    classes aggregate and one product follows; 128 -> 47: the layer multiplies first and aggregates 47 columns"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    os.environ["GAIB_PART_MODE"] = mode
    try:
        from graphaibench_amd import capi, layers as L
        from oracle import binding as orc
        from util import LONG_SUM_FLOOR, assert_close, random_graph

        transport = _transport(transport_name, capi)
        ctx = L.init(0)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        L.set_comm(comm)
        rp, ci = random_graph(3000, 10, seed=17, power_law=True, hub_deg=1500)
        g = orc.Graph(rp, ci)
        if arch == "gcn":
            g = g.add_selfloop()
        n = g.nv
        x = np.random.default_rng(5).standard_normal((n, din)).astype(np.float32)
        gin = np.random.default_rng(6).standard_normal((n, dout)).astype(np.float32)
        lo_ = (orc.GCNLayer if arch == "gcn" else orc.SAGELayer)(1, g, din, dout, True)
        want = lo_.forward(x)
        want_go = lo_.backward(gin.copy())
        part = L.HostPartition(g.rowptr, g.colidx, rank, world)
        lo, hi = part.lo, part.hi
        lg = part.make_graph(comm)
        used, n_bnd, bnd_edges = lg.partition_mode(din)
        assert L.LGraph.PART_NAMES[used] == mode, (used, mode)
        if mode != "split":  # the classes are real: some rows of this 10-edges-per-row graph are interior, some are not
            assert 0 < n_bnd < hi - lo, (n_bnd, hi - lo)
        layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, hi - lo, din, dout, lg, True)
        layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
        out = torch.full((hi - lo, dout), float("nan"), device="cuda")
        layer.forward(out)
        assert_close(out.cpu().numpy(), want[lo:hi], "forward", floor=LONG_SUM_FLOOR)
        out.copy_(torch.from_numpy(want[lo:hi]).cuda())  # identical relu masks
        layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
        grad_out = torch.full((hi - lo, din), float("nan"), device="cuda")
        layer.backward(out, grad_out)
        assert_close(grad_out.cpu().numpy(), want_go[lo:hi], "grad_out", floor=LONG_SUM_FLOOR)
        layer.update_weight(L.adam(0.01))  # (sums the gradient over the ranks first)
        want_wg = lo_.W_grad if arch == "gcn" else lo_.W_neigh_grad
        assert_close(layer.tensor(L.W_NEIGH_GRAD, (din, dout)).cpu().numpy(), want_wg, "W_grad", floor=LONG_SUM_FLOOR)
        if arch == "sage":
            assert_close(layer.tensor(L.W_SELF_GRAD, (din, dout)).cpu().numpy(), lo_.W_self_grad, "W_self_grad",
                         floor=LONG_SUM_FLOOR)
        comm.barrier()
        layer.close()
        lg.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def _gat_worker(rank, world, idfile, q, heads, mode="fused", transport_name="ipc"):
    """GAT_layer 64 -> 64 on a vertex-range partition (h halo rows for the scores, partial gradient rows returned to
    their owners, alpha gradients all-reduced) against the GLOBAL oracle: head by head for heads > 1"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    try:
        from graphaibench_amd import capi, layers as L
        from oracle import binding as orc
        from util import LONG_SUM_FLOOR, assert_close, random_graph

        ctx = L.init(0)
        if mode == "staged":  # the round-2 pieces (SDDMM, row-side softmax backward, transposed SpMM, reverse exchange)
            ctx.set_option("gat_fused_fwd", 0)
            ctx.set_option("gat_fused_bwd", 0)
        if mode == "fwd-only":  # one-sweep forward (row statistics only), staged backward: the attention is formed again
            ctx.set_option("gat_fused_bwd", 0)
        transport = _transport(transport_name, capi)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        L.set_comm(comm)
        rp, ci = random_graph(2500, 14, seed=21, power_law=True, hub_deg=1300)
        g = orc.Graph(rp, ci).add_selfloop()
        n, din, d = g.nv, 48, 64
        x = np.random.default_rng(5).standard_normal((n, din)).astype(np.float32)
        gin = np.random.default_rng(6).standard_normal((n, d)).astype(np.float32)
        W = orc.init_glorot(din, d, 1)
        al, ar = orc.init_glorot(d, 1, 2).ravel(), orc.init_glorot(d, 1, 3).ravel()
        hfeat = orc.matmul(x, W)
        agg, temp, _, norm = orc.gat_aggregate_mh(g, hfeat, al, ar, heads)
        want = orc.relu(agg)
        g_act = orc.d_relu(gin, want)
        T, _, _, lg_w, rg_w = orc.gat_d_aggregate_mh(g, hfeat, g_act, norm, temp, heads)
        want_go = orc.matmul(T, W, False, True)
        want_wg = orc.matmul(x, T, True, False)

        part = L.HostPartition(g.rowptr, g.colidx, rank, world, gat=True)
        lo, hi = part.lo, part.hi
        lg = part.make_graph(comm)
        layer = L.Layer(L.GAT, 1, hi - lo, din, d, lg, True)
        if heads > 1:
            layer.set_heads(heads)
        layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
        out = torch.empty(hi - lo, d, device="cuda")
        layer.forward(out)
        assert_close(out.cpu().numpy(), want[lo:hi], "forward", floor=LONG_SUM_FLOOR)
        out.copy_(torch.from_numpy(want[lo:hi]).cuda())  # identical relu masks
        layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
        grad_out = torch.empty(hi - lo, din, device="cuda")
        ctx.prof_reset()
        ctx.prof_enable(True)
        layer.backward(out, grad_out)
        ctx.prof_enable(False)
        n_fused, _ = ctx.prof_get("gat_bwd_fused")
        assert (n_fused > 0) == (mode == "fused"), (mode, n_fused)  # the path that was asked for is the one that ran
        assert_close(grad_out.cpu().numpy(), want_go[lo:hi], "grad_out", floor=LONG_SUM_FLOOR)
        # weight and alpha gradients are partial sums until the optimizer step all-reduces them
        for which, shape, want_g, name in ((L.W_NEIGH_GRAD, (din, d), want_wg, "W_grad"), (L.ALPHA_LGRAD, (d,), lg_w, "alpha_l"),
                                           (L.ALPHA_RGRAD, (d,), rg_w, "alpha_r")):
            t = layer.tensor(which, shape).reshape(-1).contiguous()
            comm.allreduce(t)
            assert_close(t.cpu().numpy().reshape(shape), want_g, name, floor=LONG_SUM_FLOOR)
        comm.barrier()
        # the GAT structures of a partition (the [owned | halo] graph, its transpose, the edge permutation), the aggregator's
        # partition tables and per-edge arrays go back with LearningGraph::dealloc / GAT_Aggregator::release
        def free_now():
            L.sync()
            comm.barrier()
            f = torch.cuda.mem_get_info()[0]
            comm.barrier()
            return f

        def build_use_close():
            lg2 = part.make_graph(comm)
            l2 = L.Layer(L.GAT, 1, hi - lo, din, d, lg2, True)
            if heads > 1:
                l2.set_heads(heads)
            l2.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
            l2.forward(out)
            l2.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
            l2.backward(out, grad_out)
            L.sync()
            l2.close()
            lg2.close()

        build_use_close()
        base = free_now()
        for _ in range(3):
            build_use_close()
        lost, tries = comm.allreduce_host([float(base - free_now())])[0] / world, 0
        while lost > (4 << 20) * world and tries < 12:  # (see _worker: a peer's freed memory can come back a little later)
            time.sleep(0.25)
            lost, tries = comm.allreduce_host([float(base - free_now())])[0] / world, tries + 1
        assert lost <= (4 << 20) * world, f"{lost / 2**20:.1f} MiB of the device not returned by {world} ranks"
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def _halo_realloc_worker(rank, world, idfile, q, transport_name="ipc", chunk_bytes=0):
    """raw halo plan: exchange and reverse exchange with GROWING row lengths in the order that left a stale hipIpc
    handle behind (exchange 16, reduce 16, exchange 64, reduce 64: the reduce's table was reallocated by the exchange
    before it, the exchange's send buffer by the reduce before it), and more plans over a communicator's life than it
    has slot rows (ids are released by gaib_halo_destroy)"""
    sys.path.insert(0, str(ROOT))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    if chunk_bytes:  # IPC: send buffers above this size are cut into separately exported chunks (comm.hip)
        os.environ["GAIB_IPC_CHUNK_BYTES"] = str(chunk_bytes)
    try:
        from graphaibench_amd import capi, layers as L

        ctx = L.init(0)
        transport = _transport(transport_name, capi)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        n_own = 5000
        rng = np.random.default_rng(100)  # the same plan on every rank: rank r needs rows need[r][q] of rank q
        need = [[np.sort(rng.choice(n_own, 700 + 50 * (r + q), replace=False)) if q != r else np.empty(0, np.int64)
                 for q in range(world)] for r in range(world)]
        send_idx = np.concatenate([need[q][rank] for q in range(world)]).astype(np.int64)  # grouped by destination
        send_counts = [len(need[q][rank]) for q in range(world)]
        recv_counts = [len(need[rank][q]) for q in range(world)]

        def rows_of(r, length, salt):  # rank r's rows, reproducible on every rank
            return np.random.default_rng(1000 * r + length + salt).standard_normal((n_own, length)).astype(np.float32)

        def check(halo, length, salt):
            mine = torch.from_numpy(rows_of(rank, length, salt)).cuda()
            halo.begin(mine, length)
            ptr = halo.end()
            ctx.sync()
            got = torch.empty(max(halo.rows, 1), length, device="cuda")
            capi._check(ctx.lib.gaib_memcpy_d2d(ctx.h, got.data_ptr(), ptr, halo.rows * length * 4), "d2d")
            ctx.sync()
            want = np.concatenate([rows_of(q, length, salt)[need[rank][q]] for q in range(world)])
            assert np.array_equal(got[:halo.rows].cpu().numpy(), want), f"exchange len {length}"
            # reverse: every rank returns (its copy of the halo rows) * (rank + 1); owners add peer by peer
            partial = got[:halo.rows] * float(rank + 1)
            acc = torch.zeros(n_own, length, device="cuda")
            halo.reduce(partial, acc, length)
            ctx.sync()
            want_acc = np.zeros((n_own, length), np.float32)
            for q in range(world):
                if q != rank:
                    want_acc[need[q][rank]] += rows_of(rank, length, salt)[need[q][rank]] * np.float32(q + 1)
            assert np.array_equal(acc.cpu().numpy(), want_acc), f"reduce len {length}"

        halo = comm.halo(send_counts, send_idx, recv_counts)
        for salt, length in enumerate((16, 16, 64, 8, 200)):
            check(halo, length, salt)
        halo.close()
        for k in range(10):  # > GAIB_COMM_MAX_HALOS plans, one alive at a time
            h2 = comm.halo(send_counts, send_idx, recv_counts)
            check(h2, 24 + k, 50 + k)
            h2.close()
        with pytest.raises(capi.GaibError):  # a refused plan does not leak its slot
            comm.halo([1] * world, send_idx, recv_counts)
        alive = [comm.halo(send_counts, send_idx, recv_counts) for _ in range(8)]
        with pytest.raises(capi.GaibError, match="at most 8"):
            comm.halo(send_counts, send_idx, recv_counts)
        check(alive[7], 32, 99)
        for h in alive:
            h.close()
        comm.barrier()
        if chunk_bytes:
            # a table that an exchange of long rows left behind is too large to export although the reverse exchange that
            # follows moves few bytes: its rows are staged in an allocation of their own
            os.environ["GAIB_IPC_CHUNK_BYTES"], os.environ["GAIB_IPC_EXPORT_LIMIT_BYTES"] = str(2 << 20), str(5 << 19)
            h4 = comm.halo(send_counts, send_idx, recv_counts)
            check(h4, 512, 7)
            check(h4, 16, 8)
            h4.close()
            # an allocation above the export limit is refused loudly (hipIpcOpenMemHandle hangs above 2 GiB)
            os.environ["GAIB_IPC_CHUNK_BYTES"] = str(chunk_bytes)
            os.environ["GAIB_IPC_EXPORT_LIMIT_BYTES"] = str(1 << 20)
            h3 = comm.halo(send_counts, send_idx, recv_counts)
            with pytest.raises(capi.GaibError, match="above what this transport exports"):
                h3.begin(torch.zeros(n_own, 100, device="cuda"), 100)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def _spawn(world, target, args, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world) + args[:1] + (q,) + args[1:]) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:  # (pytest shortens the assertion's repr: the ranks' own messages in full)
        if r[1] != "ok":
            print(f"--- rank {r[0]}: {r[1]}", file=sys.stderr)
    return res


@pytest.mark.parametrize("arch,world", [("gcn", 2), ("sage", 2), ("gcn", 3)])
def test_ipc_ranks_on_one_gpu_match_global_oracle(tmp_path, arch, world):
    res = _spawn(world, _worker, (str(tmp_path / "id"), arch, "ipc"))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("mode", ["split", "classes", "onepass"])
@pytest.mark.parametrize("arch,world,din,dout", [("gcn", 2, 128, 128), ("sage", 3, 128, 128), ("gcn", 3, 64, 64),
                                                 ("gcn", 2, 200, 64), ("sage", 2, 128, 47), ("sage", 2, 100, 128)])
def test_ipc_row_class_modes_match_global_oracle(tmp_path, arch, world, din, dout, mode):
    res = _spawn(world, _mode_worker, (str(tmp_path / "id"), arch, mode, din, dout))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("mode", ["classes", "onepass"])
def test_rccl_branch_row_class_modes(tmp_path, mode):
    res = _spawn(2, _mode_worker, (str(tmp_path / "id"), "gcn", mode, 128, 128, "fake-rccl"))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("heads,world,mode", [(1, 2, "fused"), (8, 2, "fused"), (8, 3, "fused"), (4, 3, "fused"),
                                              (1, 2, "staged"), (8, 3, "staged"), (8, 2, "fwd-only")])
def test_ipc_gat_layer_on_partition_matches_global_oracle(tmp_path, heads, world, mode):
    """GAT_layer 48 -> 64 on a vertex-range partition against the GLOBAL oracle.  fused: the one-sweep kernels on the
    rank's rectangular [owned | halo] graph (forward: h halo rows, the chunks over owned columns swept while they travel;
    backward: grad rows + (rowdot, max, 1 / sum) records of the halo vertices, everything about an owned row from its
    own edge list -- no transposed structure, no reverse exchange).  staged: the round-2 pieces (options off)."""
    res = _spawn(world, _gat_worker, (str(tmp_path / "id"), heads, mode))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_halo_buffers_regrow_between_exchange_and_reduce(tmp_path, world):
    res = _spawn(world, _halo_realloc_worker, (str(tmp_path / "id"),))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("world,chunk_bytes", [(2, 50000), (3, 50000), (3, 24000)])
def test_ipc_send_buffer_in_chunks(tmp_path, world, chunk_bytes):
    """hipIpcOpenMemHandle of an allocation above 2 GiB does not return (bench.py --gpus 3 --cut-fraction 0.3 on one device:
    2.4 GB send buffers), so the IPC transport cuts a send buffer into separately exported chunks of whole rows and a receiver
    opens the chunks its segment touches.  Here with chunks of 50 / 24 kB: the exchanges and reverse exchanges of the regrow
    test (1 to 55 chunks as the row length goes 16, 16, 64, 8, 200, ...; a layout that shrinks again) deliver the same rows,
    and an allocation above the export limit is an error on every rank, not a hang"""
    res = _spawn(world, _halo_realloc_worker, (str(tmp_path / "id"), "ipc", chunk_bytes))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("arch,world", [("gcn", 2), ("sage", 2), ("gcn", 3)])
def test_rccl_branch_with_strict_double_matches_global_oracle(tmp_path, arch, world):
    """comm.hip's RCCL branch AS WRITTEN with N > 1 ranks: grouped send / recv of the halo exchange and of the reverse
    exchange (offsets, counts, peers), fp32 / fp64 all-reduces on the communication stream, layer results against the
    GLOBAL oracle.  The bytes travel through tests/fake_rccl (this box has one GPU and RCCL wants one per rank)."""
    res = _spawn(world, _worker, (str(tmp_path / "id"), arch, "fake-rccl"))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("heads,world,mode", [(8, 2, "fused"), (4, 3, "fused"), (8, 3, "staged")])
def test_rccl_branch_gat_on_partition(tmp_path, heads, world, mode):
    res = _spawn(world, _gat_worker, (str(tmp_path / "id"), heads, mode, "fake-rccl"))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_branch_exchange_and_reduce_with_growing_rows(tmp_path, world):
    res = _spawn(world, _halo_realloc_worker, (str(tmp_path / "id"), "fake-rccl"))
    assert all(r[1] == "ok" for r in res), res


def _slices(rows, K):
    """slice k of a peer pair's `rows` rows (gaib_halo_piece_slice's arithmetic, restated)"""
    return [(rows * k // K, rows * (k + 1) // K) for k in range(K)]


def _pieces_raw_worker(rank, world, idfile, q, transport_name, chunk_bytes=0):
    """a raw halo plan whose exchanges travel in K time slices (gaib_halo_set_pieces): after wait_piece(k) the table's rows of
    slices 0..k are the peers' rows (a snapshot enqueued behind every wait proves the ordering the stream sees), the table
    after end() is the whole exchange, K changes between exchanges (1 -> 4 -> 3 -> 16 -> 1), the reverse exchange is untouched,
    and ranks that disagree on K get an error, not wrong rows (the peer-to-peer transport checks)"""
    sys.path.insert(0, str(ROOT))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    if chunk_bytes:
        os.environ["GAIB_IPC_CHUNK_BYTES"] = str(chunk_bytes)
    try:
        from graphaibench_amd import capi, layers as L

        ctx = L.init(0)
        transport = _transport(transport_name, capi)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        n_own = 4000
        rng = np.random.default_rng(100)
        need = [[np.sort(rng.choice(n_own, 600 + 37 * (r + 2 * q), replace=False)) if q != r else np.empty(0, np.int64)
                 for q in range(world)] for r in range(world)]
        if world > 2:  # one pair moves fewer rows than there are slices, one peer's list is a run of consecutive rows (direct send)
            need[0][1] = np.array([5, 9, 11], np.int64)
            need[1][2] = np.arange(100, 900, dtype=np.int64)
        send_idx = np.concatenate([need[q][rank] for q in range(world)]).astype(np.int64)
        send_counts = [len(need[q][rank]) for q in range(world)]
        recv_counts = [len(need[rank][q]) for q in range(world)]
        recv_off = np.concatenate([[0], np.cumsum(recv_counts)])

        def rows_of(r, length, salt):
            return np.random.default_rng(1000 * r + length + salt).standard_normal((n_own, length)).astype(np.float32)

        halo = comm.halo(send_counts, send_idx, recv_counts)
        assert halo.pieces == 1
        for salt, (K, length) in enumerate([(1, 16), (4, 16), (4, 128), (3, 40), (16, 8), (1, 64), (2, 200)]):
            halo.set_pieces(K)
            assert halo.pieces == K
            want = np.concatenate([rows_of(q_, length, salt)[need[rank][q_]] for q_ in range(world)]) if halo.rows else \
                np.zeros((0, length), np.float32)
            # the library's column ranges == the restated arithmetic
            for k in range(K):
                mine = [(int(recv_off[q_] + lo), int(recv_off[q_] + hi)) for q_ in range(world)
                        for lo, hi in [_slices(recv_counts[q_], K)[k]] if hi > lo]
                assert halo.piece_ranges(k) == mine, (k, halo.piece_ranges(k), mine)
            src = torch.from_numpy(rows_of(rank, length, salt)).cuda()
            snaps = []
            halo.begin(src, length)
            for k in range(K):
                ptr = halo.wait_piece(k)
                snap = torch.full((max(halo.rows, 1), length), float("nan"), device="cuda")
                if halo.rows:  # on the compute stream, behind the wait
                    capi._check(ctx.lib.gaib_memcpy_d2d(ctx.h, snap.data_ptr(), ptr, halo.rows * length * 4), "d2d")
                snaps.append(snap)
            ptr = halo.end()
            got = torch.empty(max(halo.rows, 1), length, device="cuda")
            capi._check(ctx.lib.gaib_memcpy_d2d(ctx.h, got.data_ptr(), ptr, halo.rows * length * 4), "d2d")
            ctx.sync()
            assert np.array_equal(got[:halo.rows].cpu().numpy(), want), f"exchange K {K} len {length}"
            for k, snap in enumerate(snaps):
                sn = snap.cpu().numpy()
                for j in range(k + 1):
                    for b, e in halo.piece_ranges(j):
                        assert np.array_equal(sn[b:e], want[b:e]), f"K {K} len {length}: piece {j} not there after wait_piece({k})"
            # the reverse exchange does not care about K
            partial = got[:halo.rows] * float(rank + 1)
            acc = torch.zeros(n_own, length, device="cuda")
            halo.reduce(partial, acc, length)
            ctx.sync()
            want_acc = np.zeros((n_own, length), np.float32)
            for q_ in range(world):
                if q_ != rank:
                    want_acc[need[q_][rank]] += rows_of(rank, length, salt)[need[q_][rank]] * np.float32(q_ + 1)
            assert np.array_equal(acc.cpu().numpy(), want_acc), f"reduce K {K} len {length}"
        # misuse
        with pytest.raises(capi.GaibError):
            halo.set_pieces(0)
        with pytest.raises(capi.GaibError):
            halo.set_pieces(17)
        with pytest.raises(capi.GaibError, match="no exchange in flight"):
            halo.wait_piece(0)
        halo.set_pieces(2)
        halo.begin(torch.zeros(n_own, 8, device="cuda"), 8)
        with pytest.raises(capi.GaibError, match="in flight"):
            halo.set_pieces(3)
        with pytest.raises(capi.GaibError, match="piece 2 of 2"):
            halo.wait_piece(2)
        halo.end()
        comm.barrier()
        if transport_name == "ipc":  # ranks that disagree on K: an error on the ranks that pull from the odd one out
            halo.set_pieces(3 if rank == 0 else 2)
            try:
                halo.begin(torch.zeros(n_own, 8, device="cuda"), 8)
                halo.end()
                failed = False
            except capi.GaibError:
                failed = True
            assert failed or rank == 0, "a peer cut the exchange differently and nobody noticed"
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def _pieces_layer_worker(rank, world, idfile, q, arch, mode, K, transport_name):
    """a GCN / SAGE layer on a vertex-range partition whose exchanges travel in K slices: the library cuts the halo-column half
    into K piece graphs and aggregates them as the slices land.  Against (a) the oracle's run on the GLOBAL graph and (b) the
    UNPIPED exchange -- the same partition with a plan of one piece and the halo graph's rows in piece-major order (the order
    the pieces add a row's terms in; with one peer the column order itself): bit for bit on rows below the heavy threshold"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "60"
    os.environ["GAIB_PART_MODE"] = mode
    os.environ["GAIB_HALO_PIECES"] = str(K)   # slices on the wire ...
    os.environ["GAIB_HALO_CONSUME"] = str(K)  # ... consumed one to one (the rule would take one piece on a graph this small)
    try:
        from graphaibench_amd import capi, layers as L
        from oracle import binding as orc
        from util import LONG_SUM_FLOOR, assert_close, random_graph, rel_err

        transport = _transport(transport_name, capi)
        ctx = L.init(0)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        L.set_comm(comm)
        rp, ci = random_graph(3000, 12, seed=19, power_law=True, hub_deg=700)
        g = orc.Graph(rp, ci)
        if arch == "gcn":
            g = g.add_selfloop()
        n, D = g.nv, 128
        x = np.random.default_rng(5).standard_normal((n, D)).astype(np.float32)
        gin = np.random.default_rng(6).standard_normal((n, D)).astype(np.float32)
        lo_ = (orc.GCNLayer if arch == "gcn" else orc.SAGELayer)(1, g, D, D, True)
        want = lo_.forward(x)
        want_go = lo_.backward(gin.copy())
        part = L.HostPartition(g.rowptr, g.colidx, rank, world)
        lo, hi = part.lo, part.hi

        def run(lg, pieces):
            used, _, _ = lg.partition_mode(D)
            assert L.LGraph.PART_NAMES[used] == mode
            assert lg.halo_pieces(D) == pieces, (lg.halo_pieces(D), pieces)
            layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, hi - lo, D, D, lg, True)
            layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
            out = torch.full((hi - lo, D), float("nan"), device="cuda")
            layer.forward(out)
            L.sync()
            fwd = out.cpu().numpy().copy()
            out.copy_(torch.from_numpy(want[lo:hi]).cuda())  # identical relu masks
            layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
            grad_out = torch.full((hi - lo, D), float("nan"), device="cuda")
            layer.backward(out, grad_out)
            L.sync()
            go = grad_out.cpu().numpy().copy()
            layer.update_weight(L.adam(0.01))  # (sums the gradient over the ranks first)
            wg = layer.tensor(L.W_NEIGH_GRAD, (D, D)).cpu().numpy().copy()
            comm.barrier()
            layer.close()
            return fwd, go, wg

        lg = part.make_graph(comm)  # (GAIB_HALO_PIECES: the plan travels in K slices)
        fwd, go, wg = run(lg, K)
        assert_close(fwd, want[lo:hi], "forward", floor=LONG_SUM_FLOOR)
        assert_close(go, want_go[lo:hi], "grad_out", floor=LONG_SUM_FLOOR)
        assert_close(wg, lo_.W_grad if arch == "gcn" else lo_.W_neigh_grad, "W_grad", floor=LONG_SUM_FLOOR)
        lg.close()
        # (b) the unpiped exchange over the piece-major halo graph
        n_own, nh = hi - lo, len(part.halo_gids)
        recv_off = np.concatenate([[0], np.cumsum(part.recv_counts)])
        piece = np.zeros(max(nh, 1), np.int64)
        for q_ in range(world):
            for k, (a, b) in enumerate(_slices(int(part.recv_counts[q_]), K)):
                piece[recv_off[q_] + a:recv_off[q_] + b] = k
        cih = part.colidx_halo.astype(np.int64)
        rows = np.repeat(np.arange(n_own), np.diff(part.rowptr_halo))
        order = np.lexsort((cih, piece[cih], rows))
        if world == 2:
            assert np.array_equal(order, np.arange(len(cih)))  # one peer: piece-major is the column order
        def norms(deg):
            t = np.sqrt(deg.astype(np.float32))
            vd = np.where(t == 0, 0.0, 1.0 / np.maximum(t, 1e-30).astype(np.float64)).astype(np.float32)
            inv = (1.0 / deg.astype(np.float32).astype(np.float64)).astype(np.float32)
            return torch.from_numpy(vd).cuda(), torch.from_numpy(inv).cuda()
        vd, inv = norms(part.degree)
        vd_h, inv_h = norms(part.halo_degree if nh else np.ones(1, np.int64))
        g_own = ctx.graph(part.rowptr_own, part.colidx_own.view(np.int32))
        g_own.set_vertex_norm(vd, vd, inv, row_inv_deg=inv)
        g_halo = ctx.graph(part.rowptr_halo, cih[order].astype(np.int32), ncols=max(nh, 1))
        g_halo.set_vertex_norm(vd, vd_h, inv_h, row_inv_deg=inv)
        plan = comm.halo(part.send_counts, part.send_idx, part.recv_counts)  # one piece
        lg2 = L.LGraph.adopt(g_own)
        lg2.set_halo_plan(g_halo, plan)
        fwd2, go2, wg2 = run(lg2, 1)
        light = (np.diff(part.rowptr_own) <= 1024) & (np.diff(part.rowptr_halo) <= 1024)
        assert light.sum() >= n_own - 2
        for a, b, what in ((fwd, fwd2, "forward"), (go, go2, "grad_out")):
            assert np.isfinite(a).all() and np.isfinite(b).all()
            assert np.array_equal(a[light].view(np.uint32), b[light].view(np.uint32)), f"{what}: piped != unpiped"
            assert rel_err(a, b) < 1e-5
        assert rel_err(wg, wg2) < 1e-5
        comm.barrier()
        lg2.close()
        plan.close()
        g_halo.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("transport_name,world,chunk_bytes", [("ipc", 2, 0), ("ipc", 3, 0), ("ipc", 3, 30000), ("fake-rccl", 2, 0),
                                                             ("fake-rccl", 3, 0)])
def test_halo_exchange_in_time_slices(tmp_path, transport_name, world, chunk_bytes):
    res = _spawn(world, _pieces_raw_worker, (str(tmp_path / "id"), transport_name, chunk_bytes))
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("arch,mode,K,world,transport_name", [
    ("gcn", "split", 4, 2, "ipc"), ("gcn", "split", 2, 3, "ipc"), ("sage", "split", 3, 3, "ipc"), ("gcn", "classes", 4, 2, "ipc"),
    ("gcn", "classes", 2, 3, "ipc"), ("gcn", "split", 4, 2, "fake-rccl"), ("gcn", "split", 3, 3, "fake-rccl"),
    ("sage", "classes", 2, 3, "fake-rccl")])
def test_layers_consume_the_halo_piece_by_piece(tmp_path, arch, mode, K, world, transport_name):
    res = _spawn(world, _pieces_layer_worker, (str(tmp_path / "id"), arch, mode, K, transport_name))
    assert all(r[1] == "ok" for r in res), res


def _fewer_vertices_than_ranks(rank, world, idfile, q, transport_name):
    """ranks WITHOUT rows (2 vertices on 3 ranks, 1 vertex on 3 ranks) take part in every exchange, reverse exchange and
    all-reduce with empty buffers, take the SAME path (one sweep / staged) as the ranks with rows -- the choice follows
    from the shape and the options alone -- and contribute zero gradients; plus a few larger cases through the same
    checker (scripts/fuzz_partition.py::run_case: the layer against the fp64 GLOBAL evaluation)"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "scripts"))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "30"
    os.environ["GAIB_FAKE_RCCL_TIMEOUT_S"] = "30"
    try:
        from fuzz_partition import run_case
        from graphaibench_amd import capi, layers as L

        transport = _transport(transport_name, capi)
        ctx = L.init(0)
        comm = capi.Comm(ctx, rank, world, _id_via_file(idfile, rank, transport, capi), transport)
        L.set_comm(comm)
        # a DIRECTED graph (the one-sweep kernels read a row's in-edges off its out-edges, which is only right on a
        # structurally symmetric graph): make_partitioned_graph finds the missing reverse edges, all ranks agree, and the layer
        # takes the staged path -- transposed structure + reverse exchange -- whose results match the fp64 evaluation
        for heads in (1, 8):
            run_case(comm, rank, world, dict(n=2500, avg=12.0, hub=0, gseed=77, arch="gat", din=48, d=64, heads=heads,
                                             drop_reverse=0.3))
        for n, avg in ((1, 0.0), (2, 4.0), (3, 0.7), (5, 4.0), (300, 6.0)):
            for arch, d, heads in (("gcn", 64, 1), ("gcn", 16, 1), ("sage", 64, 1), ("gat", 64, 8), ("gat", 64, 1), ("gat", 128, 4),
                                   ("gat", 16, 2)):
                cfg = dict(n=n, avg=avg, hub=0, gseed=1234 + n, arch=arch, din=48, d=d, heads=heads)
                run_case(comm, rank, world, cfg)
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("transport_name", ["ipc", "fake-rccl"])
def test_ranks_without_rows_follow_the_others(tmp_path, transport_name):
    res = _spawn(3, _fewer_vertices_than_ranks, (str(tmp_path / "id"), transport_name), timeout=300)
    assert all(r[1] == "ok" for r in res), res


def _double_is_strict(rank, world, idfile, q):
    """the double itself, through its C entry points: what it must REFUSE"""
    import ctypes

    try:
        lib = ctypes.CDLL(str(FAKE_RCCL))
        os.environ["GAIB_FAKE_RCCL_TIMEOUT_S"] = "20"

        class Uid(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]

        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
        for f in (lib.ncclSend, lib.ncclRecv):
            f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        F32, SUM = 7, 0  # ncclFloat32, ncclSum (rccl.h)
        uid = Uid()
        if rank == 0:
            assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
            with open(idfile + ".tmp", "wb") as f:
                f.write(ctypes.string_at(ctypes.byref(uid), 128))  # (uid.internal would stop at the first NUL)
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile) and time.time() - t0 < 60:
                time.sleep(0.005)
            raw = open(idfile, "rb").read()
            ctypes.memmove(ctypes.byref(uid), raw.ljust(128, b"\0"), 128)
        comm = ctypes.c_void_p()
        assert lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
        t = torch.arange(1000, dtype=torch.float32, device="cuda") + 1000 * rank
        peer = 1 - rank
        # 1. a matched pair works and carries the bytes
        got = torch.empty(1000, device="cuda")
        if rank == 0:
            assert lib.ncclSend(t.data_ptr(), 1000, F32, peer, comm, None) == 0
            assert lib.ncclRecv(got.data_ptr(), 1000, F32, peer, comm, None) == 0
        else:
            assert lib.ncclRecv(got.data_ptr(), 1000, F32, peer, comm, None) == 0
            assert lib.ncclSend(t.data_ptr(), 1000, F32, peer, comm, None) == 0
        assert torch.equal(got, torch.arange(1000, dtype=torch.float32, device="cuda") + 1000 * peer)
        # 2. all-reduce: the sum, identical on both
        a = torch.full((333,), float(rank + 1), device="cuda")
        assert lib.ncclAllReduce(a.data_ptr(), a.data_ptr(), 333, F32, SUM, comm, None) == 0
        assert torch.all(a == 3.0)
        # 3. a range that leaves its allocation is refused before anything moves (local check, both ranks)
        small = torch.empty(16, device="cuda")
        assert lib.ncclSend(small.data_ptr(), 1 << 22, F32, peer, comm, None) != 0
        assert lib.ncclSend(t.data_ptr(), 10, F32, world, comm, None) != 0  # peer out of range
        # 4. counts that do not pair up: the receive fails (the real library would hang or overwrite)
        if rank == 0:
            assert lib.ncclSend(t.data_ptr(), 100, F32, peer, comm, None) == 0
        else:
            assert lib.ncclRecv(got.data_ptr(), 50, F32, peer, comm, None) != 0
        lib.ncclCommDestroy(comm)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_the_rccl_double_refuses_what_the_real_library_would_not_survive(tmp_path):
    res = _spawn(2, _double_is_strict, (str(tmp_path / "id"),), timeout=180)
    assert all(r[1] == "ok" for r in res), res


def test_rccl_one_rank(tmp_path):
    """the RCCL transport end to end with the one GPU this box has"""
    res = _spawn(1, _worker, (str(tmp_path / "id"), "gcn", "rccl"))
    assert all(r[1] == "ok" for r in res), res


def _lonely(rank, world, idfile, q):
    sys.path.insert(0, str(ROOT))
    os.environ["GAIB_COMM_TIMEOUT_S"] = "4"
    try:
        from graphaibench_amd import capi, layers as L

        ctx = L.init(0)
        t0 = time.time()
        try:
            capi.Comm(ctx, 0, 2, capi.comm_unique_id(capi.COMM_IPC), capi.COMM_IPC)  # rank 1 never shows up
        except capi.GaibError as e:
            q.put((rank, "ok" if "timed out" in str(e) and time.time() - t0 < 30 else f"FAIL: {e}"))
            return
        q.put((rank, "FAIL: init returned without a peer"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_missing_peer_is_an_error_not_a_hang(tmp_path):
    res = _spawn(1, _lonely, (str(tmp_path / "id"),), timeout=120)
    assert res[0][1] == "ok", res
