"""Randomised sweep of the partitioned layers: WORLD processes on one GPU (vertex-range partition built by the host C++ from the
global CSR, halo exchange + reverse exchange + gradient all-reduce behind the C ABI) run GCN / SAGE / GAT layers forward and
backward on random graphs -- fewer vertices than ranks, ranks without rows, isolated vertices, hubs -- and every rank compares
its slice with an fp64 evaluation of the GLOBAL layer on the device.  Transport: the peer-to-peer pull (ipc) or comm.hip's
RCCL branch bound to tests/fake_rccl (fake-rccl).
    python scripts/fuzz_partition.py [--world 3] [--seconds 90] [--seed 0] [--transport ipc|fake-rccl]
Test infrastructure (a development tool: what it finds becomes a case in tests/)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "scripts"))


def run_case(comm, rank, world, cfg):
    """one layer forward + backward on this rank's share of the graph cfg describes, compared with the fp64 evaluation of the
    GLOBAL layer; collective (every rank of `comm` calls it with the same cfg).  Returns the worst error; raises on a mismatch."""
    from fuzz_gat_layer import fp64_layer
    from graphaibench_amd import layers as L
    from util import random_graph

    n, avg, hub, gseed, arch, din, d, heads = (cfg[k] for k in ("n", "avg", "hub", "gseed", "arch", "din", "d", "heads"))
    worst = 0.0
    rp, ci = random_graph(n, avg, seed=gseed, power_law=bool(gseed & 1), hub_deg=hub)
    if arch != "sage":  # A + I (net.cpp:96); SAGE aggregates over A
        rows = np.repeat(np.arange(n), np.diff(rp))
        key = np.unique(np.concatenate([rows * n + ci.astype(np.int64), np.arange(n) * (n + 1)]))
        rp = np.zeros(n + 1, np.int64)
        np.add.at(rp, key // n + 1, 1)
        rp, ci = np.cumsum(rp), (key % n).astype(np.uint32)
    if cfg.get("drop_reverse", 0.0) > 0:  # a DIRECTED graph: a share of the edges u -> v (u < v) goes, v -> u stays
        rows = np.repeat(np.arange(n), np.diff(rp))
        keep = ~((rows < ci) & (np.random.default_rng(gseed + 7).random(len(ci)) < cfg["drop_reverse"]))
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int64)
        ci = ci[keep]
    part = L.HostPartition(rp, ci, rank, world, gat=arch == "gat")
    lo, hi = part.lo, part.hi
    lg = part.make_graph(comm)
    if "part_mode" in cfg:  # row classes of the partition (LearningGraph::partition_mode): -1 rule, 0 split, 1 classes, 2 / 3 one pass
        lg.set_partition_mode(cfg["part_mode"])
    kind = {"gcn": L.GCN, "sage": L.SAGE, "gat": L.GAT}[arch]
    layer = L.Layer(kind, 1, hi - lo, din, d, lg, False)
    if heads > 1:
        layer.set_heads(heads)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(gseed)
    x = torch.randn(n, din, device="cuda", generator=gen)
    gin = torch.randn(n, d, device="cuda", generator=gen)
    W = layer.tensor(L.W_NEIGH, (din, d))
    layer.write(L.FEAT_IN, x[lo:hi].contiguous())
    out = torch.empty(hi - lo, d, device="cuda")
    layer.forward(out)
    layer.write(L.GRAD_IN, gin[lo:hi].contiguous())
    go = torch.empty(hi - lo, din, device="cuda")
    layer.backward(out, go)
    L.sync()
    got = dict(out=out, grad_out=go)
    sums = dict(W_grad=layer.tensor(L.W_NEIGH_GRAD, (din, d)))
    rowptr = torch.from_numpy(rp).cuda()
    col = torch.from_numpy(ci.astype(np.int64)).cuda()
    r_ = torch.repeat_interleave(torch.arange(n, device="cuda"), rowptr[1:] - rowptr[:-1])
    deg = (rowptr[1:] - rowptr[:-1]).double()
    X, G, Wd = x.double(), gin.double(), W.double()

    def A(w, src, dst, M):  # sum_e w_e M[src_e] into row dst_e
        return torch.zeros(n, M.shape[1], dtype=torch.float64, device="cuda").index_add_(0, dst, w[:, None] * M[src])

    if arch == "gcn":
        vd = torch.where(deg > 0, deg.sqrt().reciprocal(), torch.zeros_like(deg))
        w = vd[r_] * vd[col]
        ax = A(w, col, r_, X)
        want = dict(out=ax @ Wd, grad_out=A(w, col, r_, G) @ Wd.t(), W_grad=ax.t() @ G)
    elif arch == "sage":
        Ws = layer.tensor(L.W_SELF, (din, d)).double()
        inv = torch.where(deg > 0, deg.reciprocal(), torch.zeros_like(deg))
        mx = A(inv[r_], col, r_, X)
        want = dict(out=mx @ Wd + X @ Ws, grad_out=A(inv[col], col, r_, G @ Wd.t()) + G @ Ws.t(), W_grad=mx.t() @ G,
                    W_self_grad=X.t() @ G)
        sums["W_self_grad"] = layer.tensor(L.W_SELF_GRAD, (din, d))
    else:
        al, ar = layer.tensor(L.ALPHA_L, (d,)), layer.tensor(L.ALPHA_R, (d,))
        y, go64, wg, lg64, rg64, t, mag = fp64_layer(rowptr, col, x, W, al, ar, gin, heads, False)
        want = dict(out=y, grad_out=go64, W_grad=wg, alpha_l=lg64, alpha_r=rg64)
        sums["alpha_l"], sums["alpha_r"] = layer.tensor(L.ALPHA_LGRAD, (d,)), layer.tensor(L.ALPHA_RGRAD, (d,))
        # a score within rounding of zero makes leaky_relu' a coin flip between two correct fp32 evaluations (and fp64): the
        # gradients that go through it are then not compared (scripts/fuzz_gat_layer.py does the same; the tests impose the
        # GPU's own signs on the fp64 evaluation instead, tests/test_gpu_fullsize.py)
        if bool((t.abs() < 1e-5 * t.abs().max()).any()):
            for k in ("alpha_l", "alpha_r"):
                want.pop(k)
                sums.pop(k)
    for name, tns in sums.items():  # gradients of replicated parameters: the sum over the ranks
        tns = tns.contiguous()
        comm.allreduce(tns)
        got[name] = tns
    for name, gv in got.items():
        ref = want[name][lo:hi] if name in ("out", "grad_out") else want[name]
        scale = max(float(want[name].abs().max()), 1e-30)
        if name.startswith("alpha"):
            scale = max(float(lg64.abs().max()), float(rg64.abs().max()), 1e-2 * mag)
        if not torch.isfinite(gv).all():
            raise AssertionError(f"{name}: non-finite")
        e = float((gv.double() - ref).abs().max()) / scale if gv.numel() else 0.0
        worst = max(worst, e)
        if e > 2e-4:
            raise AssertionError(f"{name}: {e:.3e} of the tensor's scale from fp64 (rows {lo}:{hi})")
    layer.close()
    lg.close()
    part.close()
    return worst


def worker(rank, world, idfile, q, seconds, seed, transport_name):
    os.environ["GAIB_COMM_TIMEOUT_S"] = "15"
    os.environ["GAIB_FAKE_RCCL_TIMEOUT_S"] = "15"
    try:
        from graphaibench_amd import capi, layers as L

        if transport_name == "fake-rccl":
            os.environ["GAIB_RCCL_LIB"] = str(ROOT / "tests" / "fake_rccl" / "librccl_fake.so")
        transport = capi.COMM_IPC if transport_name == "ipc" else capi.COMM_RCCL
        ctx = L.init(0)
        if rank == 0:
            uid = capi.comm_unique_id(transport)
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not (os.path.exists(idfile) and os.path.getsize(idfile) == 128) and time.time() - t0 < 60:
                time.sleep(0.005)
            uid = open(idfile, "rb").read()
        comm = capi.Comm(ctx, rank, world, uid, transport)
        L.set_comm(comm)
        rng = np.random.default_rng(seed)  # the SAME stream on every rank: the ranks agree on every case
        t_end = time.time() + seconds
        n_cases, fails, worst = 0, [], 0.0
        while True:
            # all ranks stop together: rank 0's clock decides
            stop = comm.allreduce_host([1.0 if (rank == 0 and time.time() > t_end) else 0.0])[0]
            if stop > 0:
                break
            n = int(rng.choice([1, 2, 3, 5, 40, 500, 3000]))
            avg = float(rng.choice([0.0, 0.7, 4, 20]))
            hub = int(rng.choice([0, 0, 900])) if n >= 3000 else 0
            gseed = int(rng.integers(1 << 30))
            arch = str(rng.choice(["gcn", "sage", "gat"]))
            din = int(rng.choice([16, 48, 64, 128, 200]))
            d = int(rng.choice([16, 47, 64, 128]))
            heads = int(rng.choice([h for h in (1, 2, 4, 8) if d % h == 0])) if arch == "gat" else 1
            cfg = dict(n=n, avg=avg, hub=hub, gseed=gseed, arch=arch, din=din, d=d, heads=heads, world=world,
                       part_mode=int(rng.choice([-1, 0, 1, 2, 3])))
            err = ""
            try:
                worst = max(worst, run_case(comm, rank, world, cfg))
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"[:300]
            # every rank learns whether ANY rank failed (and keeps going in step)
            bad = comm.allreduce_host([1.0 if err else 0.0])[0]
            if bad:
                fails.append(dict(cfg, rank=rank, error=err or "(another rank)"))
                if err:
                    print("FAIL", json.dumps(fails[-1]), flush=True)
            n_cases += 1
        q.put((rank, dict(cases=n_cases, failures=len(fails), worst=worst)))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=90)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--transport", choices=["ipc", "fake-rccl"], default="ipc")
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = f"/tmp/fuzz_partition_id_{os.getpid()}"
    procs = [ctx.Process(target=worker, args=(r, args.world, idfile, q, args.seconds, args.seed, args.transport)) for r in range(args.world)]
    for p in procs:
        p.start()
    res, deadline = [], time.time() + args.seconds + 120
    while len(res) < len(procs) and time.time() < deadline:
        try:
            res.append(q.get(timeout=1.0))
        except Exception:  # noqa: BLE001  (queue.Empty)
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead:  # a rank left through the C++ mirror's print-and-exit: the others would only wait for their deadline
                print(f"rank process(es) exited with {[p.exitcode for p in dead]}: stopping the others", flush=True)
                break
    for p in procs:
        if p.is_alive() and len(res) < len(procs):
            p.terminate()  # (exactly the children started above)
        p.join(timeout=30)
    print(json.dumps(dict(transport=args.transport, world=args.world, ranks=sorted(res, key=lambda r: r[0]))))
    return 0 if all(isinstance(r[1], dict) and r[1]["failures"] == 0 for r in res) else 1


if __name__ == "__main__":
    sys.exit(main())
