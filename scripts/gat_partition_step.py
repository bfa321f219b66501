"""One GAT layer step (64 -> 64, 8 heads, forward + backward) on the reddit-shaped graph: one rank, and R ranks on a
vertex-range partition sharing this box's one GPU over the peer-to-peer transport (each rank owns 1/R of the rows and
needs nearly every other vertex as halo: the vertex order is random).  With the ranks time-sharing one GPU the R-rank step
costs what the R shares cost together, so  step(R) / step(1)  is the overhead of partitioning: halo copies, exchanges,
and the sweep's owned / halo split.  GAIB_OPTS="gat_fused_fwd=0,gat_fused_bwd=0" gives the staged (round-2) pieces.
    python scripts/gat_partition_step.py [ranks=2] [steps=10]"""
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


def worker(rank, world, idfile, steps, q):
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("GAIB_COMM_TIMEOUT_S", "300")
    from graphaibench_amd import capi, layers as L, synth

    ctx = L.init(0)
    for kv in filter(None, os.environ.get("GAIB_OPTS", "").split(",")):
        k, v = kv.split("=")
        ctx.set_option(k.strip(), int(v))
    d, H = 64, 8
    sg = synth.make("reddit", seed=7, device="cuda")
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g1 = g0.add_selfloop()
    g0.close()
    del sg
    n = g1.nv
    if world == 1:
        lg = L.LGraph.adopt(g1)
        lo, hi = 0, n
    else:
        rp = g1.rowptr().cpu().numpy()
        ci = g1.colidx().cpu().numpy().view(np.uint32)
        g1.close()
        torch.cuda.empty_cache()
        if rank == 0:
            uid = capi.comm_unique_id(capi.COMM_IPC)
            open(idfile + ".tmp", "wb").write(uid)
            os.replace(idfile + ".tmp", idfile)
        else:
            while not (os.path.exists(idfile) and os.path.getsize(idfile) == 128):
                time.sleep(0.01)
            uid = open(idfile, "rb").read()
        comm = capi.Comm(ctx, rank, world, uid, capi.COMM_IPC)
        L.set_comm(comm)
        part = L.HostPartition(rp, ci, rank, world, gat=True)
        lo, hi = part.lo, part.hi
        lg = part.make_graph(comm)
        del rp, ci
    nv = hi - lo
    layer = L.Layer(L.GAT, 1, nv, d, d, lg, True)
    layer.set_heads(H)
    torch.manual_seed(3 + rank)
    layer.write(L.FEAT_IN, torch.randn(nv, d, device="cuda"))
    gin = torch.randn(nv, d, device="cuda")
    out, go = torch.empty(nv, d, device="cuda"), torch.empty(nv, d, device="cuda")

    def step():
        layer.write(L.GRAD_IN, gin)
        layer.forward(out)
        layer.backward(out, go)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if world > 1:
        comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        comm.barrier()
    q.put((rank, (time.perf_counter() - t0) / steps * 1e3, nv, lg.ne))


def run(world, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = f"/dev/shm/gaib_gatstep_{os.getpid()}_{world}"
    procs = [ctx.Process(target=worker, args=(r, world, idfile, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1500) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    if os.path.exists(idfile):
        os.unlink(idfile)
    return res


if __name__ == "__main__":
    ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    one = run(1, steps)
    many = run(ranks, steps)
    rec = {"single_rank_ms_per_step": one[0][1], "ranks": ranks, "ms_per_step_slowest_rank": max(r[1] for r in many),
           "rows_per_rank": [r[2] for r in many], "edges_per_rank": [r[3] for r in many],
           "ratio": max(r[1] for r in many) / one[0][1], "opts": os.environ.get("GAIB_OPTS", "")}
    print(json.dumps(rec))
