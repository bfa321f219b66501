#!/usr/bin/env python3
"""bench.py -- GCN-layer forward+backward on an ogbn-products-shaped graph, 1..8 MI355X.

Metric (BASELINE.json): "GCN-layer fwd+bwd: aggregated edges/sec + achieved HBM GB/s".
A step = one pass of the hot path: GCN_layer(128 -> 128, hidden layer, relu)::forward +
::backward on the host C++ layer API (include/layers/graph_conv_layer.h) over the C ABI
(include/gaib.h): 2 SpMM at D=128 + 3 fp32 MFMA GEMMs + relu + d_relu.
`value` = aggregated edges (2 * E per step, E incl. self loops, summed over ranks) / wall time,
inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One JSON line on rank 0.  `roofline` prices the dominant kernel (the one-wave-per-row SpMM,
spmm_w64_kernel) by ALGORITHMIC bytes per launch / mean launch time measured with HIP events in
the timed region; `cpu_baseline` times the oracle's restatement of the OpenMP path on the host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

D = 128  # feature width of the layer under test (north star: D = 128)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured stream copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cores() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(sg_rowptr, sg_colidx, nv, budget_s=15.0):
    """The oracle (port of the reference's OpenMP GCN layer, oracle/gnn_oracle.c) timed on the host
    cores on a bounded row-prefix sample of the SAME graph: rows [0, R) forward + backward."""
    import numpy as np
    from oracle import binding as orc

    cores = usable_cores()
    orc.set_threads(cores)
    rp = sg_rowptr.cpu().numpy()
    ci = sg_colidx.cpu().numpy().view(np.uint32)
    g_full = orc.Graph(rp, ci).add_selfloop()  # net.cpp:96
    vd = g_full.vertex_data()
    rng = np.random.default_rng(43)
    W = orc.init_glorot(D, D, 1)

    libs = {"lib": orc.lib()}

    def run(R):
        g = orc.Graph.__new__(orc.Graph)
        g.rowptr, g.colidx, g.nv, g.ne, g.vd = g_full.rowptr[:R + 1], g_full.colidx, R, int(g_full.rowptr[R]), vd
        layer = orc.GCNLayer(1, g, D, D, True, W=W)
        # feature tables keep ALL rows: the aggregation gathers any column
        x = rng.standard_normal((nv, D), dtype=np.float32)
        layer.in_temp = np.zeros((nv, D), np.float32)
        gin = rng.standard_normal((R, D), dtype=np.float32)
        t0 = time.perf_counter()
        import ctypes as C
        s = g._struct()
        out = np.empty((R, D), np.float32)
        lib = libs["lib"]
        p = orc._p
        lib.orc_gcn_layer_forward(C.byref(s), C.c_int(D), C.c_int(D), C.c_int(1), p(x), p(W), p(layer.in_temp1[:R]),
                                  p(layer.out_temp[:R]), p(out))
        grad_out = np.zeros((R, D), np.float32)
        lib.orc_gcn_layer_backward(C.byref(s), C.c_int(1), C.c_int(D), C.c_int(D), C.c_int(1), p(x), p(W), p(out),
                                   p(gin), p(layer.in_temp), p(layer.in_temp1[:R]), p(layer.out_temp[:R]),
                                   p(grad_out), p(layer.W_grad))
        return time.perf_counter() - t0, 2 * g.ne

    probe_R = max(nv // 64, 1024)
    run(min(probe_R, nv))  # thread spin-up: the first pass is ~2x slower
    t_probe, e_probe = run(min(probe_R, nv))
    rate = e_probe / max(t_probe, 1e-9)
    total_edges = 2 * g_full.ne
    R = nv if total_edges / rate <= budget_s else max(int(nv * budget_s * rate / total_edges), probe_R)
    R = min(R, nv)
    t, e = run(R)
    res = dict(value=e / t, unit="edges/s", cores=cores, kind="port",
               sample=f"rows [0,{R}) of the same graph ({e // 2} edges incl. self loops), 1 layer fwd+bwd, "
                      f"{t:.2f} s, gcc -O3 -fopenmp no -march=native (reference Makefile flags)")
    nat = orc.native_lib()  # second figure with -march=native built on this host, same sample
    if nat is not None:
        libs["lib"] = nat
        run(min(probe_R, nv))
        tn, en = run(R)
        res["value_march_native"] = en / tn
    return res


def emit(result: dict) -> None:
    """the ONE JSON line, last on stdout: libraries that write to C stdio (RCCL prints its library path
    there) sit in libc's buffer until exit and would otherwise land after it"""
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cut-fraction", type=float, default=None,
                    help="N>1: fraction of each partition's edges that cross partitions")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)

    from graphaibench_amd import capi, layers as L, synth

    ctx = L.init(local_rank)

    if world > 1 or os.environ.get("GAIB_FORCE_DIST") == "1":  # the env knob runs the N>1 code on one GPU
        import torch.distributed as dist
        from graphaibench_amd import dist as gdist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("GAIB_DIST_BACKEND", "nccl")  # "gloo": several ranks on ONE GPU (tests)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        result = gdist.bench_gcn_layer(ctx, args, rank, world, D, log)
        dist.barrier()
        if rank == 0:
            emit(result)
        dist.destroy_process_group()
        return

    # ---------------- single GPU -----------------------------------------------------------------
    t0 = time.time()
    sg = synth.make("ogbn-products", seed=42, device="cuda", scale=args.scale)
    torch.cuda.synchronize()
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g1 = g0.add_selfloop()  # GCN aggregates over A + I (net.cpp:96)
    g0.close()
    ctx.sync()
    nv, ne = g1.nv, g1.ne
    stats = ctx.graph_stats(g1)
    log(f"[bench] products-shaped graph: nv={nv} ne={ne} (incl. self loops) max_deg={stats['max_degree']} "
        f"heavy rows={stats['n_heavy']} holding {stats['heavy_edges']} edges; gen+upload {time.time()-t0:.1f}s")
    lg = L.LGraph.adopt(g1)
    gview = lg.device_graph()
    torch.manual_seed(43)
    layer = L.Layer(L.GCN, 1, nv, D, D, lg, act=True, lr=0.01)
    layer.write(L.FEAT_IN, torch.randn(nv, D, device="cuda"))
    layer.write(L.GRAD_IN, torch.randn(nv, D, device="cuda"))
    feat_out = torch.empty(nv, D, device="cuda")
    grad_out = torch.empty(nv, D, device="cuda")

    def step():
        layer.forward(feat_out)
        layer.backward(feat_out, grad_out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    ctx.prof_enable(False)
    n_light, ms_light = ctx.prof_get("spmm_light")
    n_heavy, ms_heavy = ctx.prof_get("spmm_heavy")
    n_gemm, ms_gemm = ctx.prof_get("sgemm")
    n_fused, ms_fused = ctx.prof_get("spmm_gemm_fused")
    ctx.prof_reset()

    edges_per_step = 2 * ne
    value = edges_per_step * args.steps / elapsed
    # algorithmic bytes of ONE launch of the dominant kernel (light rows of one SpMM), SURVEY 8d:
    #   E*(4D + 4 colidx + 4 per-edge weight) + N*4D written + (N+1)*8 rowptr
    e_light = ne - stats["heavy_edges"]
    n_light_rows = nv - stats["n_heavy"]
    if n_fused > 0:
        # the layer's two aggregations run as spmm_gemm_kernel (dense product fused in): per launch the same
        # gathers + the rows it stores -- forward: A.X and the layer output, backward: the input gradient
        # (average 1.5 N x D matrices; the heavy rows' aggregates and the 64 KB of W are noise)
        kernel_name = "spmm_gemm_kernel<VEC=2,edge-weights,U=16,buffer> (aggregation + MFMA dense product)"
        traffic_key = "spmm_gemm_kernel_bytes_per_launch"
        alg_bytes = e_light * (4 * D + 4 + 4) + int(1.5 * nv * 4 * D) + (nv + 1) * 8
        n_dom, ms_dom = n_fused, ms_fused
    else:
        kernel_name = "spmm_w64_kernel<VEC=2,CT=1,edge-weights,U=16,buffer>"
        traffic_key = "spmm_w64_kernel_bytes_per_launch"
        alg_bytes = e_light * (4 * D + 4 + 4) + n_light_rows * 4 * D + (nv + 1) * 8
        n_dom, ms_dom = n_light, ms_light
    avg_ms = ms_dom / max(n_dom, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = None
    tf = ROOT / "profiles" / "hbm_traffic.json"
    if tf.exists() and args.scale == 1.0:
        try:
            traffic = json.loads(tf.read_text()).get(traffic_key)
        except Exception:
            traffic = None
    result = {
        "metric": "GCN-layer fwd+bwd aggregated edges/sec",
        "value": value,
        "unit": "edges/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "ogbn-products-shaped Chung-Lu graph (seed 42), GCN hidden layer 128->128 fwd+bwd "
                        "(2 SpMM D=128 + 3 dense products + relu/d_relu; 2 of the products ride on the SpMMs)",
            "nv": nv, "ne_with_selfloops": ne, "D": D, "scale": args.scale,
            "parallelism": "1 GPU",
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel_name,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches": n_dom,
            # SURVEY 8d extras: perfect-reuse lower bound of one aggregation (every feature row read once) and the
            # PMC-measured bytes as a fraction of the peak over the same launch time
            "b_min_bytes_per_launch": 2 * nv * 4 * D + 4 * ne,
            "hbm_measured_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        },
        # the aggregation kernels alone (both SpMM launches of the step incl. heavy rows; the dense products that
        # ride on them are inside): aggregated edges per second of kernel time
        "aggregation_only_edges_per_s": edges_per_step * args.steps / max((ms_fused + ms_light + ms_heavy) * 1e-3, 1e-9),
        "breakdown_ms_per_step": {
            "spmm_gemm_fused": ms_fused / args.steps, "spmm_light": ms_light / args.steps,
            "spmm_heavy": ms_heavy / args.steps, "sgemm": ms_gemm / args.steps,
        },
    }
    if not args.no_cpu_baseline:
        t1 = time.time()
        result["cpu_baseline"] = cpu_baseline(sg.rowptr, sg.colidx, sg.nv)
        log(f"[bench] cpu baseline took {time.time()-t1:.1f}s")
    emit(result)


if __name__ == "__main__":
    main()
