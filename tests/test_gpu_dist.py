"""GPU suite: the distributed layer path (DistLayerGraph = rectangular local graph + halo hook +
the single-GPU C++ GCN layer) with 2 processes sharing cuda:0 over gloo, against the oracle's
GLOBAL result.  RCCL itself needs >1 GPU; everything else of the N>1 path runs here."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, arch="gcn"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphaibench_amd import dist as gd, layers as L
        from oracle import binding as orc
        from util import random_graph, rel_err

        ctx = L.init(0)
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
        b = gd.partition_bounds(n, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = g.rowptr[lo], g.rowptr[hi]
        rp_l = torch.from_numpy((g.rowptr[lo:hi + 1] - e0).astype(np.int64)).cuda()
        ci_g = torch.from_numpy(g.colidx[e0:e1].astype(np.int64)).cuda()
        part = gd.build_partition(rp_l, ci_g, n, rank, world)
        dg = gd.DistLayerGraph(ctx, part)
        layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, 1, hi - lo, D, D, dg.lgraph, True)
        layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
        out = torch.empty(hi - lo, D, device="cuda")
        layer.forward(out)
        assert rel_err(out.cpu().numpy(), want[lo:hi]) < 1e-4
        layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
        grad_out = torch.empty(hi - lo, D, device="cuda")
        layer.backward(out, grad_out)
        assert rel_err(grad_out.cpu().numpy(), want_go[lo:hi]) < 1e-4
        which = [L.W_NEIGH_GRAD] if arch == "gcn" else [L.W_NEIGH_GRAD, L.W_SELF_GRAD]
        gd.allreduce_layer_grads(ctx, layer, which, (D, D))
        want_wg = lo_.W_grad if arch == "gcn" else lo_.W_neigh_grad
        assert rel_err(layer.tensor(L.W_NEIGH_GRAD, (D, D)).cpu().numpy(), want_wg) < 1e-4
        if arch == "sage":  # the self term rode on the halo half of the aggregation (gaib_spmm_gemm2 + GAIB_ACCUMULATE)
            assert rel_err(layer.tensor(L.W_SELF_GRAD, (D, D)).cpu().numpy(), lo_.W_self_grad) < 1e-4
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("arch", ["gcn", "sage"])
def test_two_ranks_on_one_gpu_match_global_oracle(arch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + (0 if arch == "gcn" else 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, arch)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("backend,scaling", [("gloo", "weak"), ("ipc", "weak"), ("ipc", "strong"), ("rccl", "weak")])
def test_bench_multi_rank_path_on_one_gpu(tmp_path, backend, scaling):
    """bench.py's N>1 leg end to end (graph generator with cut edges, partition, split aggregation, dW all-reduce,
    JSON line) with 2 ranks sharing cuda:0 at 2 % scale: over torch.distributed/gloo (host-staged), and over the
    C ABI's gaib_comm (hipIpc peer-to-peer pull) in both scaling modes.  Every field the record promises is there.
    backend rccl (the default of a real run): RCCL refuses two ranks on one device, so here it exercises the set-up
    fallback -- all ranks agree that RCCL is out and move to the peer-to-peer pull transport together."""
    import json
    import subprocess

    port = 29900 + (os.getpid() % 90) + {"gloo": 0, "ipc": 100, "rccl": 300}[backend] + (200 if scaling == "strong" else 0)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GAIB_DIST_BACKEND=backend, GAIB_COMM_TIMEOUT_S="120")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                                       "1", "--scale", "0.02", "--no-cpu-baseline", "--scaling", scaling], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    cfg = res["config"]
    assert res["n_gpus"] == 2 and res["value"] > 0 and cfg["halo_rows_total"] > 0
    assert res["scaling"] == scaling and res["roofline"]["achieved"] > 0
    want = {"ipc": "gaib_comm/ipc", "gloo": "torch.distributed/gloo", "rccl": "gaib_comm/ipc (after rccl failed"}[backend]
    assert cfg["transport"].startswith(want), cfg["transport"]
    assert cfg["rccl_ranks"] == 0 and "xgmi_link_probe" in cfg  # one GPU here: no RCCL ranks, no link to probe
    assert cfg["halo_exchange_standalone_ms"] > 0 and cfg["halo_bytes_per_step_total"] > 0
    # round 5: whichever mode is `value`, the other is a sub-record of the same invocation, and north_star's curve (the N = 1
    # bench graph partitioned N ways) is under config.strong_products either way
    sp, wk = cfg["strong_products"], cfg["weak_products_range"]
    for rec in (sp, wk):
        assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["halo_bytes_per_step_total"] > 0
        assert rec["halo_exchange_standalone_ms"] > 0 and rec["partition_mode_rank0"]["mode"] in ("split", "classes", "onepass")
    assert sp["one_rank_same_graph"]["value"] > 0 and sp["speedup_vs_n1"] == pytest.approx(sp["value"] / sp["one_rank_same_graph"]["value"])
    assert 0.4 < sp["cut_fraction_measured"] < 0.6 and 0.05 < wk["cut_fraction_measured"] < 0.15
    assert cfg["ranks_share_device"] is True and res["roofline"]["ranks_share_device"] is True
    assert "skipped" in cfg["xgmi_link_probe"] and cfg["comm_init_timed_out"] == []
    # both ends of the weak case's partition-quality axis in the same invocation
    assert cfg["random_order"]["cut_fraction"] == 0.5 and cfg["random_order"]["value"] > 0
    assert cfg["random_order"]["halo_rows_total"] >= wk["halo_rows_total"]  # (equal if both halos are complete at this scale)
    if scaling == "weak":
        assert cfg["cut_fraction"] == 0.1 and res["value"] == wk["value"]
    else:
        assert cfg["cut_fraction"] == 0.5 and res["value"] == sp["value"]
    assert res["cpu_baseline"] == {"skipped": "--no-cpu-baseline"}


def _ab_worker(rank, world, port, q):
    """the A/B legs of the N > 1 bench (dist.BenchCase.cu_reserve_ab / transport_ab) with the RCCL BRANCH of comm.hip carrying the
    case -- bound to tests/fake_rccl's strict double, this box has one GPU -- and the peer-to-peer pull as the other transport"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GAIB_RCCL_LIB"] = str(ROOT / "tests" / "fake_rccl" / "librccl_fake.so")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse

        from graphaibench_amd import capi, dist as gd, layers as L, synth

        ctx = L.init(0)
        rccl, err = gd.comm_attempt(ctx, rank, world, capi.COMM_RCCL)
        assert rccl is not None, err
        ipc, err = gd.comm_attempt(ctx, rank, world, capi.COMM_IPC)
        assert ipc is not None, err
        # the communicator's default: 32 CUs left to RCCL's kernels with more than one rank; an explicit 0 stays 0
        assert ctx.get_option("comm_reserve_cus") == 32 and ctx.get_option("comm_reserve_cus_raw") == -1
        args = argparse.Namespace(steps=2, warmup=1, scale=0.02)
        rows = synth.block_rows("ogbn-products", rank, world, seed=42, cut_fraction=0.1, device="cuda", scale=0.02, selfloops=True,
                                boundary="clustered", band=0.2)  # (interior rows: the class modes, whose fused pass overlaps the exchange)
        case = gd.BenchCase(ctx, rccl, args, rank, world, 128, lambda *a: None, rows, "ab test")
        res = case.measure()
        ab = case.cu_reserve_ab(steps=2)
        hp = case.halo_pipeline_ab(pieces=(1, 3), steps=2)  # (over the RCCL branch: one group per slice)
        assert "skipped" in hp or (hp["slices_consumed"] == {"1": 1, "3": 3} and hp["inf_vs_one_slice"]["3"] <= 1e-5), hp
        tb = case.transport_ab({"rccl": rccl, "ipc": ipc, "absent": (None, "not built")})
        after = (ctx.get_option("comm_reserve_cus"), ctx.get_option("comm_reserve_cus_raw"))
        ctx.set_option("comm_reserve_cus", 0)
        zero = ctx.get_option("comm_reserve_cus")
        ctx.set_option("comm_reserve_cus", 100000)
        clamped = ctx.get_option("comm_reserve_cus")
        ctx.set_option("comm_reserve_cus", -1)
        case.close()
        ipc.close()
        rccl.close()
        q.put((rank, "ok", dict(mode=res["partition_mode"]["mode"], ab=ab, tb=tb, after=after, zero=zero, clamped=clamped,
                                num_cus=ctx.get_option("num_cus"), reserve_after_close=ctx.get_option("comm_reserve_cus"))))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_ab_legs_through_the_rccl_branch():
    """VERDICT r4 #2 "done" line: the CU-reserve A/B and the transport A/B with the RCCL branch (strict double) carrying the plan"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 200)
    procs = [ctx.Process(target=_ab_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    for _, _, d in res:
        assert d["mode"] in ("classes", "onepass")  # the fused interior pass overlaps the exchange: the option reaches it
        assert sorted(d["ab"]["ms_per_step"]) == ["0", "32", "64"] and all(v > 0 for v in d["ab"]["ms_per_step"].values())
        assert d["ab"]["fused_launches_overlap_an_exchange"] is True and d["ab"]["in_effect_for_the_timed_steps"] == 32
        assert d["after"] == (32, -1)  # the option went back to "unset": the communicator's default applies again
        assert d["tb"]["rccl"]["exchange_standalone_ms"] > 0 and d["tb"]["ipc"]["exchange_standalone_ms"] > 0
        assert d["tb"]["absent"] == {"skipped": "not built"}
        assert d["zero"] == 0 and d["clamped"] == d["num_cus"] - 64  # an explicit 0 is 0; the fused kernel keeps >= 64 CUs
        assert d["reserve_after_close"] == 0  # RCCL's kernels are gone with its communicator


def _direct_worker(rank, world, port, q):
    """round 5: a partition whose ranges need all of each other's rows (dense graph -> complete halos, send lists = the peers' whole
    row ranges): over the RCCL branch (strict double) the rows leave STRAIGHT from the layer's matrix -- no pack, no send buffer --
    and the layer's results are bit-identical to the same partition over the peer-to-peer pull (which packs)"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GAIB_RCCL_LIB"] = str(ROOT / "tests" / "fake_rccl" / "librccl_fake.so")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphaibench_amd import capi, dist as gd, layers as L
        from oracle import binding as orc
        from util import random_graph, rel_err

        ctx = L.init(0)
        rp, ci = random_graph(2400, 70, seed=21, power_law=True, hub_deg=1500)
        g = orc.Graph(rp, ci).add_selfloop()
        n, D = g.nv, 128
        x = np.random.default_rng(5).standard_normal((n, D)).astype(np.float32)
        gin = np.random.default_rng(6).standard_normal((n, D)).astype(np.float32)
        lo_ = orc.GCNLayer(1, g, D, D, True)
        want = lo_.forward(x)
        want_go = lo_.backward(gin.copy())
        b = gd.partition_bounds(n, world)
        lo, hi = b[rank], b[rank + 1]
        e0, e1 = g.rowptr[lo], g.rowptr[hi]
        rp_l = torch.from_numpy((g.rowptr[lo:hi + 1] - e0).astype(np.int64)).cuda()
        ci_g = torch.from_numpy(g.colidx[e0:e1].astype(np.int64)).cuda()
        outs = {}
        for name, tr in (("rccl", capi.COMM_RCCL), ("ipc", capi.COMM_IPC)):
            comm, err = gd.comm_attempt(ctx, rank, world, tr)
            assert comm is not None, err
            part = gd.build_partition(rp_l, ci_g, n, rank, world)
            assert part.n_halo == n - (hi - lo)  # complete halos
            dg = gd.DistLayerGraph(ctx, part, comm)
            layer = L.Layer(L.GCN, 1, hi - lo, D, D, dg.lgraph, True)
            layer.write(L.FEAT_IN, torch.from_numpy(x[lo:hi]).cuda())
            out = torch.empty(hi - lo, D, device="cuda")
            layer.forward(out)
            layer.write(L.GRAD_IN, torch.from_numpy(gin[lo:hi]).cuda())
            go = torch.empty(hi - lo, D, device="cuda")
            layer.backward(out, go)
            torch.cuda.synchronize()
            st = dg.ex.halo.send_stats()
            outs[name] = (out.clone(), go.clone(), st)
            plan = dg.ex.halo
            layer.close()
            dg.lgraph.close()
            plan.close()
            comm.close()
        (o_r, g_r, st_r), (o_i, g_i, st_i) = outs["rccl"], outs["ipc"]
        assert st_r["direct_peers"] == world - 1 and st_r["packs"] == 0 and st_r["direct_sends"] >= 2 * (world - 1), st_r
        assert st_i["direct_peers"] == 0 and st_i["packs"] >= 2 and st_i["direct_sends"] == 0, st_i
        assert torch.equal(o_r, o_i) and torch.equal(g_r, g_i)
        assert rel_err(o_r.cpu().numpy(), want[lo:hi]) < 1e-4 and rel_err(g_r.cpu().numpy(), want_go[lo:hi]) < 1e-4
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_complete_halo_rows_leave_straight_from_the_matrix_over_rccl(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29150 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                                              "MASTER_PORT", "GAIB_DIST_BACKEND", "GAIB_FORCE_DIST")}


def test_bench_plain_invocation_config5_shape_matches_global_oracle():
    """BASELINE config 5 as a test (reference shape: include/gnn/configs.h:8-11, "ogbn-papers100M"): `python bench.py
    --gpus 4 --workload gcn-papers` INVOKED PLAINLY -- no ranks from outside, bench.py starts and supervises them -- on the
    papers100M-shaped graph at 1/50 (2.2 M vertices, 64 M edges) as 4 vertex ranges / 4 processes sharing this box's GPU
    over the peer-to-peer transport, both ends of the partition-quality axis; every rank's forward output and input
    gradient and the summed weight gradient are compared ELEMENT-WISE with the oracle's run on the global graph."""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--workload", "gcn-papers", "--scale", "0.04",
                        "--steps", "2", "--warmup", "1", "--check-oracle", "--deadline-s", "1000"],
                       capture_output=True, text=True, timeout=1100, env=dict(_clean_env(), GAIB_COMM_TIMEOUT_S="300"))
    assert r.returncode == 0, r.stderr[-4000:]
    line = r.stdout.strip().splitlines()[-1]
    res = json.loads(line)
    cfg, par = res["config"], res["parity"]
    assert res["n_gpus"] == 4 and res["value"] > 0 and "config 5" in cfg["workload"]
    # (the kept 4-rank records of round 4 had cpu_baseline null -- they were taken with --no-cpu-baseline, which
    # now says {"skipped": ...}; a 4-rank run without the flag carries the baseline)
    assert res["cpu_baseline"] is not None and res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["kind"] == "port"
    assert cfg["ranks_share_device"] is (torch.cuda.device_count() < 4) and res["scaling"] == "weak"
    assert cfg["launcher"].startswith("bench.py itself")
    assert cfg["nv_per_gpu"] * 4 == pytest.approx(111_059_956 / 50, rel=0.01)
    assert cfg["transport"].startswith("gaib_comm/ipc") and cfg["rccl_ranks"] == 0  # 4 ranks on one device
    assert cfg["cut_fraction"] == 0.1 and cfg["random_order"]["cut_fraction"] == 0.75
    assert cfg["random_order"]["halo_rows_total"] > cfg["halo_rows_total"] > 0
    for rec in (par, par["random_order"]):
        assert rec["ok"], rec
        assert max(rec["forward"]["elem"], rec["forward"]["inf"], rec["grad_out"]["elem"], rec["grad_out"]["inf"],
                   rec["W_grad"]["elem"], rec["W_grad"]["inf"]) <= 1e-4, rec
        assert "GLOBAL graph" in rec["against"]
    assert par["ok"]
    bd, pm = cfg["breakdown_ms_per_step_rank0"], cfg["partition_mode_rank0"]
    assert bd["pack_ms"] > 0 and pm["mode"] in ("split", "classes", "onepass") and 0 < pm["boundary_rows"] <= cfg["nv_per_gpu"]
    if pm["mode"] == "split":
        assert bd["owned_edge_spmm_ms"] > 0 and bd["halo_half_ms"] > 0
    else:  # row classes: the interior pass and one of the boundary forms
        assert bd["part_fused_ms"] > 0 and (bd.get("part_fused_2t_ms", 0) > 0 or bd.get("part_fused_acc_ms", 0) > 0)


def test_bench_plain_two_gpus_default_workload():
    """VERDICT r2 "done" line: `python3 bench.py --gpus 2 --scale 0.02` with a clean environment exits 0 on a one-GPU
    box (peer-to-peer transport, named in config.transport) and prints one JSON line"""
    import json
    import subprocess

    # GAIB_BENCH_CONFIG5=force: the sub-record the default 8-GPU run adds (config 5's graph in the same invocation), here at 2 ranks
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=dict(_clean_env(), GAIB_BENCH_CONFIG5="force"))
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-4000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1
    res = json.loads(out[0])
    # every sub-case is budgeted: the whole run stays inside --budget-s (default 420 s, under the driver's 600 s)
    bud = res["config"]["budget"]
    assert wall < bud["budget_s"] and bud["elapsed_s"] < bud["budget_s"] and "partial" not in res
    # the sub-records of the default run: the clustered-boundary generator next to the uniform one, the N = 1 CPU baseline
    cfg = res["config"]
    # round 5: `value` is north_star's curve -- the N = 1 bench graph partitioned N ways -- and the record says so;
    # config.strong_products names it with the one-rank timing of the same graph taken in the run; the weak case runs second
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and cfg["transport"].startswith("gaib_comm/")
    sp, wk = cfg["strong_products"], cfg["weak_products_range"]
    assert sp["value"] == res["value"] and sp["ms_per_step"] == res["ms_per_step"] and "partitioned into 2 vertex ranges" in sp["workload"]
    assert sp["halo_bytes_per_step_total"] > 0 and sp["halo_exchange_standalone_ms"] > 0 and sp["partition_mode_rank0"]["mode"]
    assert sp["one_rank_same_graph"]["ms_per_step"] > 0 and sp["speedup_vs_n1"] > 0
    assert wk["value"] > 0 and wk["cut_fraction"] == 0.1 and 0 < wk["halo_rows_total"]
    cl = cfg["clustered_boundary"]
    assert cl["value"] > 0 and 0 < cl["halo_rows_total"] < wk["halo_rows_total"]
    assert cl["partition_mode_rank0"]["boundary_row_share"] < wk["partition_mode_rank0"]["boundary_row_share"]
    # the strong case's comparison is the oracle's run on the WHOLE bench graph: its timing is the N = 1 workload's CPU baseline
    assert res["cpu_baseline"]["value"] > 0 and "N = 1 workload" in res["cpu_baseline"]["of"] and "whole" in res["cpu_baseline"]["sample"]
    par = res["parity"]
    assert par["ok"] is True and par["forward"]["elem"] <= 1e-4 and par["grad_out"]["elem"] <= 1e-4 and par["W_grad"]["inf"] <= 1e-4
    assert "GLOBAL graph" in par["against"] and "cpu_baseline" not in par
    # round 4's legs on the weak generator stay: every rank's rows against the oracle's GLOBAL run on a leg of bounded size
    pw = par["weak_generator_scaled"]
    assert pw["ok"] is True and pw["forward"]["elem"] <= 1e-4 and pw["scale"] <= 0.02 and "2 ranks" in pw["of"]
    pc = pw["clustered_boundary"]  # ... and on the clustered generator, through the row classes
    assert pc["ok"] is True and pc["forward"]["elem"] <= 1e-4 and pc["partition_mode_rank0"] in ("classes", "onepass")
    # the run measures its own constants -- on a one-device box each slot says why it did not
    if torch.cuda.device_count() < 2:
        assert "skipped" in cfg["xgmi_link_probe"] and "skipped" in cfg["cu_reserve_ab"] and "skipped" in cfg["halo_pipeline_ab"]
        assert cfg["transport_ab"]["carried_the_run"] == "ipc" and cfg["transport_ab"]["ipc"]["exchange_standalone_ms"] > 0
        assert "share devices" in cfg["transport_ab"]["rccl"]["skipped"]
        assert cfg["ranks_share_device"] is True and res["roofline"]["ranks_share_device"] is True
        assert "library's default" in cfg["partition_mode_rank0"]["link_gbs_source"]
    assert cfg["comm_init_timed_out"] == []
    if torch.cuda.device_count() < 2:
        assert res["config"]["transport"].startswith("gaib_comm/ipc") and res["config"]["rccl_ranks"] == 0
    else:
        assert res["config"]["rccl_ranks"] in (0, 2)
    c5 = res["config"]["config5_papers100M"]
    assert c5["value"] > 0 and c5["halo_rows_total"] > 0 and "config 5" in c5["workload"]
    assert c5["nv_per_gpu"] == int(13_882_495 * 0.02)


def test_bench_ab_legs_bookkeeping_on_a_shared_device():
    """the A/B legs of the run's own constants (config.cu_reserve_ab, config.transport_ab) with their bookkeeping forced on a
    one-device box (GAIB_BENCH_AB=force; 3 ranks): three timings of the headline step at 0 / 32 / 64 reserved CUs, the option back
    where it was, the plan re-timed over the transport that carried the run, both stamped as taken on a shared device"""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-parity"], capture_output=True, text=True, timeout=900,
                       env=dict(_clean_env(), GAIB_BENCH_AB="force", GAIB_BENCH_CLUSTERED="0", GAIB_BENCH_RANDOM_ORDER="0",
                                GAIB_PART_MODE="split"))  # (a mode with a halo-column half: what the slices are consumed by)
    assert r.returncode == 0, r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    cfg = res["config"]
    ab = cfg["cu_reserve_ab"]
    assert sorted(ab["ms_per_step"]) == ["0", "32", "64"] and all(v > 0 for v in ab["ms_per_step"].values())
    assert ab["ranks_share_device"] is True and ab["in_effect_for_the_timed_steps"] == cfg["cu_reserve_for_transport"] == 0
    assert ab["mode"] == cfg["partition_mode_rank0"]["mode"]
    # round 6: the exchange in 1 / 2 / 4 / 8 time slices on the live headline case; the plan back where it was; the same sums
    hp = cfg["halo_pipeline_ab"]
    assert sorted(hp["ms_per_step"], key=int) == ["1", "2", "4", "8"] and all(v > 0 for v in hp["ms_per_step"].values())
    assert hp["slices_consumed"] == {"1": 1, "2": 2, "4": 4, "8": 8} and hp["ranks_share_device"] is True
    assert all(v <= 1e-5 for v in hp["inf_vs_one_slice"].values()) and sorted(hp["inf_vs_one_slice"], key=int) == ["2", "4", "8"]
    assert hp["in_effect_for_the_timed_steps"] == {"plan": 1, "consumed_rank0": 1} and cfg["halo_pieces_rank0"] == {"plan": 1, "consumed": 1}
    assert hp["mode"] == cfg["partition_mode_rank0"]["mode"]
    tb = cfg["transport_ab"]
    assert tb["ipc"]["exchange_standalone_ms"] > 0 and "skipped" in tb["rccl"] and tb["ranks_share_device"] is True
    assert res["n_gpus"] == 3 and cfg["strong_products"]["value"] == res["value"]


@pytest.mark.parametrize("boundary", ["uniform", "clustered"])
def test_block_generator_gives_the_same_edges_on_every_call_and_a_symmetric_global_graph(boundary):
    """synth.block_rows: a range generated twice (a rank's own copy, and the one rank 0 makes for the oracle's global run) must
    be the same edges bit for bit, and the cross block (p, q) must be the same seen from p and from q.  Its cdf used to come from
    a device scan whose floating-point association depends on tile timing: with four busy ranks on one device a handful of
    sampled edges differed between two generations in some runs (found by the N > 1 parity leg, round 4)"""
    from graphaibench_amd import synth

    world, scale = 4, 0.25
    gens = [[synth.block_rows("ogbn-products", q, world, seed=42, cut_fraction=0.1, device="cuda", scale=scale, selfloops=True,
                              boundary=boundary) for q in range(world)] for _ in range(2)]
    for a, b in zip(*gens):
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.colidx_global, b.colidx_global)
    nv = gens[0][0].n_local
    n = nv * world
    keys = []
    for q, r in enumerate(gens[0]):
        rows = torch.repeat_interleave(torch.arange(nv, device="cuda"), r.rowptr[1:] - r.rowptr[:-1]) + q * nv
        keys.append(rows * n + r.colidx_global)
    key = torch.cat(keys)
    key_t = (key % n) * n + key // n
    assert torch.equal(torch.sort(key).values, torch.sort(key_t).values)  # the global graph is symmetric


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_bench_keeps_the_headline_when_a_rank_fails_in_a_sub_case(failing_rank):
    """a rank raises after the headline case (injected: GAIB_BENCH_FAIL_AFTER_HEADLINE = its rank) while its peer goes on into
    the next sub-case's collectives: rank 0 prints the record it holds, marked partial (its own exception, or the SIGTERM the
    launcher sends when rank 1 exits non-zero), the launcher ends the stragglers, one JSON line, status 0, well inside a minute"""
    import json
    import subprocess
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=dict(_clean_env(), GAIB_BENCH_FAIL_AFTER_HEADLINE=str(failing_rank)))
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-4000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and wall < 120, (len(out), wall)
    res = json.loads(out[0])
    assert res["value"] > 0 and res["n_gpus"] == 2 and "partial" in res
    # rank 0's own failure is named; a peer's shows up as the signal from the launcher or as the control plane's broken connection
    assert ("injected failure" in res["partial"]["reason"]) == (failing_rank == 0) and res["partial"]["reason"]
    assert res["config"]["clustered_boundary"] is None  # (never started)


def test_bench_n1_keeps_its_record_when_a_later_leg_fails():
    """N = 1: the record is held once the GPU measurement is complete; an exception in what follows (the CPU baseline, the
    comparison with the oracle -- injected here) prints it marked partial, status 0, one line"""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--scale", "0.02", "--steps", "2", "--warmup", "1", "--sustain-s", "0"],
                       capture_output=True, text=True, timeout=600, env=dict(_clean_env(), GAIB_BENCH_FAIL_AFTER_HEADLINE="0"))
    assert r.returncode == 0, r.stderr[-4000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1
    res = json.loads(out[0])
    assert res["value"] > 0 and res["n_gpus"] == 1 and "injected failure" in res["partial"]["reason"]
    assert res["roofline"]["avg_launch_ms"] > 0 and "cpu_baseline" not in res
    assert res["parity"]["ok"] is None and "did not complete" in res["parity"]["reason"]  # unchecked is SAID


def test_bench_n1_record_at_small_scale():
    """`python bench.py --scale 0.05`: the N = 1 record with every block the contract names, element-wise parity, and -- round 5 --
    the outputs whose relu mask differs from the oracle's checked against an fp64 evaluation (inside the fp32 rounding bound of
    their sums: both signs are correct roundings)"""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--scale", "0.05", "--steps", "3", "--warmup", "1", "--sustain-s", "0"],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1
    res = json.loads(out[0])
    assert res["n_gpus"] == 1 and res["scaling"] == "strong" and res["unit"] == "edges/s" and res["dtype"] == "f32" and res["vs_baseline"] is None
    assert res["roofline"]["bound"] == "hbm" and res["roofline"]["frac"] > 0 and res["cpu_baseline"]["value"] > 0
    par = res["parity"]
    assert par["ok"] is True and par["forward"]["elem"] <= 1e-4 and par["grad_out"]["elem"] <= 1e-4
    fl = par["relu_mask_flips"]
    assert "fp64" in fl and fl["fp64"].get("ok", True) is True
    if fl["count"]:
        assert fl["fp64"]["checked"] >= 1 and fl["fp64"]["max_abs_fp64_value_over_fp32_rounding_bound"] <= 1.0
    # round 6: BASELINE configs 2-4 as short legs of the default run, after the headline record is complete
    oc = res["other_configs"]
    for k in ("sage_layer_128", "sage_layer_256", "gat_layer_reddit_8x8"):
        assert oc[k]["value"] > 0 and oc[k]["ms_per_step"] > 0 and oc[k]["steps"] == 10, (k, oc[k])
    assert 0 < oc["sage_layer_256"]["roofline"]["frac"] < 1.5 and oc["sage_layer_256"]["roofline"]["per_key"]
    for k, hid in (("epoch_sage_products_hidden256", 256), ("epoch_sage_products_hidden128", 128), ("epoch_gcn_products", 128),
                   ("epoch_gat_reddit", 64), ("epoch_gcn_cora", 16)):
        e = oc[k]
        assert e["value"] > 0 and e["ms_per_epoch"] > 0 and e["hidden"] == hid and e["steps"] == 10 and e["roofline"]["frac"] > 0, (k, e)
        assert len(e["train_loss_timed_epochs"]) == 10 and e["train_loss_timed_epochs"][-1] < e["train_loss_timed_epochs"][0]
    assert oc["epoch_gcn_cora"]["recorded_epochs"] is True and oc["elapsed_s"] < oc["budget_s"] + 60


def test_bench_budget_skips_sub_cases_and_keeps_the_headline():
    """--budget-s smaller than any sub-case: the headline case runs, every further slot says {"skipped": "budget", ...},
    exit 0, one line"""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                        "--budget-s", "1"], capture_output=True, text=True, timeout=900,
                       env=dict(_clean_env(), GAIB_BENCH_CONFIG5="force"))
    assert r.returncode == 0, r.stderr[-4000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1
    res = json.loads(out[0])
    assert res["value"] > 0 and res["ms_per_step"] > 0
    for slot in (res["cpu_baseline"], res["parity"], res["config"]["weak_products_range"], res["config"]["clustered_boundary"],
                 res["config"]["random_order"], res["config"]["config5_papers100M"]):
        assert slot["skipped"] == "budget" and slot["elapsed_s"] > 1 and slot["needed_s_estimate"] > 0, slot


def test_bench_under_torch_distributed_run_prints_one_line():
    """the driver's form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- the ranks come from outside, and the job's stdout is EXACTLY the one JSON line
    (gloo and RCCL print to fd 1 on their own, from every rank: bench.py points fd 1 at stderr and writes the record to the
    saved original)"""
    import json
    import subprocess

    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                        "1", "--scale", "0.02"], capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["launcher"].startswith("ranks given from outside")
    assert "Gloo" in r.stderr or True  # (the libraries' chatter, if any, went to stderr)


def test_bench_gat_reddit_across_ranks_on_one_gpu():
    """round 6 (VERDICT r5 missing #4): `bench.py --gpus 2 --workload gat-reddit` -- BASELINE config 4's layer on a vertex-range
    partition of the reddit-shaped graph: one JSON line, the one-rank step of the same layer taken in the run, rank 0's rows held to
    the one-rank layer's (forward, input gradient on identical masks, the all-reduced weight and attention-vector gradients)"""
    import json
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "gat-reddit", "--scale", "0.05",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr[-4000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1
    res = json.loads(out[0])
    cfg, par = res["config"], res["parity"]
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["metric"].startswith("GAT-layer") and res["value"] > 0
    assert cfg["transport"].startswith("gaib_comm/") and cfg["heads"] == 8 and cfg["halo_rows_total"] > 0
    assert cfg["one_rank_same_graph"]["ms_per_step"] > 0 and cfg["speedup_vs_n1"] > 0
    assert par["ok"] is True and par["forward"]["inf"] <= 1e-4 and par["grad_out"]["inf"] <= 1e-4 and par["W_grad"]["inf"] <= 1e-4
    assert any(k.startswith("gat_") for k in res["breakdown_ms_per_step_rank0"])
