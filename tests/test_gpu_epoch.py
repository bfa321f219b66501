"""GPU suite: the epoch-level records (bench.py --workload epoch-*) and the work accounting they are built from
(gaib_prof_get_work / gaib_prof_table: every timed launch states its algorithmic bytes and flops)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from graphaibench_amd import capi, layers as L
from util import random_graph

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_prof_table_states_the_algorithmic_work_of_each_launch():
    """one aggregation at 47 and one at 128 columns + one dense product: the table lists the gather kernels per row width, the
    bytes are SURVEY 8(d)'s formula, the flops 2 M N K, and roof_ms = max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) summed"""
    ctx = L.init(0)
    rp, ci = random_graph(4000, 12, seed=3, power_law=True, hub_deg=1500)
    g0 = ctx.graph(torch.from_numpy(rp.astype(np.int64)).cuda(), torch.from_numpy(ci.astype(np.int32)).cuda())
    st = ctx.graph_stats(g0)
    nv, ne = g0.nv, g0.ne
    ctx.prof_reset()
    ctx.prof_enable(True)
    for d in (47, 128):
        x = torch.randn(nv, d, device="cuda")
        out = torch.empty(nv, d, device="cuda")
        ctx.spmm(g0, capi.W_GCN, x, out)
    A, B, Cm = torch.randn(nv, 128, device="cuda"), torch.randn(128, 64, device="cuda"), torch.empty(nv, 64, device="cuda")
    ctx.sgemm(A, B, Cm)
    ctx.prof_enable(False)
    tab = ctx.prof_table()
    ctx.prof_reset()
    g0.close()
    e_l, r_l = ne - st["heavy_edges"], nv - st["n_heavy"]
    for d in (47, 128):
        light = tab[f"spmm_light@{d}"]
        assert light["count"] == 1 and light["ms"] > 0
        assert light["bytes"] == pytest.approx(e_l * (4 * d + 8) + r_l * 4 * d + (r_l + 1) * 8)
        assert light["flops"] == pytest.approx(2 * e_l * d)
        assert light["roof_ms"] == pytest.approx(1e3 * max(light["bytes"] / 8e12, light["flops"] / 157.3e12), rel=1e-4)
        if st["n_heavy"]:
            hv = tab[f"spmm_heavy@{d}"]
            assert hv["bytes"] == pytest.approx(st["heavy_edges"] * (4 * d + 8) + st["n_heavy"] * 4 * d + (st["n_heavy"] + 1) * 8)
    sg = tab[f"sgemm@{nv}x64x128"]  # (dense products are listed per shape M x N x K)
    assert sg["flops"] == pytest.approx(2.0 * nv * 128 * 64) and sg["bytes"] == pytest.approx(4.0 * (nv * 128 + 128 * 64 + nv * 64))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "GAIB_RANKS", "GAIB_GAT_HEADS", "DATASET_PATH")}


@pytest.mark.parametrize("workload,arch", [("epoch-gcn-products", "gcn"), ("epoch-sage-products", "sage"), ("epoch-gat-reddit", "gat"),
                                           ("epoch-gcn-cora", "cora")])
def test_bench_epoch_workloads_at_small_scale(workload, arch):
    """`python bench.py --workload epoch-*` at 2 % scale: one JSON line with the contract's keys, a roofline whose fraction is the
    launches' time at the roofs over the measured epoch, the line floor next to the 47-wide gathers, a CPU baseline from the
    oracle's Model and the first five train_loss values against it"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", workload, "--scale", "0.02", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["unit"] == "edges/s" and res["value"] > 0 and res["n_gpus"] == 1 and res["steps"] == 3 and res["ms_per_step"] > 0
    cfg, roof = res["config"], res["roofline"]
    assert cfg["aggregated_edges_per_epoch"] > 0 and len(cfg["train_loss_timed_epochs"]) == 3
    assert 0 < roof["frac"] < 1.2 and roof["roof_ms_per_epoch"] == pytest.approx(roof["frac"] * res["ms_per_step"])
    if arch == "cora":
        # BASELINE config 2 on the reference's own topology: launch bound, so the TIMED epochs are the recorded HIP graphs the
        # trainer replays by default and the work table comes from a call-by-call run of the same kernels
        rec = cfg["recorded_epochs"]
        assert rec["call_by_call_ms_per_epoch"] > res["ms_per_step"] > 0 and cfg["nv"] == 2708 and cfg["C"] == 7
        assert res["parity"]["ok"] is True and res["cpu_baseline"]["value"] > 0
        return
    assert roof["timed_launches_ms_per_epoch"] <= res["ms_per_step"] * 1.01
    keys = roof["per_key"]
    gemms = [k for k in keys if k.startswith("sgemm@")]
    assert gemms and all(keys[k]["flops_per_epoch"] > 0 for k in gemms)
    if arch in ("gcn", "sage"):  # the 47-wide output layer's gathers: two 128-B lines per 188-B row
        k47 = [k for k in keys if k.endswith("@47")]
        assert k47, list(keys)
        lf = keys[k47[0]]["line_floor"]
        assert lf["lines_per_row"] == 2.0 and lf["bytes_per_edge_in_lines"] == 264 and lf["bytes_per_edge_algorithmic"] == 196
    else:
        assert any(k.startswith("gat_fwd_fused") or k.startswith("gat_edge_softmax") for k in keys), list(keys)
    assert res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["kind"] == "port" and "oracle" in res["cpu_baseline"]["sample"]
    par = res["parity"]
    assert par["ok"] is True and par["max_abs_loss_diff"] <= 2e-3 and len(par["train_loss_gpu"]) == 5
