"""Layer-level timing of the other BASELINE.json configs on one MI355X (development aid):
  * GraphSAGE layer 128->128 fwd+bwd on the products-shaped graph (config 3, scripts/run-sage-products.sh shape)
  * GAT layer 64->64 fwd+bwd on a reddit-shaped graph (config 4, single head like the reference)
with per-kernel HIP-event times (gaib_prof_*) and the oracle on a bounded sample beside them.
    python scripts/microbench_layers.py [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import layers as L, synth  # noqa: E402

KEYS = ["spmm_gemm_fused", "spmm_light", "spmm_heavy", "spmm_sub", "spmm_chunk", "spmm_chunk_reduce", "sgemm", "relu", "d_relu", "gat_vertex_dots", "gat_edge_softmax",
        "gat_sddmm", "gat_softmax_bwd_alpha"]


def run_layer(ctx, kind, name, graph_name, din, dout, selfloop, steps, scale, heads=1):
    sg = synth.make(graph_name, device="cuda", scale=scale)
    g = ctx.graph(sg.rowptr, sg.colidx)
    if selfloop:
        g2 = g.add_selfloop()
        g.close()
        g = g2
    nv, ne = g.nv, g.ne
    lg = L.LGraph.adopt(g)
    layer = L.Layer(kind, 1, nv, din, dout, lg, act=True)
    if heads > 1:
        layer.set_heads(heads)
    layer.write(L.FEAT_IN, torch.randn(nv, din, device="cuda"))
    layer.write(L.GRAD_IN, torch.randn(nv, dout, device="cuda"))
    out = torch.empty(nv, dout, device="cuda")
    gout = torch.empty(nv, din, device="cuda")

    def step():
        layer.forward(out)
        layer.backward(out, gout)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctx.prof_enable(False)
    br = {}
    for k in KEYS:
        n, ms = ctx.prof_get(k)
        if n:
            br[k] = dict(launches_per_step=n / steps, ms_per_step=ms / steps)
    ctx.prof_reset()
    n_spmm = 2
    print(json.dumps(dict(layer=name, graph=graph_name, nv=nv, ne=ne, din=din, dout=dout, ms_per_step=el / steps * 1e3,
                          gedges_s=n_spmm * ne * steps / el / 1e9, breakdown=br)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default="", help="substring of the layer names to run (e.g. 'GAT')")
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (gaib_set_option), repeatable")
    args = ap.parse_args()
    ctx = L.init(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    if args.only:
        _run = run_layer

        def run_layer_filtered(c, kind, name, *a, **k):
            if args.only.replace("_", " ") in name:
                _run(c, kind, name, *a, **k)
        globals()["run_layer"] = run_layer_filtered
    run_layer(ctx, L.SAGE, "SAGE 128->128", "ogbn-products", 128, 128, False, args.steps, args.scale)
    run_layer(ctx, L.SAGE, "SAGE 256->256 (hidden 256 of scripts/run-sage-products.sh)", "ogbn-products", 256, 256, False, args.steps, args.scale)
    run_layer(ctx, L.GCN, "GCN 128->128", "ogbn-products", 128, 128, True, args.steps, args.scale)
    run_layer(ctx, L.GCN, "GCN 100->128 (layer 0 shape, level 1)", "ogbn-products", 100, 128, True, args.steps, args.scale)
    run_layer(ctx, L.GCN, "GCN 128->47", "ogbn-products", 128, 47, True, args.steps, args.scale)
    run_layer(ctx, L.SAGE, "SAGE 128->47", "ogbn-products", 128, 47, False, args.steps, args.scale)
    run_layer(ctx, L.SAGE, "SAGE 100->128 (layer 0 shape, level 1)", "ogbn-products", 100, 128, False, args.steps, args.scale)
    run_layer(ctx, L.GCN, "GCN 64->64 reddit-shaped", "reddit", 64, 64, True, args.steps, args.scale)
    run_layer(ctx, L.SAGE, "SAGE 64->64 reddit-shaped", "reddit", 64, 64, False, args.steps, args.scale)
    run_layer(ctx, L.GAT, "GAT 64->64", "reddit", 64, 64, True, args.steps, args.scale)
    run_layer(ctx, L.GAT, "GAT 602->64 ", "reddit", 602, 64, True, args.steps, args.scale)
    run_layer(ctx, L.GAT, "GAT 64->64 8 heads", "reddit", 64, 64, True, args.steps, args.scale, heads=8)


if __name__ == "__main__":
    main()
