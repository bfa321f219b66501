"""(scripts/fuzz_spmm.py sweeps gaib_spmm alone, incl. per-edge weights and heads, on small graphs.)
Randomised sweep of the aggregation entry points against an fp64 evaluation on the device: gaib_spmm and gaib_spmm_gemm /
gaib_spmm_gemm2 over random graph shapes (empty rows, one vertex, hubs, dense), feature widths (odd ones too), operator
kinds, flags (accumulate, relu, transW, scratch aggregate, self term) and every dispatch option that changes which kernel
runs (tile supply, edge-stream form, padding, heavy threshold, unroll, fusion on / off).
    python scripts/fuzz_aggregation.py [--seconds 120] [--seed 0]
Prints one line per failure with the configuration that reproduces it, and a summary; exit 1 if anything failed.
Test infrastructure (a development tool: what it finds becomes a case in tests/)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from graphaibench_amd import capi  # noqa: E402
from util import random_graph  # noqa: E402

OPTS = {
    "spmm_tile_xcd": [-1, 0, 16, 1024],
    "spmm_flat": [-1, 0, 1],
    "spmm_pad": [0, 1],
    "spmm_heavy_threshold": [1, 64, 256, 1024, 1 << 30],
    "spmm_unroll": [0, 4, 8, 16],
    "spmm_fuse": [0, 1],
    "spmm_xcd_swizzle": [0, 1, 2],
}
DEFAULTS = {"spmm_tile_xcd": -1, "spmm_flat": -1, "spmm_pad": 1, "spmm_heavy_threshold": 1024, "spmm_unroll": 0, "spmm_fuse": 1,
            "spmm_xcd_swizzle": 2}  # gaib_ctx_create (runtime.hip)


def weights(kind, rowptr, col):
    """per-edge fp64 weights of the operator kinds (gcn_aggregator.cpp:61-66, sage_aggregator.cpp:18,44), formed from the
    SAME fp32 per-vertex factors the library uses"""
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
    rows = torch.repeat_interleave(torch.arange(deg.numel(), device=col.device), rowptr[1:] - rowptr[:-1])
    if kind == capi.W_GCN:
        t = torch.sqrt(deg)
        vd = torch.where(t == 0, torch.zeros_like(t), (1.0 / t.double()).float())
        return rows, (vd[rows] * vd[col]).double()  # the fp32 product the path forms
    inv = (1.0 / deg.double()).float()
    inv = torch.where(deg == 0, torch.zeros_like(inv), inv)
    return rows, (inv[rows] if kind == capi.W_MEAN else inv[col]).double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = capi.Context(0)
    t0, n_cases, fails = time.time(), 0, []
    while time.time() - t0 < args.seconds:
        n = int(rng.choice([1, 2, 17, 300, 4097, 20000, 70000]))
        avg = float(rng.choice([0.0, 0.5, 3, 16, 60]))
        hub = int(rng.choice([0, 0, 200, 3000])) if n > 3500 else 0
        rp, ci = random_graph(n, avg, seed=int(rng.integers(1 << 30)), power_law=bool(rng.integers(2)), hub_deg=hub)
        selfloop = bool(rng.integers(2))
        g = ctx.graph(rp, ci.view(np.int32))
        if selfloop:
            g = g.add_selfloop()
        g.compute_vertex_data()
        rowptr, col = g.rowptr(), g.colidx().long()
        nv = g.nv
        kind = int(rng.choice([capi.W_GCN, capi.W_MEAN, capi.W_MEAN_T]))
        opts = {k: int(rng.choice(v)) for k, v in OPTS.items() if rng.integers(3) == 0}
        for k, v in opts.items():
            ctx.set_option(k, v)
        cfg = dict(n=n, avg=avg, hub=hub, selfloop=selfloop, kind=kind, opts=opts)
        try:
            rows, w = weights(kind, rowptr, col)
            fused = bool(rng.integers(2))
            if not fused:
                D = int(rng.choice([1, 3, 16, 47, 64, 100, 128, 130, 256, 260]))
                acc, relu = bool(rng.integers(2)), bool(rng.integers(2))
                cfg.update(call="spmm", D=D, accumulate=acc, relu=relu)
                x = torch.randn(nv, D, device="cuda")
                out0 = torch.randn(nv, D, device="cuda")
                out = out0.clone()
                ctx.spmm(g, kind, x, out, accumulate=acc, relu=relu)
                ctx.sync()
                want = torch.zeros(nv, D, dtype=torch.float64, device="cuda").index_add_(0, rows, w[:, None] * x.double()[col])
                if acc:
                    want += out0.double()
                if relu:
                    want.clamp_(min=0)
                outs = [("out", out, want)]
            else:
                din = int(rng.choice([16, 47, 64, 100, 128, 256]))
                dout = int(rng.choice([16, 47, 64, 128, 256]))
                transW, relu, scratch = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
                two = bool(rng.integers(2)) and kind != capi.W_GCN
                cfg.update(call="spmm_gemm", din=din, dout=dout, transW=transW, relu=relu, scratch=scratch, self_term=two)
                x = torch.randn(nv, din, device="cuda")
                W = torch.randn((dout, din) if transW else (din, dout), device="cuda") * 0.2
                agg = torch.full((nv, din), float("nan"), device="cuda")
                out = torch.full((nv, dout), float("nan"), device="cuda")
                rows2 = torch.randn(nv, din, device="cuda") if two else None
                W2 = torch.randn_like(W) * 0.2 if two else None
                ctx.spmm_gemm(g, kind, x, agg, W, out, transW=transW, relu=relu, agg_scratch=scratch, rows2=rows2, W2=W2)
                ctx.sync()
                wagg = torch.zeros(nv, din, dtype=torch.float64, device="cuda").index_add_(0, rows, w[:, None] * x.double()[col])
                op = (lambda M: M.double().t()) if transW else (lambda M: M.double())
                want = wagg.float().double() @ op(W)  # the product reads the fp32 aggregate
                if two:
                    want += rows2.double() @ op(W2)
                if relu:
                    want.clamp_(min=0)
                outs = [("out", out, want)]
                if not scratch:
                    outs.append(("agg", agg, wagg))
            for name, got, want in outs:
                scale = max(float(want.abs().max()), 1e-30)
                if not torch.isfinite(got).all():
                    raise AssertionError(f"{name}: non-finite values")
                err = float(((got.double() - want).abs() / (want.abs() + 0.1 * scale)).max()) if got.numel() else 0.0
                if err > 1e-4:
                    raise AssertionError(f"{name}: element-wise error {err:.3e}")
        except Exception as e:  # noqa: BLE001
            fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:300]))
            print("FAIL", json.dumps(fails[-1]), flush=True)
        finally:
            for k in opts:  # back to the defaults
                ctx.set_option(k, DEFAULTS[k])
            g.close()
        n_cases += 1
        if n_cases % 50 == 0:
            print(f"{n_cases} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "seconds": round(time.time() - t0, 1), "seed": args.seed}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
