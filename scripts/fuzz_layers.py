"""Randomised check of the GCN / SAGE layer classes on one GPU against fp64 evaluations of gcn_layer.cpp / sage_layer.cpp on the
device: widths 16 .. 256 in every combination (aggregate-first and product-first branches, the 47- and 100-wide padded tables,
the two-K-slab path at 256), level 0 (no input gradient) and level 1, with and without relu, random graphs from 3 vertices to
60 000 with hubs and isolated vertices, options that move the dispatch (fusion off, heavy threshold, tile supply).
    python scripts/fuzz_layers.py [--seconds 120] [--seed 0]
The relu mask of backward is the GPU's own forward output (as in the layers); forward is compared before the mask matters.
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
from graphaibench_amd import layers as L  # noqa: E402
from util import random_graph  # noqa: E402

OPTS = {"spmm_fuse": [0, 1], "spmm_heavy_threshold": [64, 1024], "spmm_tile_xcd": [-1, 0, 1024], "spmm_flat": [-1, 0, 1], "spmm_pad": [0, 1]}
DEFAULTS = {"spmm_fuse": 1, "spmm_heavy_threshold": 1024, "spmm_tile_xcd": -1, "spmm_flat": -1, "spmm_pad": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = L.init(0)
    t0, n_cases, fails, worst = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        n = int(rng.choice([3, 50, 700, 6000, 60000]))
        avg = float(rng.choice([0.5, 4, 14, 40]))
        hub = int(rng.choice([0, 0, 3000])) if n >= 6000 else 0
        arch = str(rng.choice(["gcn", "sage"]))
        din, d = int(rng.choice([16, 47, 64, 100, 128, 200, 256])), int(rng.choice([16, 47, 64, 128, 256]))
        level, act = int(rng.integers(2)), bool(rng.integers(2))
        opts = {k: int(rng.choice(v)) for k, v in OPTS.items() if rng.integers(3) == 0}
        gseed = int(rng.integers(1 << 30))
        cfg = dict(n=n, avg=avg, hub=hub, arch=arch, din=din, d=d, level=level, act=act, opts=opts, gseed=gseed)
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            rp, ci = random_graph(n, avg, seed=gseed, power_law=bool(gseed & 1), hub_deg=hub)
            g = L.LGraph.from_host(rp, ci, add_selfloop=arch == "gcn")  # SAGE aggregates over A (net.cpp:96)
            dg = g.device_graph()
            dg.ctx = ctx
            rowptr, col = dg.rowptr().long(), dg.colidx().long()
            rows = torch.repeat_interleave(torch.arange(n, device="cuda"), rowptr[1:] - rowptr[:-1])
            deg = (rowptr[1:] - rowptr[:-1]).double()
            layer = L.Layer(L.GCN if arch == "gcn" else L.SAGE, level, n, din, d, g, act)
            gen = torch.Generator(device="cuda")
            gen.manual_seed(gseed)
            x = torch.randn(n, din, device="cuda", generator=gen)
            gin = torch.randn(n, d, device="cuda", generator=gen)
            W = layer.tensor(L.W_NEIGH, (din, d)).double()
            if level == 0:
                layer.set_feat_in(x)
            else:
                layer.write(L.FEAT_IN, x)
            out = torch.empty(n, d, device="cuda")
            layer.forward(out)
            L.sync()
            out_fwd = out.clone()
            layer.write(L.GRAD_IN, gin)
            go = torch.empty(n, din, device="cuda") if level > 0 else None
            layer.backward(out, go)
            L.sync()
            X, G = x.double(), gin.double()

            def A(w, src, dst, M):
                return torch.zeros(n, M.shape[1], dtype=torch.float64, device="cuda").index_add_(0, dst, w[:, None] * M[src])

            if arch == "gcn":
                vd = torch.where(deg > 0, deg.sqrt().reciprocal(), torch.zeros_like(deg))
                w = vd[rows] * vd[col]
                ax = A(w, col, rows, X)
                y = ax @ W
            else:
                Ws = layer.tensor(L.W_SELF, (din, d)).double()
                inv = torch.where(deg > 0, deg.reciprocal(), torch.zeros_like(deg))
                ax = A(inv[rows], col, rows, X)
                y = ax @ W + X @ Ws
            want = {"out": y.clamp(min=0) if act else y}
            gm = G * (out_fwd > 0) if act else G  # the layer's own mask (d_relu on its forward output)
            want["W_grad"] = ax.t() @ gm
            got = {"out": out_fwd, "W_grad": layer.tensor(L.W_NEIGH_GRAD, (din, d))}
            if arch == "sage":
                want["W_self_grad"] = X.t() @ gm
                got["W_self_grad"] = layer.tensor(L.W_SELF_GRAD, (din, d))
            if level > 0:
                got["grad_out"] = go
                want["grad_out"] = (A(w, col, rows, gm) @ W.t()) if arch == "gcn" else (A(inv[col], col, rows, gm @ W.t()) + gm @ Ws.t())
            for name, ref in want.items():
                gv = got[name]
                if not torch.isfinite(gv).all():
                    raise AssertionError(f"{name}: non-finite")
                scale = max(float(ref.abs().max()), 1e-30)
                e = float((gv.double() - ref).abs().max()) / scale if gv.numel() else 0.0
                worst[name] = max(worst.get(name, 0.0), e)
                if e > 1e-4:
                    raise AssertionError(f"{name}: {e:.3e} of the tensor's scale from fp64")
            layer.close()
            g.close()
        except Exception as e:  # noqa: BLE001
            fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:300]))
            print("FAIL", json.dumps(fails[-1]), flush=True)
        finally:
            for k in opts:
                ctx.set_option(k, DEFAULTS[k])
        n_cases += 1
        if n_cases % 50 == 0:
            print(f"{n_cases} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "worst_rel_err": {k: float(f"{v:.2e}") for k, v in sorted(worst.items())},
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
