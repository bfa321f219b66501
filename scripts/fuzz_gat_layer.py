"""(scripts/fuzz_gat.py sweeps the staged GAT kernels one by one.)
Randomised cross-check of the GAT layer's two implementations behind the same C++ class: the one-sweep kernels (default where
the shape allows) against the staged pieces (options gat_fused_fwd = gat_fused_bwd = 0) and both against an fp64 evaluation
of GAT_Aggregator's formulas (gat_aggregator.cpp:57-200) on the device -- random graphs (isolated vertices, hubs, dense),
input / output widths and head counts, with and without activation.
    python scripts/fuzz_gat_layer.py [--seconds 120] [--seed 0]
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


def fp64_layer(rowptr, col, x, W, al, ar, gin, heads, act):
    """forward output, input gradient, W / alpha gradients of GAT_layer (gat_layer.cpp) in fp64; no gradient through the
    scores into h (the reference's choice)"""
    n, d = x.shape[0], W.shape[1]
    dh = d // heads
    rows = torch.repeat_interleave(torch.arange(n, device=x.device), rowptr[1:] - rowptr[:-1])
    h = x.double() @ W.double()
    out = torch.zeros(n, d, dtype=torch.float64, device=x.device)
    ps, ts = [], []
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        hk = h[:, sl]
        t = (hk @ al[sl].double())[rows] + (hk @ ar[sl].double())[col]
        s = torch.where(t > 0, t, 0.2 * t)
        M = torch.full((n,), -float("inf"), dtype=torch.float64, device=x.device).scatter_reduce(0, rows, s, "amax")
        e = torch.exp(s - M[rows])
        S = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, rows, e)
        p = e / S[rows]
        out[:, sl] = torch.zeros(n, dh, dtype=torch.float64, device=x.device).index_add_(0, rows, p[:, None] * hk[col])
        ps.append(p)
        ts.append(t)
    y = out.clamp(min=0) if act else out
    g = gin.double() * (y > 0) if act else gin.double()
    gh = torch.zeros(n, d, dtype=torch.float64, device=x.device)
    lg, rg = torch.zeros(d, dtype=torch.float64, device=x.device), torch.zeros(d, dtype=torch.float64, device=x.device)
    mag = 0.0  # size of the SUMMANDS of the alpha gradients: the sums themselves cancel (exactly, where a row has one edge)
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        hk, gk, p, t = h[:, sl], g[:, sl], ps[k], ts[k]
        dp = (gk[rows] * hk[col]).sum(1)
        mag = max(mag, float(dp.abs().max() * hk.abs().max()) * float(dp.numel()) ** 0.5)  # one summand x sqrt(edges)
        rowdot = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, rows, p * dp)
        ge = p * (dp - rowdot[rows]) * torch.where(t > 0, 1.0, 0.2)
        lg[sl] = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, rows, ge) @ hk
        rg[sl] = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, col, ge) @ hk
        gh[:, sl] = torch.zeros(n, dh, dtype=torch.float64, device=x.device).index_add_(0, col, p[:, None] * gk[rows])  # P^T g
    return y, gh @ W.double().t(), x.double().t() @ gh, lg, rg, torch.stack(ts, 1), mag


def run_layer(g, n, din, d, heads, act, x, gin, first):
    layer = L.Layer(L.GAT, 1, n, din, d, g, act)
    layer.set_heads(heads)
    W = layer.tensor(L.W_NEIGH, (din, d))
    al, ar = layer.tensor(L.ALPHA_L, (d,)), layer.tensor(L.ALPHA_R, (d,))
    layer.write(L.FEAT_IN, x)
    out = torch.empty(n, d, device="cuda")
    layer.forward(out)
    layer.write(L.GRAD_IN, gin)
    go = torch.empty(n, din, device="cuda")
    layer.backward(out, go)
    L.sync()
    res = dict(out=out.clone(), grad_out=go, W_grad=layer.tensor(L.W_NEIGH_GRAD, (din, d)), alpha_l=layer.tensor(L.ALPHA_LGRAD, (d,)),
               alpha_r=layer.tensor(L.ALPHA_RGRAD, (d,)))
    layer.close()
    return res, W, al, ar


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = L.init(0)
    t0, n_cases, fails, worst = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        n = int(rng.choice([3, 40, 500, 3000, 12000]))
        avg = float(rng.choice([0.5, 3, 12, 40]))
        hub = int(rng.choice([0, 0, 1500])) if n >= 3000 else 0
        rp, ci = random_graph(n, avg, seed=int(rng.integers(1 << 30)), power_law=bool(rng.integers(2)), hub_deg=hub)
        d = int(rng.choice([8, 16, 32, 64, 64, 128]))
        heads = int(rng.choice([h for h in (1, 2, 4, 8, 16) if d % h == 0 and d // h >= 1]))
        din = int(rng.choice([16, 48, 64, 100]))
        act = bool(rng.integers(2))
        cfg = dict(n=n, avg=avg, hub=hub, din=din, d=d, heads=heads, act=act)
        try:
            g = L.LGraph.from_host(rp, ci, add_selfloop=True)  # (GAT attends over A + I, net.cpp:96)
            dg = g.device_graph()
            dg.ctx = ctx  # (the non-owning view carries no context of its own)
            rowptr, col = dg.rowptr().long(), dg.colidx().long()
            x = torch.randn(n, din, device="cuda")
            gin = torch.randn(n, d, device="cuda")
            res = {}
            for mode in ("default", "staged"):
                v = 0 if mode == "staged" else -1
                ctx.set_option("gat_fused_fwd", v)
                ctx.set_option("gat_fused_bwd", v)
                res[mode], W, al, ar = run_layer(g, n, din, d, heads, act, x, gin, mode == "default")
            ctx.set_option("gat_fused_fwd", -1)
            ctx.set_option("gat_fused_bwd", -1)
            y, go, wg, lg, rg, t, mag = fp64_layer(rowptr, col, x, W, al, ar, gin, heads, act)
            # scores within rounding of zero make leaky_relu' (and, with act, outputs within rounding of zero the relu mask)
            # a coin flip between two correct fp32 evaluations: such cases are counted, not compared
            fragile = bool((t.abs() < 1e-5 * t.abs().max()).any()) or (act and bool((y.abs() < 1e-6 * y.abs().max())[y != 0].any()))
            want = dict(out=y, grad_out=go, W_grad=wg, alpha_l=lg, alpha_r=rg)
            for mode in res:
                for name, ref in want.items():
                    got = res[mode][name].double()
                    if not torch.isfinite(got).all():
                        raise AssertionError(f"{mode} {name}: non-finite")
                    scale = max(float(ref.abs().max()), 1e-30)
                    if name.startswith("alpha"):  # (the one-sweep backward's <grad, out> - p dp is rounding where p = 1)
                        scale = max(float(lg.abs().max()), float(rg.abs().max()), 1e-2 * mag)  # tol = 2e-6 sqrt(ne) summands
                    err = float((got - ref).abs().max()) / scale
                    key = f"{mode}.{name}"
                    if not fragile or name == "out":
                        worst[key] = max(worst.get(key, 0.0), err)
                        if err > (2e-4 if name.startswith("alpha") or name == "W_grad" else 1e-4):
                            raise AssertionError(f"{key}: {err:.3e} of the tensor's scale from fp64")
            cfg["fragile"] = fragile
            g.close()
        except Exception as e:  # noqa: BLE001
            fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:300]))
            print("FAIL", json.dumps(fails[-1]), flush=True)
            ctx.set_option("gat_fused_fwd", -1)
            ctx.set_option("gat_fused_bwd", -1)
        n_cases += 1
        if n_cases % 25 == 0:
            print(f"{n_cases} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "worst_rel_err": {k: float(f"{v:.2e}") for k, v in sorted(worst.items())},
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
