"""Randomised sweep of the row movers and of the relabelling: gaib_gather_rows / gaib_gather_scatter_rows (the halo pack) over
widths 1 .. 300, empty and repeated index lists, and gaib_graph_reorder (degree / BFS / Cuthill-McKee order) on random graphs -- one vertex,
isolated vertices, several components, hubs -- checked for: a permutation, its inverse, rows that keep their edge order,
and aggregations that are BIT-identical once un-permuted.
    python scripts/fuzz_rows_and_order.py [--seconds 90] [--seed 0]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = capi.Context(0)
    t0, n_cases, fails = time.time(), 0, []
    while time.time() - t0 < args.seconds:
        what = ["gather", "scatter", "reorder"][int(rng.integers(3))]
        cfg = dict(what=what)
        try:
            if what in ("gather", "scatter"):
                n_src = int(rng.choice([1, 7, 1000, 50_000]))
                n_idx = int(rng.choice([0, 1, 63, 64, 65, 4096, 100_001]))
                D = int(rng.choice([1, 2, 3, 16, 47, 64, 100, 128, 129, 256, 300]))
                cfg.update(n_src=n_src, n_idx=n_idx, D=D)
                x = torch.randn(n_src, D, device="cuda")
                idx = torch.from_numpy(rng.integers(0, n_src, n_idx)).cuda()
                if what == "gather":
                    out = torch.full((n_idx, D), float("nan"), device="cuda")
                    ctx.gather_rows(idx, x, out)
                    ctx.sync()
                    ok = torch.equal(out, x[idx])
                else:
                    n_dst = n_idx + int(rng.integers(0, 50))
                    dst = torch.from_numpy(rng.permutation(n_dst)[:n_idx].astype(np.int64)).cuda()
                    order = torch.argsort(idx, stable=True)  # the pack runs in SOURCE order (comm.hip)
                    out = torch.full((n_dst, D), 7.0, device="cuda")
                    ctx.gather_scatter_rows(idx[order].contiguous(), dst[order].contiguous(), x, out)
                    ctx.sync()
                    want = torch.full((n_dst, D), 7.0, device="cuda")
                    want[dst] = x[idx]
                    ok = torch.equal(out, want)
                if not ok:
                    raise AssertionError("rows differ")
            else:
                n = int(rng.choice([1, 2, 9, 300, 5000, 40000]))
                avg = float(rng.choice([0.0, 0.6, 3, 20]))
                hub = int(rng.choice([0, 2500])) if n >= 5000 else 0
                method = int(rng.choice([capi.ORDER_DEGREE, capi.ORDER_BFS, capi.ORDER_CM]))
                gseed = int(rng.integers(1 << 30))
                cfg.update(n=n, avg=avg, hub=hub, method=method, gseed=gseed)
                rp, ci = random_graph(n, avg, seed=gseed, power_law=bool(gseed & 1), hub_deg=hub)
                g = ctx.graph(rp, ci.view(np.int32))
                if rng.integers(2):
                    g = g.add_selfloop()
                g.compute_vertex_data()
                r, new_of_old, old_of_new = g.reorder(method)
                r.compute_vertex_data()
                no, on = new_of_old.cpu().numpy(), old_of_new.cpu().numpy()
                if not (np.array_equal(np.sort(no), np.arange(n)) and np.array_equal(no[on], np.arange(n))):
                    raise AssertionError("not a permutation and its inverse")
                rp_o, ci_o = g.rowptr().cpu().numpy(), g.colidx().cpu().numpy().view(np.uint32)
                rp_n, ci_n = r.rowptr().cpu().numpy(), r.colidx().cpu().numpy().view(np.uint32)
                if not np.array_equal(np.diff(rp_n), np.diff(rp_o)[on]):
                    raise AssertionError("row lengths")
                for v in rng.integers(0, n, 8):
                    k = no[v]
                    if not np.array_equal(ci_n[rp_n[k]:rp_n[k + 1]], no[ci_o[rp_o[v]:rp_o[v + 1]]]):
                        raise AssertionError(f"row {v} does not keep its edge order")
                D = int(rng.choice([16, 64, 128]))
                x = torch.randn(n, D, device="cuda")
                x_new = x[old_of_new].contiguous()
                for kind in (capi.W_GCN, capi.W_MEAN, capi.W_MEAN_T):
                    y, y_new = torch.empty_like(x), torch.empty_like(x)
                    ctx.spmm(g, kind, x, y)
                    ctx.spmm(r, kind, x_new, y_new)
                    ctx.sync()
                    if not torch.equal(y_new[new_of_old], y):
                        raise AssertionError(f"kind {kind}: outputs not bit-identical once un-permuted")
                r.close()
                g.close()
        except Exception as e:  # noqa: BLE001
            fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:300]))
            print("FAIL", json.dumps(fails[-1]), flush=True)
        n_cases += 1
    print(json.dumps({"cases": n_cases, "failures": len(fails), "seconds": round(time.time() - t0, 1)}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
