"""Randomised self-consistency sweep of the fused aggregation kernels (development aid, run on the GPU box):
gaib_spmm_gemm / gaib_spmm_gemm2 against gaib_spmm + torch fp64 on many random shapes, graphs with empty rows,
all-heavy rows, row counts around the 16-row tile and 64-row workgroup boundaries, accumulate mode.

    python scripts/fuzz_fused.py [n_cases] [seed]
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from graphaibench_amd import capi  # noqa: E402
from util import random_graph  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = capi.Context(0)
    worst = 0.0
    for case in range(n_cases):
        nv = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 1000, 3001, int(rng.integers(1, 5000)), int(rng.integers(5000, 150000))]))
        deg = float(rng.choice([0.5, 3, 12, 40]))
        hub = int(rng.choice([0, 0, min(nv - 1, 1500), min(nv - 1, 3000)])) if nv > 1200 else 0
        rp, ci = random_graph(max(nv, 2), deg, seed=int(rng.integers(1 << 30)), power_law=bool(rng.integers(2)), hub_deg=hub)
        nv = len(rp) - 1
        thr = int(rng.choice([1024, 1024, 8, 1]))  # tiny thresholds make (almost) every row heavy
        ctx.set_option("spmm_heavy_threshold", thr)
        flat = int(rng.choice([-1, 0, 1]))  # fused kernel: row by row / one edge stream per strip / by average degree
        ctx.set_option("spmm_flat", flat)
        # tile supply: by the graph / one counter / XCD-affine chunks of 16 or 64 tiles (the affine form with and without the
        # request of the next row's column ids a row ahead)
        ctx.set_option("spmm_tile_xcd", int(rng.choice([-1, 0, 1, 64])))
        ctx.set_option("spmm_prefetch_ids", int(rng.integers(2)))
        len_in = int(rng.choice([1, 7, 16, 33, 47, 64, 66, 100, 128]))
        len_out = int(rng.choice([1, 7, 16, 47, 64, 100, 128, 130, 200]))
        kind = int(rng.choice([capi.W_MEAN, capi.W_MEAN_T, capi.W_EDGE, capi.W_GCN]))
        transW = bool(rng.integers(2))
        relu = bool(rng.integers(2))
        dual = bool(rng.integers(2))
        scratch = bool(rng.integers(2))
        g = ctx.graph(rp, ci.view(np.int32))
        if kind == capi.W_GCN:
            g = g.add_selfloop()
        ne = g.ne
        x = torch.randn(nv, len_in, device="cuda")
        ew = torch.rand(max(ne, 1), device="cuda") if kind == capi.W_EDGE else None
        W = torch.randn(*((len_out, len_in) if transW else (len_in, len_out)), device="cuda") * 0.2
        W2 = torch.randn_like(W) * 0.2 if dual else None
        agg_ref = torch.zeros(nv, len_in, device="cuda")
        ctx.spmm(g, kind, x, agg_ref, edge_w=ew)
        opW = (W.T if transW else W).double()
        want = agg_ref.double() @ opW
        if dual:
            want = want + x.double() @ (W2.T if transW else W2).double()
        if relu:
            want = torch.relu(want)
        agg = torch.full((nv, len_in), 7.0, device="cuda")
        y = torch.full((nv, len_out), -3.0, device="cuda")
        ctx.spmm_gemm(g, kind, x, agg, W, y, transW=transW, relu=relu, agg_scratch=scratch, edge_w=ew,
                      rows2=x if dual else None, W2=W2)
        ctx.sync()
        scale = max(want.abs().max().item(), 1e-6)
        err = (y.double() - want).abs().max().item() / scale
        worst = max(worst, err)
        ok = err < 2e-5 and (scratch or torch.equal(agg, agg_ref) or ne == 0)
        if ok and kind != capi.W_GCN and not dual and ne > 0:
            # accumulate mode: the edges split by column (a partition's own / halo halves), second half fused
            rows_of = np.repeat(np.arange(nv), np.diff(rp))
            cut = int(rng.integers(0, nv + 1))
            halves = []
            for mask in (ci < cut, ci >= cut):
                cnt = np.bincount(rows_of[mask], minlength=nv)
                gh = ctx.graph(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), ci[mask].view(np.int32))
                deg_all = np.diff(rp)
                inv = torch.from_numpy(np.where(deg_all > 0, 1.0 / np.maximum(deg_all, 1), 0.0).astype(np.float32)).cuda()
                gh.set_vertex_norm(row_vdata=inv, col_vdata=inv, col_inv_deg=inv, row_inv_deg=inv)
                ewh = ew[torch.from_numpy(np.nonzero(mask)[0]).cuda()] if ew is not None and mask.any() else ew
                halves.append((gh, ewh))
            agg2 = torch.full((nv, len_in), 5.0, device="cuda")
            y2 = torch.full((nv, len_out), -3.0, device="cuda")
            ctx.spmm(halves[0][0], kind, x, agg2, edge_w=halves[0][1])
            ctx.spmm_gemm(halves[1][0], kind, x, agg2, W, y2, transW=transW, relu=relu, edge_w=halves[1][1], accumulate=True)
            ctx.sync()
            err2 = (y2.double() - want).abs().max().item() / scale
            erra = (agg2.double() - agg_ref.double()).abs().max().item() / max(agg_ref.abs().max().item(), 1e-6)
            worst = max(worst, err2)
            if not (err2 < 2e-5 and erra < 1e-5):
                print(f"FAIL (accumulate) case {case}: nv={nv} ne={ne} thr={thr} flat={flat} len_in={len_in} len_out={len_out} "
                      f"kind={kind} cut={cut} err={err2:.2e} agg_err={erra:.2e}")
                sys.exit(1)
            for gh, _ in halves:
                gh.close()
        if not ok:
            print(f"FAIL case {case}: flat={flat} nv={nv} ne={ne} thr={thr} len_in={len_in} len_out={len_out} kind={kind} transW={transW} "
                  f"relu={relu} dual={dual} scratch={scratch} err={err:.2e} agg_equal={torch.equal(agg, agg_ref)}")
            sys.exit(1)
        g.close()
    ctx.set_option("spmm_heavy_threshold", 1024)
    ctx.set_option("spmm_flat", -1)
    print(f"{n_cases} cases ok, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
