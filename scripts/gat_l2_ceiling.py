"""GAT backward one-sweep kernel: what would perfect cache locality buy?  

The reddit-shaped 8-head x 64 backward sweep takes 6.1 ms with an L2 hit rate of 0.42 (profiles/r04/gat_tcc_hit_miss.json);
two layouts that aim at the hit rate were measured in round 4 and a column-blocked chunk list in round 2, all slower.  This
script measures the CEILING of that whole line of attack instead of a fourth layout: the same kernel, the same row lengths and
chunk structure, but with every column id divided by 2^k, so that the gathered rows of the whole launch come from a window of
nv / 2^k vertices -- at k = 5 a 7 K-vertex window, 4 MB of [h | grad | records], resident in every XCD's L2.  What remains is
the kernel's instruction issue (~2 000 VALU / LDS instructions per 64-edge chunk) and the linear streams.  If that
time is not well below 6.1 ms, no chunk order, table layout or XCD mapping can reach the 5.5 ms asked for.

    python scripts/gat_l2_ceiling.py [k ...]    # one JSON line per k (default 0 2 3 4 5 6 8)
    bash scripts/profile_gat_ceiling.sh         # TCC_HIT / TCC_MISS of the backward sweep at k = 0 and k = 5
"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, layers as L, synth  # noqa: E402


def main():
    ctx = L.init(0)
    D, H = 64, 8
    sg = synth.make("reddit", seed=7, device="cuda")
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g1 = g0.add_selfloop()
    g0.close()
    rp, ci = g1.rowptr(), g1.colidx()
    nv, ne = g1.nv, g1.ne
    g1.close()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    h = torch.randn(nv, D, device="cuda", generator=gen)
    grad = torch.randn(nv, D, device="cuda", generator=gen)
    fwd = torch.randn(nv, D, device="cuda", generator=gen)
    al = torch.randn(D, device="cuda", generator=gen) * 0.2
    ar = torch.randn(D, device="cuda", generator=gen) * 0.2
    ctx.set_option("gat_fused_fwd", 1)
    ctx.set_option("gat_fused_bwd", 1)
    import os
    for kv in filter(None, os.environ.get("GAIB_OPTS", "").split(",")):  # e.g. GAIB_OPTS=gat_bwd_pk=0: round 5's backward kernel
        k_, v_ = kv.split("=")
        ctx.set_option(k_.strip(), int(v_))
    shifts = [int(v) for v in sys.argv[1:]] or [0, 2, 3, 4, 5, 6, 8]  # (a counter pass names the two shifts it wants: 0 5)
    for k in shifts:
        cols = (ci.to(torch.int64) >> k).to(torch.int32)  # rows stay sorted; duplicates are fine for a timing
        g = ctx.graph(rp, cols)
        out = torch.empty(nv, D, device="cuda")
        stats = torch.empty(nv, H, 2, device="cuda")
        go = torch.empty(nv, D, device="cuda")
        lg, rg = torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
        assert ctx.gat_forward_fused(g, h, al, ar, out, stats, heads=H)

        def bwd():
            assert ctx.gat_backward_fused(g, h, grad, fwd, al, ar, None, go, lg, rg, heads=H, row_stats=stats)

        def fw():
            assert ctx.gat_forward_fused(g, h, al, ar, out, stats, heads=H)

        rec = {"shift": k, "window_vertices": (nv >> k) + 1, "window_mb_h_grad_rec": ((nv >> k) + 1) * (2 * D * 4 + H * 16) / 1e6,
               "ne": ne}
        for name, fn, key in (("backward", bwd, "gat_bwd_fused"), ("forward", fw, "gat_fwd_fused")):
            for _ in range(3):
                fn()
            ctx.sync()
            ctx.prof_reset()
            ctx.prof_enable(True)
            for _ in range(8):
                fn()
            ctx.prof_enable(False)
            n, ms = ctx.prof_get(key)
            ctx.prof_reset()
            rec[name + "_ms"] = ms / max(n, 1)
        print(json.dumps(rec), flush=True)
        g.close()
        del out, stats, go
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
