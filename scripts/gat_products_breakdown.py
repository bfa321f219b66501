"""The GAT layer step on the products shape (1 head x 64; a sparse graph: one short chunk per row, 627 MB tables), launch by
launch: one sweep vs staged, every launch's time next to the time its algorithmic bytes would take at 8 TB/s (gaib_prof_table).
The ledger line behind "why the one sweep is 0.93 x staged there and not 0.6 x" (DESIGN 3.3).  One JSON line per form."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import layers as L, synth  # noqa: E402


def main():
    ctx = L.init(0)
    sg = synth.make("ogbn-products", device="cuda")
    g0 = ctx.graph(sg.rowptr, sg.colidx)
    g = g0.add_selfloop()
    g0.close()
    del sg
    nv, ne, d = g.nv, g.ne, 64
    lg = L.LGraph.adopt(g)
    forms = [("one_sweep", 1, 4), ("one_sweep_8_in_flight", 1, 8), ("staged", 0, 4)]
    for name, opt, unroll in forms:
        ctx.set_option("gat_fused_fwd", opt)
        ctx.set_option("gat_fused_bwd", opt)
        ctx.set_option("gat_fused_unroll", unroll)
        layer = L.Layer(L.GAT, 1, nv, d, d, lg, act=True)
        layer.write(L.FEAT_IN, torch.randn(nv, d, device="cuda"))
        layer.write(L.GRAD_IN, torch.randn(nv, d, device="cuda"))
        out, gout = torch.empty(nv, d, device="cuda"), torch.empty(nv, d, device="cuda")
        for _ in range(2):
            layer.forward(out)
            layer.backward(out, gout)
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        steps = 4
        for _ in range(steps):
            layer.forward(out)
            layer.backward(out, gout)
        ctx.prof_enable(False)
        tab = ctx.prof_table()
        ctx.prof_reset()
        rec = {"form": name, "nv": nv, "ne": ne, "len": d, "heads": 1,
               "per_step": {k: {"launches": v["count"] / steps, "ms": round(v["ms"] / steps, 3), "roof_ms": round(v["roof_ms"] / steps, 3),
                                "alg_gb": round(v["bytes"] / steps / 1e9, 2)} for k, v in tab.items()},
               "timed_ms_per_step": round(sum(v["ms"] for v in tab.values()) / steps, 3),
               "roof_ms_per_step": round(sum(v["roof_ms"] for v in tab.values()) / steps, 3)}
        print(json.dumps(rec), flush=True)
        layer.close()
        del out, gout
        torch.cuda.empty_cache()
    ctx.set_option("gat_fused_fwd", -1)
    ctx.set_option("gat_fused_bwd", -1)


if __name__ == "__main__":
    main()
