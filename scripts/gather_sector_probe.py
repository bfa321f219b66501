"""Does a gather of a row that fills only part of its last 128-B line move the whole line?  gaib_spmm (GCN weights, light rows
only: heavy threshold lifted) on the products-shaped graph at D = 16 / 32 / 48 / 64 (64 / 128 / 192 / 256-byte rows, all starting on
64-byte boundaries), one launch per width after a warm-up: run under `rocprofv3 --pmc FETCH_SIZE` and read the counter per launch
(x 2: gfx950 half count) against E x row bytes.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/gather_sector_probe.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
sg = synth.make("ogbn-products", seed=42, device="cuda")
g0 = ctx.graph(sg.rowptr, sg.colidx)
g = g0.add_selfloop()
g0.close()
ctx.set_option("spmm_pad", 0)
for d in (16, 32, 48, 64, 96, 128):
    x = torch.randn(g.nv, d, device="cuda")
    y = torch.empty(g.nv, d, device="cuda")
    for _ in range(2):
        ctx.spmm(g, capi.W_GCN, x, y)
    ctx.sync()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        ctx.spmm(g, capi.W_GCN, x, y)
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({"D": d, "row_bytes": 4 * d, "ms": a.elapsed_time(b) / 4, "edges": g.ne, "gather_gb": g.ne * 4 * d / 1e9,
                      "lines_gb": g.ne * 128 * ((4 * d + 127) // 128) / 1e9, "launches": 6}), flush=True)
    del x, y
