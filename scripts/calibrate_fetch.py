"""FETCH_SIZE calibration for the SpMM access pattern (MI355X_MICROARCH.md, HBM section: widths
other than 16 B/lane are uncalibrated).  A permutation graph (every row has exactly one neighbour,
a random permutation) makes gaib_spmm gather each 512-B feature row of a 1.25 GB table exactly
once: known HBM read bytes = nv*(4*D) + nv*(8 rowptr + 4 col + 4 weight).  Run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 scripts/calibrate_fetch.py
and compare the counter of spmm_w64_kernel with the printed byte count."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402

nv, D = 2_449_029, 128
ctx = capi.Context(0)
g = torch.Generator(device="cuda")
g.manual_seed(1)
perm = torch.randperm(nv, generator=g, device="cuda").to(torch.int32)
rowptr = torch.arange(nv + 1, dtype=torch.int64, device="cuda")
gr = ctx.graph(rowptr, perm)
x = torch.randn(nv, D, device="cuda")
out = torch.empty_like(x)
ew = torch.ones(nv, device="cuda")
for _ in range(3):
    ctx.spmm(gr, capi.W_EDGE, x, out, edge_w=ew)
torch.cuda.synchronize()
print(json.dumps(dict(kernel="spmm_w64_kernel", known_read_bytes=nv * 4 * D + nv * 16, known_write_bytes=nv * 4 * D)))
