"""SGEMM-only driver for counter collection (rocprofv3 --pmc ...): the three layer GEMMs at the
products / SAGE shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402

ctx = capi.Context(0)
nv = 2_449_029
for d in (128, 256):
    x = torch.randn(nv, d, device="cuda")
    w = torch.randn(d, d, device="cuda")
    y = torch.empty(nv, d, device="cuda")
    dw = torch.empty(d, d, device="cuda")
    for _ in range(3):
        ctx.sgemm(x, w, y)
        ctx.sgemm(x, w, y, False, True)
        ctx.sgemm(x, y, dw, True, False)
    torch.cuda.synchronize()
    del x, w, y, dw
print("done")
