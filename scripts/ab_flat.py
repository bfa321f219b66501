"""A/B of the fused aggregation kernel's two forms (spmm_flat 0 = row by row, 1 = one edge stream per strip of rows) on
uniform random graphs of a given average degree, plain and in accumulate mode (development aid).
    python scripts/ab_flat.py [n_rows] [degrees...]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_449_029
    degs = [float(v) for v in sys.argv[2:]] or [3, 5, 12, 30]
    ctx = capi.Context(0)
    D = 128
    x = torch.randn(n, D, device="cuda"); W = torch.randn(D, D, device="cuda") * 0.1
    agg = torch.zeros(n, D, device="cuda"); y = torch.empty(n, D, device="cuda")
    for deg in degs:
        ne = int(n * deg)
        rows = torch.randint(0, n, (ne,), device="cuda").sort().values
        cols = torch.randint(0, n, (ne,), device="cuda", dtype=torch.int32)
        rp = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); rp[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        g = ctx.graph(rp, cols); ew = torch.rand(ne, device="cuda")
        for acc in (False, True):
            res = {}
            for flat in (0, 1):
                ctx.set_option("spmm_flat", flat)
                for it in range(8):
                    if it == 3:
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                    ctx.spmm_gemm(g, capi.W_EDGE, x, agg, W, y, relu=True, edge_w=ew, accumulate=acc)
                torch.cuda.synchronize(); res[flat] = (time.perf_counter() - t0) / 5 * 1e3
                if acc: agg.zero_()
            gb = (ne * (4 * D + 8) + n * 4 * D * (3 if acc else 2)) / 1e9
            print(f"deg {deg:5.1f} accumulate={acc!s:5} row-by-row {res[0]:.3f} ms  edge-stream {res[1]:.3f} ms  ({gb:.1f} GB -> {gb / 7.7:.2f} ms at 7.7 TB/s)", flush=True)
        g.close()
    ctx.set_option("spmm_flat", -1)
main()
