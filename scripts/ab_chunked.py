"""Where does the ordered-chunk aggregation (spmm_chunked) start to pay?  Uniform random graphs, D = 64:
row kernels vs chunk kernels over average degree and table size.   python scripts/ab_chunked.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi

def main():
    ctx = capi.Context(0)
    D = 64
    for n in (250_000, 1_000_000, 2_000_000):
        x = torch.randn(n, D, device="cuda"); out = torch.empty(n, D, device="cuda")
        for deg in (32, 64, 128, 256):
            if n * deg > 300_000_000: continue
            cols = torch.randint(0, n, (n, deg), device="cuda", dtype=torch.int32).sort(1).values.reshape(-1).contiguous()
            rp = torch.arange(n + 1, device="cuda", dtype=torch.int64) * deg
            g = ctx.graph(rp, cols); del cols
            res = {}
            for opt in (0, 1):
                ctx.set_option("spmm_chunked", opt)
                for it in range(6):
                    if it == 2:
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                    ctx.spmm(g, capi.W_MEAN, x, out)
                torch.cuda.synchronize(); res[opt] = (time.perf_counter() - t0) / 4 * 1e3
            print(f"n={n} table={n * D * 4 / 1e6:.0f} MB deg={deg}: rows {res[0]:.3f} ms  chunks {res[1]:.3f} ms", flush=True)
            g.close()
    ctx.set_option("spmm_chunked", -1)
main()
