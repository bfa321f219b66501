"""Does the kernel before a GEMM change which GEMM kernel wins?  NN product 2.45 M x 256 x 256 timed with HIP events,
back to back (warm) and with another kernel in front of every launch: a 20 GB streaming copy (cold caches, few pages
touched at a time) or a random row gather over 20 GB (cold caches AND a swept TLB).  sgemm_variant 40 = LDS-tiled kernel,
41 = persistent streaming kernel.      python scripts/ab_cold_gemm.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi

def main():
    ctx = capi.Context(0)
    nv, d = 2449029, 256
    x = torch.randn(nv, d, device="cuda"); W = torch.randn(d, d, device="cuda") * 0.1; y = torch.empty(nv, d, device="cuda")
    big = torch.empty(5 * 1024 * 1024 * 1024 // 4 * 4, device="cuda", dtype=torch.float32).view(-1, 128)  # 20 GB, 512-B rows
    big2 = torch.empty_like(big[: big.shape[0] // 8])
    idx = torch.randint(0, big.shape[0], (big2.shape[0],), device="cuda")
    fronts = {"back to back": lambda: None,
              "after a 2.5 GB streaming copy": lambda: big2.copy_(big[: big2.shape[0]]),
              "after a random 512-B row gather over 20 GB": lambda: torch.index_select(big, 0, idx, out=big2)}
    for name, front in fronts.items():
        for variant in (40, 41, 40, 41):
            ctx.set_option("sgemm_variant", variant)
            ts = []
            for it in range(8):
                front()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ctx.sgemm(x, W, y); b.record(); torch.cuda.synchronize()
                if it >= 2: ts.append(a.elapsed_time(b))
            print(f"{name:45s} variant {variant}: {sum(ts) / len(ts):.3f} ms", flush=True)
    ctx.set_option("sgemm_variant", 0)
main()
