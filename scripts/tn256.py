"""The K = 2.45 M weight gradients of the hidden width 256 (scripts/run-sage-products.sh), alone: C = A^T . G plain and with
the d_relu mask folded in, sgemm_variant 0 (teams of quadrant waves, round 3) vs 32 (the LDS-tiled kernel at 256, the round-2
path), correctness against fp64 on the device.  Shapes: 256x256, 100x256 (SAGE layer 0), 256x128.
    python scripts/tn256.py [variants ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ctx = capi.Context(0)
    nv = 2449029
    variants = [int(v) for v in sys.argv[1:]] or [0, 32]
    for (m, n) in ((256, 256), (100, 256), (256, 128), (128, 128)):
        torch.manual_seed(m + n)
        a = torch.randn(nv, m, device="cuda")
        g = torch.randn(nv, n, device="cuda")
        mask = torch.randn(nv, n, device="cuda")
        gm = torch.where(mask > 0, g, torch.zeros_like(g))
        ref = torch.zeros(m, n, dtype=torch.float64, device="cuda")
        refm = torch.zeros(m, n, dtype=torch.float64, device="cuda")
        for s in range(0, nv, 1 << 19):
            ref += a[s:s + (1 << 19)].double().T @ g[s:s + (1 << 19)].double()
            refm += a[s:s + (1 << 19)].double().T @ gm[s:s + (1 << 19)].double()
        dW = torch.empty(m, n, device="cuda")
        fl = 2.0 * nv * m * n
        for v in variants:
            ctx.set_option("sgemm_variant", v)
            t = timeit(lambda: ctx.sgemm(a, g, dW, True, False))
            e = ((dW.double() - ref).abs().max() / ref.abs().max()).item()
            g2 = g.clone()
            ctx.sgemm_drelu(a, g2, mask, dW)
            torch.cuda.synchronize()
            em = ((dW.double() - refm).abs().max() / refm.abs().max()).item()
            exact = bool(torch.equal(g2, gm))
            tm = timeit(lambda: ctx.sgemm_drelu(a, g2, mask, dW))  # (masking a masked G again: same traffic, same result)
            print(f"{m}x{n} K={nv} variant {v}: plain {t:.3f} ms ({fl / t / 1e9:.0f} TF/s = {fl / t / 1e9 / 157:.2f} of peak) err {e:.1e} | "
                  f"masked {tm:.3f} ms ({fl / tm / 1e9:.0f} TF/s = {fl / tm / 1e9 / 157:.2f}) err {em:.1e} G masked exactly: {exact}", flush=True)
        ctx.set_option("sgemm_variant", 0)
        del a, g, mask, gm


if __name__ == "__main__":
    main()
