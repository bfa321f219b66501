"""Dense products with an odd-width operand (the 47-class output layer of the products configs; round 5): the register-resident
weight gradient's NUNAL form vs the LDS-tiled kernel (sgemm_variant 36), and the tiled kernel's 16-byte loads at 4-byte
alignment vs the 4-byte loads it used until round 4 (sgemm_variant 37).  One JSON line per shape."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402


def ev(fn, iters=8, warm=3):
    best = None
    for _ in range(2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / iters
        best = t if best is None else min(best, t)
    return best


def main():
    ctx = capi.Context(0)
    nv = 2_449_029
    shapes = [("TN", 128, 47), ("TN", 256, 47), ("TN", 100, 47), ("NT", 256, 47), ("NT", 128, 47), ("NN", 47, 128), ("NN", 47, 256),
              ("TN", 128, 128)]
    for kind, a, b in shapes:
        if kind == "TN":    # C [a x b] = A^T [a x nv] . B [nv x b]
            A, B, C = torch.randn(nv, a, device="cuda"), torch.randn(nv, b, device="cuda"), torch.empty(a, b, device="cuda")
            fn = lambda: ctx.sgemm(A, B, C, True, False)
            byts = 4.0 * nv * (a + b)
        elif kind == "NT":  # C [nv x a] = A [nv x b] . W^T, W [a x b]
            A, B, C = torch.randn(nv, b, device="cuda"), torch.randn(a, b, device="cuda"), torch.empty(nv, a, device="cuda")
            fn = lambda: ctx.sgemm(A, B, C, False, True)
            byts = 4.0 * nv * (a + b)
        else:               # C [nv x a] = A [nv x b] . W, W [b x a]
            A, B, C = torch.randn(nv, b, device="cuda"), torch.randn(b, a, device="cuda"), torch.empty(nv, a, device="cuda")
            fn = lambda: ctx.sgemm(A, B, C, False, False)
            byts = 4.0 * nv * (a + b)
        rec = {"shape": f"{kind} {a} x {b}, long side {nv}", "roof_ms": 1e3 * max(byts / 8e12, 2.0 * nv * a * b / 157.3e12)}
        for name, v in (("default", 0), ("tiled_kernel_for_odd_weight_gradients(36)", 36), ("tiled_kernel_4_byte_loads(37)", 37)):
            ctx.set_option("sgemm_variant", v)
            rec[name + "_ms"] = ev(fn)
        ctx.set_option("sgemm_variant", 0)
        print(json.dumps(rec), flush=True)
        del A, B, C
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
