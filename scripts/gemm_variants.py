"""SGEMM variants A/B at the layer shapes (NN / NT / TN, N_v x d x d) with a correctness check against torch."""
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
    for d in (128, 256):
        x = torch.randn(nv, d, device="cuda")
        g = torch.randn(nv, d, device="cuda")
        W = torch.randn(d, d, device="cuda") * 0.1
        y = torch.empty(nv, d, device="cuda")
        dW = torch.empty(d, d, device="cuda")
        ref_nn = (x[:4096].double() @ W.double())
        ref_tn = (x.double().T @ g.double())
        for variant in [int(v) for v in sys.argv[1:]] or [0, 20, 21]:
            ctx.set_option("sgemm_variant", variant)
            t_nn = timeit(lambda: ctx.sgemm(x, W, y))
            e_nn = ((y[:4096].double() - ref_nn).norm() / ref_nn.norm()).item()
            t_nt = timeit(lambda: ctx.sgemm(x, W, y, False, True))
            e_nt = ((y[:4096].double() - x[:4096].double() @ W.double().T).norm() / ref_nn.norm()).item()
            t_acc = timeit(lambda: ctx.sgemm(x, W, y, accum=True, relu=True))  # the C += / relu epilogue
            t_tn = timeit(lambda: ctx.sgemm(x, g, dW, True, False))
            e_tn = ((dW.double() - ref_tn).norm() / ref_tn.norm()).item()
            fl = 2.0 * nv * d * d
            print(f"d={d} variant={variant}: NN {t_nn:.3f} ms ({fl/t_nn/1e9:.0f} TF) err {e_nn:.1e} | NT {t_nt:.3f} ms err {e_nt:.1e} | NN+= {t_acc:.3f} | "
                  f"TN {t_tn:.3f} ms ({fl/t_tn/1e9:.0f} TF) err {e_tn:.1e}", flush=True)
        ctx.set_option("sgemm_variant", 0)
        t_mm = timeit(lambda: torch.mm(x, W, out=y))
        t_mt = timeit(lambda: torch.mm(x, W.T, out=y))
        print(f"d={d} torch.mm (rocBLAS / hipBLASLt): NN {t_mm:.3f} ms | NT {t_mt:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
