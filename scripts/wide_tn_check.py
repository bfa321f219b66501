"""wide_tn_kernel (sgemm_skinny.hip) against fp64 and against the 32 x 32 register-resident kernels it replaces (sgemm_variant 67):
plain and with the d_relu mask folded in (G rewritten in place), K no whole number of register sets, C = / C +=; then ms per launch
at 2.45 M rows for both.   python scripts/wide_tn_check.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402

ctx = capi.Context(0)
gen = torch.Generator(device="cuda")
gen.manual_seed(5)
bad = 0
for M, N in ((100, 128), (100, 256), (128, 128), (128, 256), (72, 128), (112, 256), (116, 128)):
    for K in (65537, 70003, 300001):
        for masked in (False, True):
            for accum in (False, True):
                A = torch.randn(K, M, device="cuda", generator=gen)
                G0 = torch.randn(K, N, device="cuda", generator=gen)
                mask = torch.randn(K, N, device="cuda", generator=gen)
                mask[torch.rand(K, N, device="cuda", generator=gen) < 0.1] = 0.0
                C0 = torch.randn(M, N, device="cuda", generator=gen)
                Gm = G0 * (mask > 0) if masked else G0
                want = A.double().t() @ Gm.double() + (C0.double() if accum else 0)
                errs = {}
                for variant in (0, 67):
                    ctx.set_option("sgemm_variant", variant)
                    flat = torch.full((K * N + 1024,), 555.0, device="cuda")  # G at the start of its allocation, a canary behind it
                    G = flat[:K * N].view(K, N)
                    G.copy_(G0)
                    C = C0.clone()
                    if masked:
                        ctx.sgemm_drelu(A, G, mask, C, accum=accum)
                    else:
                        ctx.sgemm(A, G, C, True, False, accum=accum)
                    ctx.sync()
                    errs[variant] = float((C.double() - want).abs().max() / want.abs().max())
                    ok_g = bool(torch.equal(G, Gm)) and bool((flat[K * N:] == 555.0).all())
                    if not ok_g:
                        errs[variant] = 1.0
                ctx.set_option("sgemm_variant", 0)
                ok = errs[0] <= 2e-5
                bad += 0 if ok else 1
                print(json.dumps({"M": M, "N": N, "K": K, "masked": masked, "accum": accum, "err_wide": errs[0], "err_32x32": errs[67], "ok": ok}), flush=True)
                del A, G0, mask, C0, Gm, want
print(json.dumps({"bad": bad}), flush=True)
nv = 2449029


def ev(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for M, N in ((100, 128), (100, 256), (128, 128), (128, 256)):
    A = torch.randn(nv, M, device="cuda", generator=gen)
    G = torch.randn(nv, N, device="cuda", generator=gen)
    mask = torch.randn(nv, N, device="cuda", generator=gen)
    C = torch.empty(M, N, device="cuda")
    for masked in (False, True):
        row = {"M": M, "N": N, "masked": masked}
        for rnd in range(2):
            for variant in (0, 67):
                ctx.set_option("sgemm_variant", variant)
                fn = (lambda: ctx.sgemm_drelu(A, G, mask, C)) if masked else (lambda: ctx.sgemm(A, G, C, True, False))
                t = ev(fn)
                key = "wide_ms" if variant == 0 else "r5_32x32_ms"
                row[key] = round(min(row.get(key, 1e9), t), 4)
        ctx.set_option("sgemm_variant", 0)
        print(json.dumps(row), flush=True)
    del A, G, mask, C
sys.exit(1 if bad else 0)
