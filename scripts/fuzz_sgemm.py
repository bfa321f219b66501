"""Randomised sweep of the dense products against fp64 on the device: gaib_sgemm (NN / NT / TN, accumulate, relu) and
gaib_sgemm_drelu (weight gradient with the d_relu folded in, mask applied IN PLACE to G) over shapes that cross every
dispatch boundary -- tall-skinny with K or M in the millions (the layers' shapes), widths 1 .. 300 incl. 129 .. 256 (quadrant
teams, the LDS ring), odd sizes, K below / above the split-K and streaming thresholds -- and every sgemm_variant.
    python scripts/fuzz_sgemm.py [--seconds 120] [--seed 0]
Test infrastructure (a development tool: what it finds becomes a case in tests/)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from graphaibench_amd import capi  # noqa: E402

SMALL = [1, 2, 3, 7, 16, 17, 31, 47, 64, 65, 100, 128, 129, 160, 192, 200, 255, 256, 257, 300]
BIG = [1000, 4097, 65536, 300_001, 1_000_003]
VARIANTS = [0, 0, 0, 0, 2, 10, 11, 12, 29, 30, 32, 33, 34, 35, 38, 39, 40, 41, 44, 50, 61, 64, 66, 67]  # dispatch switches of sgemm.hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = capi.Context(0)
    t0, n_cases, fails, worst = time.time(), 0, [], 0.0
    while time.time() - t0 < args.seconds:
        shape = int(rng.integers(4))
        if shape == 0:    # rows x din . din x dout (the layers' forward / input-gradient products)
            M, N, K = int(rng.choice(BIG)), int(rng.choice(SMALL)), int(rng.choice(SMALL))
        elif shape == 1:  # weight gradient: din x rows . rows x dout
            M, N, K = int(rng.choice(SMALL)), int(rng.choice(SMALL)), int(rng.choice(BIG))
        elif shape == 2:  # small everything (cora-sized layers)
            M, N, K = (int(rng.choice(SMALL + [500, 2708])) for _ in range(3))
        else:
            M, N, K = int(rng.integers(1, 3000)), int(rng.integers(1, 300)), int(rng.integers(1, 3000))
        variant = int(rng.choice(VARIANTS))
        ctx.set_option("sgemm_variant", variant)
        drelu = shape == 1 and bool(rng.integers(2))
        cfg = dict(M=M, N=N, K=K, variant=variant)
        try:
            if drelu:
                accum = bool(rng.integers(2))
                cfg.update(call="sgemm_drelu", accum=accum)
                A = torch.randn(K, M, device="cuda")
                G = torch.randn(K, N, device="cuda")
                mask = torch.randn(K, N, device="cuda")
                mask[torch.rand(K, N, device="cuda") < 0.1] = 0.0  # exact zeros: (mask > 0) is strict
                C0 = torch.randn(M, N, device="cuda")
                C = C0.clone()
                Gm = G * (mask > 0)
                want = A.double().t() @ Gm.double() + (C0.double() if accum else 0)
                ctx.sgemm_drelu(A, G, mask, C, accum=accum)
                ctx.sync()
                if not torch.equal(G, Gm):
                    raise AssertionError("masked G differs from G * (mask > 0)")
            else:
                tA, tB = [(False, False), (False, True), (True, False)][int(rng.integers(3))]  # (TT is not on the path: refused)
                accum, relu = bool(rng.integers(2)), bool(rng.integers(2))
                cfg.update(call="sgemm", transA=tA, transB=tB, accum=accum, relu=relu)
                A = torch.randn((K, M) if tA else (M, K), device="cuda")
                B = torch.randn((N, K) if tB else (K, N), device="cuda")
                C0 = torch.randn(M, N, device="cuda")
                C = C0.clone()
                want = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double())
                if accum:
                    want += C0.double()
                if relu:
                    want.clamp_(min=0)
                ctx.sgemm(A, B, C, transA=tA, transB=tB, accum=accum, relu=relu)
                ctx.sync()
            if not torch.isfinite(C).all():
                raise AssertionError("non-finite values")
            # fp32 accumulation over K terms of unit variance: ~ sqrt(K) 2^-24 of the row scale; bound with slack (the
            # maximum is over up to 3e8 entries: 3e-6 was exceeded by <= 9 % in 16 of 30 000 cases, all at K <= 257)
            tol = 6e-6 * max(1.0, np.sqrt(K) / 8) * np.sqrt(K)
            err = float((C.double() - want).abs().max())
            worst = max(worst, err / tol)
            if err > tol:
                raise AssertionError(f"max |err| {err:.3e} > {tol:.3e}")
        except Exception as e:  # noqa: BLE001
            fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:300]))
            print("FAIL", json.dumps(fails[-1]), flush=True)
        n_cases += 1
        if n_cases % 100 == 0:
            print(f"{n_cases} cases, {len(fails)} failures, worst err / tol {worst:.2f}, {time.time() - t0:.0f} s", flush=True)
    ctx.set_option("sgemm_variant", 0)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "worst_err_over_tol": worst, "seconds": round(time.time() - t0, 1)}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
