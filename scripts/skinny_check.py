"""The products of sgemm_skinny.hip against fp64 on the device, at every shape class and at row counts that are no whole number
of tiles / register sets (C =, C +=, relu), next to the same call through the round-5 kernels (sgemm_variant 61).  One JSON
line per case: max |error| over the largest |entry| of the fp64 product, for both; exit 1 if any case is above 2e-5.
    python scripts/skinny_check.py
Test infrastructure (a development tool: its cases live on in tests/test_gpu_ops.py)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402

CASES = [  # (form, M, N, K): NN / NT: rows = M; TN: rows = K
    ("NN", 0, 47, 128), ("NN", 0, 47, 256), ("NN", 0, 41, 128), ("NN", 0, 33, 256), ("NN", 0, 48, 128),
    ("NT", 0, 128, 47), ("NT", 0, 256, 47), ("NT", 0, 128, 48), ("NT", 0, 256, 45),
    ("NN", 0, 128, 100), ("NN", 0, 41, 64), ("NT", 0, 64, 41), ("NN", 0, 256, 100), ("NN", 0, 128, 47), ("NN", 0, 256, 47), ("NN", 0, 128, 128), ("NT", 0, 128, 128), ("NN", 0, 256, 256), ("NT", 0, 256, 256),
    ("TN", 128, 47, 0), ("TN", 256, 47, 0), ("TN", 100, 47, 0), ("TN", 128, 41, 0), ("TN", 200, 33, 0), ("TN", 256, 48, 0),
]
ROWS = [65536, 65537, 65551, 100_003, 300_001, 1_000_000]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="two row counts per case")
    args = ap.parse_args()
    ctx = capi.Context(0)
    rows_list = [65537, 300_001] if args.quick else ROWS
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    bad = 0
    for form, M, N, K in CASES:
        for rows in rows_list:
            for accum, relu in ((False, False), (True, False), (False, True), (True, True)):
                if rows != rows_list[0] and (accum or relu) and rows != rows_list[-1]:
                    continue
                if form == "NN":
                    A = torch.randn(rows, K, device="cuda", generator=gen)
                    B = torch.randn(K, N, device="cuda", generator=gen)
                    want = A.double() @ B.double()
                    tr = (False, False)
                    shape = (rows, N)
                elif form == "NT":
                    A = torch.randn(rows, K, device="cuda", generator=gen)
                    B = torch.randn(N, K, device="cuda", generator=gen)
                    want = A.double() @ B.double().t()
                    tr = (False, True)
                    shape = (rows, N)
                else:
                    A = torch.randn(rows, M, device="cuda", generator=gen)
                    B = torch.randn(rows, N, device="cuda", generator=gen)
                    want = A.double().t() @ B.double()
                    tr = (True, False)
                    shape = (M, N)
                C0 = torch.randn(*shape, device="cuda", generator=gen)
                if accum:
                    want = want + C0.double()
                if relu:
                    want = torch.clamp(want, min=0)
                scale = float(want.abs().max())
                errs = {}
                for variant in (0, 61):
                    ctx.set_option("sgemm_variant", variant)
                    C = C0.clone()
                    guard = torch.full((4096,), 12345.0, device="cuda")  # (allocated right behind C more often than not)
                    ctx.sgemm(A, B, C, tr[0], tr[1], accum=accum, relu=relu)
                    ctx.sync()
                    errs[variant] = float((C.double() - want).abs().max()) / scale
                    assert bool((guard == 12345.0).all())
                ctx.set_option("sgemm_variant", 0)
                ok = errs[0] <= 2e-5
                bad += 0 if ok else 1
                print(json.dumps({"form": form, "M": M or rows, "N": N, "K": K or rows, "accum": accum, "relu": relu,
                                  "err_skinny": errs[0], "err_round5": errs[61], "ok": ok}), flush=True)
                del A, B, C, C0, want
    print(json.dumps({"bad": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
