"""The narrow dense products of the products-shaped epochs: every product of the 3-layer GCN
(hidden 128) and GraphSAGE (hidden 128 / 256) models whose output or inner width is 47 or 100, at 2.45 M rows -- HBM-stream
shapes (<= 17 flop / B) that ran at 2.5-4.9 TB/s in round 5's epoch records.  One JSON line per shape: ms per launch (in-stream
events), algorithmic bytes (both operands once, the output written), the rates against 8 TB/s and against the in-run stream copy.

    python scripts/gemm_narrow.py [--reps 10] [--only TN:128x47 ...]      # timing
    bash scripts/profile_gemm_narrow.sh                                    # + rocprofv3 kernel stats and PMC passes

Shapes as the work table names them, sgemm@MxNxK: NN = A[M x K] . B[K x N]; TN = A[K x M]^T . B[K x N] (K = vertices)."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402

NV = 2_449_029
SHAPES = [  # (form, M, N, K)
    ("TN", 128, 47, NV), ("TN", 256, 47, NV), ("TN", 100, 128, NV), ("TN", 100, 256, NV),
    ("NN", NV, 47, 128), ("NN", NV, 47, 256), ("NN", NV, 128, 100), ("NN", NV, 256, 100), ("NN", NV, 256, 47),
    ("NT", NV, 128, 47), ("NT", NV, 256, 47),          # input gradients of the 47-wide output layer: G[nv x 47] . W[din x 47]^T
    ("NN", NV, 128, 128), ("NN", NV, 256, 256), ("TN", 256, 256, NV),  # the wide ones, for scale
    ("NT", NV, 128, 128), ("NT", NV, 256, 256),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", nargs="*", default=None, help="e.g. TN:128x47 NN:47x128 (form:MxN or form:NxK with the vertex count left out)")
    ap.add_argument("--nv", type=int, default=NV)
    ap.add_argument("--variant", type=int, default=0, help="gaib_set_option sgemm_variant (61: round 5's kernels for the shapes of sgemm_skinny.hip)")
    args = ap.parse_args()
    ctx = capi.Context(0)
    ctx.set_option("sgemm_variant", args.variant)
    peak = ctx.probe_stream_copy(1 << 30, 20)
    for form, M, N, K in SHAPES:
        M, K = (args.nv if M == NV else M), (args.nv if K == NV else K)
        tag = f"{form}:" + "x".join(str(v) for v in (M, N, K) if v != args.nv)
        if args.only and tag not in args.only:
            continue
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1)
        if form == "NN":
            A, B = torch.randn(M, K, device="cuda", generator=gen), torch.randn(K, N, device="cuda", generator=gen)
            call = lambda: ctx.sgemm(A, B, C)
        elif form == "NT":
            A, B = torch.randn(M, K, device="cuda", generator=gen), torch.randn(N, K, device="cuda", generator=gen)
            call = lambda: ctx.sgemm(A, B, C, False, True)
        else:
            A, B = torch.randn(K, M, device="cuda", generator=gen), torch.randn(K, N, device="cuda", generator=gen)
            call = lambda: ctx.sgemm(A, B, C, True, False)
        C = torch.empty(M, N, device="cuda")
        for _ in range(3):
            call()
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(args.reps):
            call()
        ctx.prof_enable(False)
        tab = ctx.prof_table()
        ctx.prof_reset()
        ms = sum(v["ms"] for v in tab.values()) / args.reps
        alg = 4.0 * (M * K + K * N + M * N)
        flops = 2.0 * M * N * K
        roof = max(alg / 8e12, flops / 157.3e12) * 1e3
        print(json.dumps({"shape": tag, "variant": args.variant, "key": f"sgemm@{M}x{N}x{K}", "ms": round(ms, 4), "alg_gb": round(alg / 1e9, 3),
                          "gflop": round(flops / 1e9, 1), "roof_ms": round(roof, 4), "frac": round(roof / ms, 3),
                          "tb_s": round(alg / ms / 1e9, 3), "frac_of_stream_copy": round(alg / (ms * 1e-3) / 1e9 / peak, 3),
                          "stream_copy_gbs": round(peak, 1), "launch_keys": {k: v["count"] // args.reps for k, v in tab.items()}}), flush=True)
        del A, B, C
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
