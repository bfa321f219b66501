"""What the fp32 matrix peak is worth under sustained load: the streaming SGEMM (2.45 M x 256 x 256) and torch.mm in a loop
for a few seconds each, `rocm-smi` clocks and power sampled mid-run by a child process (as bench.py's sustained leg does)."""
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import capi  # noqa: E402


def sample(out, delay):
    time.sleep(delay)
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20)
        out["smi"] = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power"))]
    except Exception as e:  # noqa: BLE001
        out["smi"] = [repr(e)]


def run(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    box = {}
    th = threading.Thread(target=sample, args=(box, seconds / 2), daemon=True)
    th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    el = time.perf_counter() - t0
    th.join(timeout=25)
    ms = el / n * 1e3
    print(f"{name}: {ms:.3f} ms per call over {el:.1f} s = {2.0 * 2449029 * 256 * 256 / ms / 1e9:.0f} TF/s", flush=True)
    for l in box.get("smi", []):
        print("    ", l, flush=True)


def main():
    ctx = capi.Context(0)
    nv, d = 2449029, 256
    x = torch.randn(nv, d, device="cuda")
    W = torch.randn(d, d, device="cuda") * 0.1
    y = torch.empty(nv, d, device="cuda")
    g = torch.randn(nv, d, device="cuda")
    dW = torch.empty(d, d, device="cuda")
    run("warm-up (weight gradient TN)", lambda: ctx.sgemm(x, g, dW, True, False), seconds=8.0)
    run("streaming SGEMM NN", lambda: ctx.sgemm(x, W, y))
    run("torch.mm NN (rocBLAS / hipBLASLt)", lambda: torch.mm(x, W, out=y))
    run("weight gradient TN (LDS-tiled, split-K)", lambda: ctx.sgemm(x, g, dW, True, False))
    run("streaming SGEMM NN again", lambda: ctx.sgemm(x, W, y))


if __name__ == "__main__":
    main()
