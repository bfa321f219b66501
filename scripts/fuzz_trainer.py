"""Randomised check of the trainer CLI as N processes against the same run as one process: random small datasets in the
reference's binary format (vertex counts 12 .. 4000, isolated vertices, feature / class counts, the contiguous 8 / 2 / 90 %
split -- so most ranks own no training vertex), bin/gpu_train_{gcn,sage,gat} with random hidden widths and layer counts,
2 .. 4 ranks on one GPU over the peer-to-peer transport or comm.hip's RCCL branch bound to tests/fake_rccl.  Rank 0's loss /
accuracy lines must equal the single-process run's.
    python scripts/fuzz_trainer.py [--seconds 150] [--seed 0]
Test infrastructure (a development tool: what it finds becomes a case in tests/)."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from util import random_graph  # noqa: E402


def write_dataset(root: Path, n, avg, F, C, seed):
    rng = np.random.default_rng(seed)
    rp, ci = random_graph(n, avg, seed=seed, power_law=bool(seed & 1))
    d = root / "rnd"
    if d.exists():
        shutil.rmtree(d)
    d.mkdir(parents=True)
    rp.astype(np.int64).tofile(d / "graph.vertex.bin")
    ci.astype(np.uint32).tofile(d / "graph.edge.bin")
    labels = rng.integers(0, C, n)
    labels.astype(np.uint8).tofile(d / "graph.vlabel.bin")
    x = rng.standard_normal((n, F)).astype(np.float32) * 0.5
    x[np.arange(n), labels % F] += 1.5
    x.tofile(d / "graph.feats.bin")
    tr, va = max(2, int(0.08 * n)), max(3, int(0.10 * n))
    max_degree = int(np.diff(rp).max()) if n else 0
    meta = [n, len(ci), 4, 8, 1, 2, max_degree, F, C, 0, 0, tr, tr, tr, va, va - tr, va, n, n - va]
    (d / "graph.meta.txt").write_text("\n".join(str(v) for v in meta) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    fake = ROOT / "tests" / "fake_rccl" / "librccl_fake.so"
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GAIB_RANK", "GAIB_WORLD", "GAIB_COMM",
                                                                "GAIB_DEVICE", "GAIB_COMM_ID_FILE", "GAIB_RANKS", "GAIB_RCCL_LIB")}
    t0, n_cases, fails = time.time(), 0, []
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / "data"
        while time.time() - t0 < args.seconds:
            n = int(rng.choice([12, 40, 300, 1500, 4000]))
            avg = float(rng.choice([0.6, 3, 10]))
            F, C = int(rng.choice([8, 33, 100])), int(rng.choice([2, 7, 41]))
            arch = str(rng.choice(["gcn", "sage", "gat"]))
            hidden = int(rng.choice([16, 64])) if arch == "gat" else int(rng.choice([8, 16, 100, 128]))
            layers = int(rng.choice([2, 3]))
            world = int(rng.choice([2, 3, 4]))
            transport = str(rng.choice(["ipc", "rccl"]))
            dseed = int(rng.integers(1 << 30))
            cfg = dict(n=n, avg=avg, F=F, C=C, arch=arch, hidden=hidden, layers=layers, world=world, transport=transport, dseed=dseed)
            try:
                write_dataset(root, n, avg, F, C, dseed)
                exe = ROOT / "bin" / f"gpu_train_{arch}"
                cmd = [str(exe), "rnd", "4", "2", "softmax", str(hidden), "0", "0", "0.01", str(layers), "0", "2", "0"]
                base = dict(clean, DATASET_PATH=str(root) + "/", GAIB_COMM_TIMEOUT_S="30", GAIB_FAKE_RCCL_TIMEOUT_S="30")
                if arch == "gat":
                    base["GAIB_GAT_HEADS"] = str(int(rng.choice([1, 4, 8])))
                    cfg["heads"] = base["GAIB_GAT_HEADS"]
                single = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=120)
                if single.returncode != 0:
                    raise AssertionError("single process failed: " + (single.stderr or single.stdout)[-300:])
                env = dict(base, GAIB_RANKS=str(world), GAIB_RANKS_DEADLINE_S="90")
                if transport == "rccl":
                    env.update(GAIB_COMM="rccl", GAIB_RCCL_LIB=str(fake))
                multi = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=200)
                if multi.returncode != 0:
                    raise AssertionError(f"{world} ranks failed ({multi.returncode}): " + (multi.stderr or multi.stdout)[-400:])
                pat = r"train_loss ([0-9.]+) train_acc ([0-9.]+)"
                want, got = re.findall(pat, single.stdout), re.findall(pat, multi.stdout)
                if len(want) != 4 or len(got) != 4:
                    raise AssertionError(f"log lines: {len(want)} vs {len(got)}")
                for (wl, wa), (gl, ga) in zip(want, got):
                    if abs(float(wl) - float(gl)) > 3e-3 or abs(float(wa) - float(ga)) > 0.02 + 1.5 / max(2, int(0.08 * n)):
                        raise AssertionError(f"curves differ: one process {want}, {world} ranks {got}")
                pt = r"Test accuracy: ([0-9.]+)"
                if abs(float(re.search(pt, single.stdout).group(1)) - float(re.search(pt, multi.stdout).group(1))) > 0.02 + 1.5 / max(1, n - int(0.1 * n)):
                    raise AssertionError("test accuracy differs")
            except Exception as e:  # noqa: BLE001
                fails.append(dict(cfg, error=f"{type(e).__name__}: {e}"[:600]))
                print("FAIL", json.dumps(fails[-1]), flush=True)
            n_cases += 1
            if n_cases % 5 == 0:
                print(f"{n_cases} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps({"cases": n_cases, "failures": len(fails), "seconds": round(time.time() - t0, 1)}))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
