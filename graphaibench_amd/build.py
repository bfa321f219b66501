"""Build driver: hipcc for the gfx950 kernels + C ABI, g++ for the host C++ mirror, gcc for the
oracle (test infrastructure).  Everything is built IN-TREE so the .so files travel with the
repo snapshot to the GPU box.

    python -m graphaibench_amd.build [--force] [--oracle] [--ref]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "graphaibench_amd"
CSRC = PKG / "csrc"
HOST = PKG / "host"
LIB = PKG / "lib"
INCLUDE = ROOT / "include"
ORACLE = ROOT / "oracle"
REFERENCE = Path("/root/reference")

HIP_SOURCES = ["runtime.hip", "graph.hip", "spmm.hip", "spmm_part.hip", "gat.hip", "sgemm.hip", "sgemm_skinny.hip", "elementwise.hip", "probe.hip", "comm.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # the aggregation kernels reproduce the OpenMP path's separate multiply and add
    # (math_functions.cpp:266-283,336-356): no FMA contraction anywhere in device code.
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-function",
    # a kernel that misses its own __launch_bounds__ occupancy target is an error, not a remark (round 6: 21 sgemm tilings did)
    "-Werror=pass-failed",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm at /opt/rocm)")


def _run(cmd, cwd=None):
    print("+", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], cwd=cwd, check=True)


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_hip(force: bool = False) -> Path:
    """libgaib_hip.so: every HIP kernel + the C ABI of include/gaib.h."""
    LIB.mkdir(exist_ok=True)
    out = LIB / "libgaib_hip.so"
    hipcc = _hipcc()
    headers = [CSRC / "common.h", CSRC / "spmm_core.h", CSRC / "spmm_kernels.h", INCLUDE / "gaib.h"]
    objs, jobs = [], []
    for src in HIP_SOURCES:
        s = CSRC / src
        o = LIB / (src + ".o")
        if force or _stale(o, [s, *headers]):
            jobs.append([hipcc, *HIPCC_FLAGS, f"-I{INCLUDE}", f"-I{CSRC}", "-c", s, "-o", o])
        objs.append(o)
    if jobs:  # the translation units are independent: compile them side by side (a clean build drops from 4 to ~1.5 min)
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(_run, jobs))
    if force or _stale(out, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


def build_host(force: bool = False) -> Path | None:
    """libgaib_gnn.so: the host C++ mirror of the reference layer/operator API over the C ABI."""
    srcs = sorted(p for p in HOST.glob("*.cpp") if p.name != "train_main.cpp")
    if not srcs:
        return None
    out = LIB / "libgaib_gnn.so"
    hdrs = list(INCLUDE.rglob("*.h")) + list(INCLUDE.rglob("*.hh"))
    if force or _stale(out, [*srcs, *hdrs, LIB / "libgaib_hip.so"]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wall",
              f"-I{INCLUDE}", f"-I{INCLUDE}/gnn", f"-I{INCLUDE}/layers", f"-I{INCLUDE}/utils",
              *srcs, f"-L{LIB}", "-lgaib_hip", "-Wl,-rpath,$ORIGIN", "-o", out])
    build_drivers(force)
    return out


def build_drivers(force: bool = False) -> None:
    """bin/gpu_train_{gcn,sage,gat}: the trainer CLI (architecture is a -D choice like in the reference)."""
    bindir = ROOT / "bin"
    bindir.mkdir(exist_ok=True)
    src = HOST / "train_main.cpp"
    hdrs = list(INCLUDE.rglob("*.h")) + list(INCLUDE.rglob("*.hh"))
    for name, flag in (("gcn", None), ("sage", "-DUSE_SAGE"), ("gat", "-DUSE_GAT")):
        exe = bindir / f"gpu_train_{name}"
        if force or _stale(exe, [src, *hdrs, LIB / "libgaib_gnn.so"]):
            cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-Wall", f"-I{INCLUDE}", f"-I{INCLUDE}/gnn",
                   f"-I{INCLUDE}/layers", f"-I{INCLUDE}/utils", src, f"-L{LIB}", "-lgaib_gnn", "-lgaib_hip",
                   "-Wl,-rpath,$ORIGIN/../graphaibench_amd/lib", "-o", exe]
            if flag:
                cmd.insert(1, flag)
            _run(cmd)


def build_oracle(force: bool = False, ref: bool = True) -> Path:
    """oracle/libgnn_oracle.so (+ oracle/_ref when /root/reference is present)."""
    if force:
        _run(["make", "-C", ORACLE, "clean"])
    _run(["make", "-C", ORACLE, "all"])
    if ref and REFERENCE.exists():
        _run(["make", "-C", ORACLE, "ref"])
    fake = ROOT / "tests" / "fake_rccl"  # the strict RCCL double of tests/test_gpu_comm.py (test infrastructure too)
    if fake.exists():
        if force:
            _run(["make", "-C", fake, "clean"])
        _run(["make", "-C", fake, "all"])
    return ORACLE / "libgnn_oracle.so"


def build_all(force: bool = False, oracle: bool = True) -> None:
    build_hip(force)
    build_host(force)
    if oracle:
        build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, oracle=True)
