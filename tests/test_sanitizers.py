"""CPU sanitizer runs ("race detection / sanitizers"): the oracle's OpenMP restatement under
AddressSanitizer + UBSan (gcc) and under ThreadSanitizer (clang + libomp + Archer), and the host-only C++ of the
product (dataset reader, GraphSAINT sampler, vertex-range partition builder) under ASan + UBSan.  GPU code cannot be
sanitized on this pool (no GPU ASan / xnack), so this is the CPU half only."""
import glob
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ORACLE = ROOT / "oracle"
LLVM = Path(os.environ.get("LLVM", "/opt/rocm/lib/llvm"))
BAD = ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:", "ERROR: LeakSanitizer")


def _run(env_extra, args, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout + r.stderr


@pytest.fixture(scope="module")
def san_libs():
    r = subprocess.run(["make", "-C", str(ORACLE), "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("make -C oracle san failed:\n" + r.stdout + r.stderr)
    return ORACLE / "_san"


def test_oracle_asan_ubsan(san_libs):
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert Path(asan_rt).exists(), asan_rt
    rc, out = _run({"LD_PRELOAD": asan_rt, "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1",
                    "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                    "GNN_ORACLE_LIB": str(san_libs / "libgnn_oracle_asan.so")},
                   [sys.executable, str(ROOT / "tests" / "san_workload.py")])
    assert rc == 0 and "san_workload done" in out and not any(b in out for b in BAD), out[-4000:]


def test_oracle_asan_catches_a_planted_overrun(san_libs):
    """the harness really runs the instrumented library: a column id past the feature table must be reported"""
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from oracle import binding as orc; "
            "g = orc.Graph(np.array([0, 1], np.int64), np.array([40], np.uint32)); "
            "orc.sage_aggregate(g, np.ones((1, 4), np.float32))" % str(ROOT))
    rc, out = _run({"LD_PRELOAD": asan_rt, "ASAN_OPTIONS": "detect_leaks=0",
                    "GNN_ORACLE_LIB": str(san_libs / "libgnn_oracle_asan.so")}, [sys.executable, "-c", code])
    assert rc != 0 and "AddressSanitizer" in out, out[-2000:]


def test_oracle_tsan_openmp(san_libs):
    rts = glob.glob(str(LLVM / "lib" / "clang" / "*" / "lib" / "linux" / "libclang_rt.tsan-x86_64.so"))
    archer = LLVM / "lib" / "libarcher.so"
    if not rts or not archer.exists():
        pytest.skip("LLVM ThreadSanitizer runtime / Archer not in this image")
    rc, out = _run({"LD_PRELOAD": rts[0], "OMP_TOOL_LIBRARIES": str(archer),
                    "TSAN_OPTIONS": "report_signal_unsafe=0 ignore_noninstrumented_modules=1 exitcode=66",
                    "GNN_ORACLE_LIB": str(san_libs / "libgnn_oracle_tsan.so")},
                   [sys.executable, str(ROOT / "tests" / "san_workload.py")])
    assert rc == 0 and "san_workload done" in out and not any(b in out for b in BAD), out[-6000:]


def test_host_cpp_asan_ubsan(tmp_path):
    """the host-only C++ of the product under ASan + UBSan (tests/host_san_main.cpp): binary reader, LearningGraph host
    methods, GraphSAINT sampler, vertex-range partition builder with the GAT structures.  The GPU libraries are linked to
    resolve symbols only; nothing touches a device."""
    lib = ROOT / "graphaibench_amd" / "lib"
    if not (lib / "libgaib_hip.so").exists():
        pytest.skip("libgaib_hip.so not built")
    exe = tmp_path / "host_san"
    inc = [f"-I{ROOT / 'include'}"] + [f"-I{ROOT / 'include' / d}" for d in ("gnn", "layers", "utils")]
    srcs = [str(ROOT / "tests" / "host_san_main.cpp")] + [str(ROOT / "graphaibench_amd" / "host" / f"{n}.cpp")
                                                          for n in ("reader", "sampler", "partition", "lgraph", "context")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", *inc, *srcs, f"-L{lib}", "-lgaib_hip",
                        f"-Wl,-rpath,{lib}", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    data = tmp_path / "data"
    (data / "cora").mkdir(parents=True)
    rc, out = _run({"DATASET_PATH": str(data) + "/", "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1",
                    "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1", "OMP_NUM_THREADS": "4"},
                   [str(exe), str(data) + "/"])
    assert rc == 0 and "host_san_main done" in out and not any(b in out for b in BAD), out[-4000:]
