"""CPU suite: the C-ABI library loads and exports every symbol include/gaib.h declares, and the
ctypes table in graphaibench_amd/capi.py covers exactly that set.  No compute calls (no GPU)."""
import ctypes
import re
from pathlib import Path

from graphaibench_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "gaib.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(gaib_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) > 40
    lib = ctypes.CDLL(str(capi.LIB_PATH))
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, f"libgaib_hip.so does not export: {missing}"


def test_python_table_matches_header():
    assert set(capi.SIGNATURES) == declared_symbols()


def test_header_is_plain_c():
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".c") as f:
        f.write('#include "gaib.h"\nint main(void){return 0;}\n')
        f.flush()
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT/'include'}", f.name],
                       check=True)


def test_error_path_without_gpu():
    """every entry point validates its arguments before touching the device"""
    lib = capi.load()
    assert lib.gaib_version().startswith(b"graphaibench_amd")
    assert lib.gaib_sync(None) != 0
    assert b"NULL" in lib.gaib_last_error()
    assert lib.gaib_spmm(None, None, 0, None, 4, None, None) != 0
    assert lib.gaib_sgemm(None, 0, 0, 1, 1, 1, None, None, 0, None) != 0


def test_rccl_double_covers_every_entry_point_comm_hip_binds():
    """tests/fake_rccl (the strict stand-in the N > 1 tests bind comm.hip's RCCL branch to on a one-GPU box) exports each
    symbol the library dlsym()s -- and nothing that is NOT test infrastructure loads it: the product sources never name it"""
    import ctypes
    import re

    root = Path(__file__).resolve().parent.parent
    src = (root / "graphaibench_amd" / "csrc" / "comm.hip").read_text()
    wanted = re.findall(r'GAIB_SYM\(\w+, "(nccl\w+)"\)', src)
    assert len(wanted) == 11, wanted
    fake = root / "tests" / "fake_rccl" / "librccl_fake.so"
    assert fake.exists(), "python -m graphaibench_amd.build"
    lib = ctypes.CDLL(str(fake))
    for name in wanted:
        assert hasattr(lib, name), name
    for p in list((root / "graphaibench_amd").rglob("*.py")) + list((root / "graphaibench_amd").rglob("*.cpp")) + \
            list((root / "graphaibench_amd").rglob("*.hip")) + [root / "bench.py", root / "__graft_entry__.py"]:
        text = p.read_text()
        assert "librccl_fake" not in text, p
