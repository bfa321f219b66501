"""CPU suite: "the GCN/GraphSAGE/GAT drivers in src/gnn link unchanged" (north star).
Compiles the REFERENCE's own src/gnn/train.cpp and src/gnn/net.cpp, from where they lie and
unmodified, against THIS repo's include/{gnn,layers,utils} (plus the reference's own driver header
net.h, reached through a symlink so that no other reference header is visible) and links them with
libgaib_gnn.so + libgaib_hip.so.  The object list is the one INTEGRATION.md documents (the `gpu_train_gcn:` rule of its Makefile hunk), read from
that file: what a maintainer is told to build is what is built here.
Skipped where /root/reference does not exist (GPU box)."""
import os
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
LIB = ROOT / "graphaibench_amd" / "lib"


def documented_objects():
    """the prerequisites of the `gpu_train_gcn:` rule in INTEGRATION.md's Makefile hunk -> reference source files"""
    text = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"^gpu_train_gcn:(.*)$", text, re.M)
    assert m, "INTEGRATION.md lost its gpu_train_gcn rule"
    objs = m.group(1).split("#")[0].split()
    assert objs and all(re.fullmatch(r"\w+\.o", o) for o in objs), objs  # no ellipsis, no placeholders
    return [o[:-2] + ".cpp" for o in objs]


def test_documented_object_list_is_exact():
    srcs = documented_objects()
    assert "train.cpp" in srcs and "net.cpp" in srcs
    if REF.exists():
        assert all((REF / "src" / "gnn" / s).exists() for s in srcs), srcs


@pytest.mark.parametrize("flag,name", [("", "gcn"), ("-DUSE_SAGE", "sage"), ("-DUSE_GAT", "gat")])
def test_reference_driver_compiles_and_links_unchanged(tmp_path, flag, name):
    if not REF.exists():
        pytest.skip("no /root/reference on this machine")
    assert (LIB / "libgaib_gnn.so").exists() and (LIB / "libgaib_hip.so").exists(), "run graphaibench_amd.build"
    hdr = tmp_path / "driver_hdr"
    hdr.mkdir()
    os.symlink(REF / "include" / "gnn" / "net.h", hdr / "net.h")  # the driver's own header, nothing else
    inc = [f"-I{ROOT/'include'}", f"-I{ROOT/'include'/'gnn'}", f"-I{ROOT/'include'/'layers'}",
           f"-I{ROOT/'include'/'utils'}", f"-I{hdr}"]
    objs = []
    for src in documented_objects():
        o = tmp_path / (src + ".o")
        cmd = ["g++", "-O1", "-std=c++17", "-fopenmp", *inc, "-c", str(REF / "src" / "gnn" / src), "-o", str(o)]
        if flag:
            cmd.insert(1, flag)
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        objs.append(str(o))
    exe = tmp_path / f"gpu_train_{name}"
    r = subprocess.run(["g++", "-fopenmp", *objs, f"-L{LIB}", "-lgaib_gnn", "-lgaib_hip", f"-Wl,-rpath,{LIB}",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the binary starts and prints the reference's own usage text (no GPU touched: argc check comes first)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert "Usage: ./train data num_epochs" in r.stdout
