"""CPU suite: oracle vs the committed golden fixtures (tests/golden/, made by make_golden.py from
the REAL reference: oracle/_ref = lgraph.cpp + reader.cpp compiled unmodified, and libstdc++'s
RNG for init_glorot).  Bit-exact."""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as orc

GOLD = Path(__file__).resolve().parent / "golden"


def load_topology(name):
    d = GOLD / name
    meta = (d / "graph.meta.txt").read_text().split()
    nv, ne = int(meta[0]), int(meta[1])
    rp = np.fromfile(d / "graph.vertex.bin", np.int64)   # int64 on disk (reader.cpp:446)
    ci = np.fromfile(d / "graph.edge.bin", np.uint32)
    assert len(rp) == nv + 1 and len(ci) == ne
    return d, rp, ci


@pytest.mark.parametrize("name", ["tester", "cora", "citeseer"])
def test_add_selfloop_bit_exact(name):
    d, rp, ci = load_topology(name)
    g = orc.Graph(rp, ci).add_selfloop()
    assert np.array_equal(g.rowptr, np.load(d / "ref_selfloop_rowptr.npy").astype(np.int64))
    assert np.array_equal(g.colidx, np.load(d / "ref_selfloop_colidx.npy"))


@pytest.mark.parametrize("name", ["tester", "cora", "citeseer"])
def test_vertex_and_edge_data_bit_exact(name):
    d, rp, ci = load_topology(name)
    g = orc.Graph(rp, ci)
    assert np.array_equal(g.vertex_data().view(np.uint32), np.load(d / "ref_vertex_data.npy").view(np.uint32))
    assert np.array_equal(g.edge_data().view(np.uint32), np.load(d / "ref_edge_data.npy").view(np.uint32))
    gs = g.add_selfloop()
    assert np.array_equal(gs.vertex_data().view(np.uint32),
                          np.load(d / "ref_selfloop_vertex_data.npy").view(np.uint32))
    assert np.array_equal(gs.edge_data().view(np.uint32),
                          np.load(d / "ref_selfloop_edge_data.npy").view(np.uint32))


def test_reader_roundtrip_cora():
    """what Reader::bin_read_graph / bin_read_vlabels produced from the same files"""
    d, rp, ci = load_topology("cora")
    r = np.load(d / "ref_reader.npz")
    assert np.array_equal(r["rowptr"].astype(np.int64), rp)  # int64 -> uint32 narrowing (Q13)
    assert np.array_equal(r["colidx"], ci)
    assert np.array_equal(r["labels"], np.fromfile(d / "graph.vlabel.bin", np.uint8))
    assert int(r["num_cls"]) == 7


@pytest.mark.parametrize("f", sorted(p.name for p in GOLD.glob("glorot_*.npy")))
def test_init_glorot_bit_exact(f):
    want = np.load(GOLD / f)
    dims, seed = f[len("glorot_"):-len(".npy")].split("_s")
    dx, dy = (int(v) for v in dims.split("x"))
    got = orc.init_glorot(dx, dy, int(seed))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_live_reference_if_present():
    """where oracle/_ref exists (build container), compare on a fresh random graph too"""
    import ctypes as C
    from util import random_graph

    ref = orc.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rp, ci = random_graph(5000, 12, seed=99, power_law=True)
    nv, ne = len(rp) - 1, len(ci)
    rp32 = rp.astype(np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rp2, ci2 = np.zeros(nv + 1, np.uint32), np.zeros(ne + nv, np.uint32)
    ref.ref_add_selfloop(C.c_uint32(nv), C.c_uint32(ne), p(rp32), p(ci), p(rp2), p(ci2))
    g = orc.Graph(rp, ci).add_selfloop()
    assert np.array_equal(g.rowptr, rp2.astype(np.int64)) and np.array_equal(g.colidx, ci2)
    vd, ed = np.zeros(nv, np.float32), np.zeros(ne + nv, np.float32)
    ref.ref_vertex_data(C.c_uint32(nv), C.c_uint32(ne + nv), p(rp2), p(ci2), p(vd))
    ref.ref_edge_data(C.c_uint32(nv), C.c_uint32(ne + nv), p(rp2), p(ci2), p(ed))
    assert np.array_equal(g.vertex_data().view(np.uint32), vd.view(np.uint32))
    assert np.array_equal(g.edge_data().view(np.uint32), ed.view(np.uint32))
