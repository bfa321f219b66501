"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists).  Fixtures are DATA only:

  <name>/graph.{meta.txt,vertex.bin,edge.bin,vlabel.bin}   topology/labels of the reference's own
        tiny datasets (inputs/tester, inputs/cora, inputs/citeseer) -- byte copies of data files.
  <name>/ref_*.npy    outputs of the REAL reference (oracle/_ref = lgraph.cpp + reader.cpp compiled
        unmodified): add_selfloop, compute_vertex_data, compute_edge_data, reader round trip.
  sampler_*.npz       outputs of the REAL reference's Sampler::select_vertices + generateSubgraph and
        LearningGraph::generate_masked_graph (src/gnn/sampler.cpp compiled unmodified into oracle/_ref) on seeded
        graphs from tests/util.random_graph: kept vertex ids + the induced, re-indexed subgraph.
  partition_p*.npz    outputs of the REAL reference's PartitionedGraph::edgecut_induced_partition1D
        (src/partitioner/graph_partition.cc + src/common/{graph,VertexSet}.cc compiled unmodified): owned range,
        local -> global id map and local CSR of every subgraph.
  glorot_*.npy        outputs of libstdc++'s std::default_random_engine +
        std::uniform_real_distribution<float>, the two std calls init_glorot makes
        (math_functions.cpp:11-18), produced by the 12-line program below.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

GLOROT_CPP = r"""
#include <random>
#include <cstdio>
#include <cmath>
#include <cstdlib>
int main(int argc, char** argv) {
  size_t dx = atoi(argv[1]), dy = atoi(argv[2]); unsigned seed = atoi(argv[3]);
  float init_range = sqrt(6.0 / (dx + dy));
  std::default_random_engine rng(seed);
  std::uniform_real_distribution<float> dist(-init_range, init_range);
  for (size_t i = 0; i < dx * dy; ++i) { float v = dist(rng); fwrite(&v, 4, 1, stdout); }
}
"""

# tag -> (vertices, average degree, graph seed, training prefix, subgraph size n, sampler seed); frontier = 3000
SAMPLER_CASES = {
    "walk_rebuild": (12000, 8, 21, 8000, 6000, 0),   # 3000 pops: the dashboard is rebuilt on the way
    "short_walk": (9000, 6, 22, 5000, 3400, 5),
    "no_walk": (6000, 6, 23, 6000, 2000, 2),          # n < frontier size: only the initial picks
}

PARTITION_GRAPH = (1203, 7, 5)  # vertices (not a multiple of the part counts), average degree, seed
PARTITION_CASES = (2, 3, 8)

GLOROT_CASES = [(16, 7, 1), (1433, 16, 1), (128, 128, 1), (128, 128, 2), (64, 1, 2), (64, 1, 3), (100, 47, 1)]


def main():
    assert REF.exists(), "needs /root/reference"
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "ref"], check=True)
    # DATASET_PATH must be set before the reference library is loaded (configs.h:5)
    os.environ["DATASET_PATH"] = str(REF / "inputs") + "/"
    ref = C.CDLL(str(ROOT / "oracle" / "_ref" / "libref_lgraph.so"))
    for name in ["tester", "cora", "citeseer"]:
        d = HERE / name
        d.mkdir(exist_ok=True)
        for f in ["graph.meta.txt", "graph.vertex.bin", "graph.edge.bin", "graph.vlabel.bin"]:
            shutil.copyfile(REF / "inputs" / name / f, d / f)
        meta = (d / "graph.meta.txt").read_text().split()
        nv, ne = int(meta[0]), int(meta[1])
        rp64 = np.fromfile(d / "graph.vertex.bin", np.int64)
        ci = np.fromfile(d / "graph.edge.bin", np.uint32)
        assert len(rp64) == nv + 1 and len(ci) == ne and rp64[-1] == ne
        rp = rp64.astype(np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rp2 = np.zeros(nv + 1, np.uint32)
        ci2 = np.zeros(ne + nv, np.uint32)
        ref.ref_add_selfloop(C.c_uint32(nv), C.c_uint32(ne), p(rp), p(ci), p(rp2), p(ci2))
        np.save(d / "ref_selfloop_rowptr.npy", rp2)
        np.save(d / "ref_selfloop_colidx.npy", ci2)
        for tag, r_, c_, n_e in [("", rp, ci, ne), ("selfloop_", rp2, ci2, ne + nv)]:
            vd = np.zeros(nv, np.float32)
            ed = np.zeros(n_e, np.float32)
            ref.ref_vertex_data(C.c_uint32(nv), C.c_uint32(n_e), p(r_), p(c_), p(vd))
            ref.ref_edge_data(C.c_uint32(nv), C.c_uint32(n_e), p(r_), p(c_), p(ed))
            np.save(d / f"ref_{tag}vertex_data.npy", vd)
            np.save(d / f"ref_{tag}edge_data.npy", ed)
        # reader round trip through Reader::bin_read_graph / bin_read_vlabels
        if name == "cora":  # tester has no classes, citeseer no split rows: Reader asserts (reader.cpp:349,433-437)
            nv_, ne_, ncls, flen = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()
            ref.ref_read_dataset(name.encode(), C.byref(nv_), C.byref(ne_), None, None, C.byref(ncls), None,
                                 C.byref(flen), None, 0)
            rrp = np.zeros(nv_.value + 1, np.uint32)
            rci = np.zeros(ne_.value, np.uint32)
            lab = np.zeros(nv_.value, np.uint8)
            ref.ref_read_dataset(name.encode(), C.byref(nv_), C.byref(ne_), p(rrp), p(rci), C.byref(ncls), p(lab),
                                 C.byref(flen), None, 0)
            np.savez(d / "ref_reader.npz", rowptr=rrp, colidx=rci, labels=lab, num_cls=ncls.value,
                     feat_len=flen.value)
    # sampler: same seeded graphs the tests rebuild (tests/util.random_graph)
    sys.path.insert(0, str(ROOT / "tests"))
    from oracle import binding as orc
    from util import random_graph
    for tag, (nvtx, deg, gseed, ntrain, n, seed) in SAMPLER_CASES.items():
        rp, ci = random_graph(nvtx, deg, seed=gseed, power_law=True)
        masks = np.zeros(nvtx, np.uint8)
        masks[:ntrain] = 1
        kept, srp, sci = orc.ref_sample_subgraph(rp, ci, masks, n, seed)
        mrp, mci = orc.ref_masked_graph(rp, ci, masks)
        np.savez_compressed(HERE / f"sampler_{tag}.npz", kept=kept, sub_rowptr=srp, sub_colidx=sci,
                            masked_rowptr=mrp, masked_colidx_crc=np.uint32(__import__("zlib").crc32(mci.tobytes())),
                            params=np.array([nvtx, deg, gseed, ntrain, n, seed]))
    # vertex-range partitioner: PartitionedGraph::edgecut_induced_partition1D on a seeded graph
    for parts in PARTITION_CASES:
        rp, ci = random_graph(*PARTITION_GRAPH[:2], seed=PARTITION_GRAPH[2], power_law=True)
        ref_parts = orc.ref_partition(rp, ci, parts)
        blob = {}
        for i, r in enumerate(ref_parts):
            blob[f"range{i}"] = np.array([r["begin"], r["end"]], np.int64)
            blob[f"idx_map{i}"] = r["idx_map"]
            blob[f"rowptr{i}"] = r["rowptr"]
            blob[f"colidx{i}"] = r["colidx"]
        np.savez_compressed(HERE / f"partition_p{parts}.npz", graph=np.array(PARTITION_GRAPH), **blob)
    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "g.cpp"
        src.write_text(GLOROT_CPP)
        exe = Path(td) / "g"
        subprocess.run(["g++", "-O2", "-std=c++11", str(src), "-o", str(exe)], check=True)
        for dx, dy, seed in GLOROT_CASES:
            raw = subprocess.run([str(exe), str(dx), str(dy), str(seed)], check=True, capture_output=True).stdout
            np.save(HERE / f"glorot_{dx}x{dy}_s{seed}.npy", np.frombuffer(raw, np.float32).reshape(dx, dy))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
