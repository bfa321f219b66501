"""Seeded synthetic graphs shaped like the datasets BASELINE.json names.

No dataset ships with the reference beyond cora/citeseer topology, and there is no network, so the
bench regenerates graphs from seeds on the device it runs on: a Chung-Lu power-law graph
(expected degree of vertex v proportional to a weight w_v), symmetric, duplicate-free, without
self-loops, rows sorted -- the invariants LearningGraph::add_selfloop and symmetric_csr_transpose
require (lgraph.h:185-218, math_functions.cpp:46-74).  Vertex ids are randomly permuted, i.e. the
vertex order carries NO locality (the conservative case for the gather).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

# name -> (num_vertices, target nnz (directed, no self-loops), mean degree, max degree, F, C)
SHAPES = {
    "ogbn-products": (2_449_029, 123_718_280, 17_000, 100, 47),
    "reddit": (232_965, 114_615_892, 21_657, 602, 41),
    # symmetrised edge count as SURVEY.md 8 quotes it; the maximum degree is an assumption (not in the public stats)
    "ogbn-papers100M": (111_059_956, 3_231_371_744, 30_000, 128, 172),
    # one of 8 vertex ranges of it (BASELINE config 5: "vertex-partitioned across 8 x MI355X"): block_rows' per-range shape
    "ogbn-papers100M/8": (13_882_495, 403_921_468, 30_000, 128, 172),
    "cora": (2_708, 10_556, 168, 1433, 7),
    "tiny": (20_000, 400_000, 800, 32, 7),
}


@dataclass
class SynthGraph:
    name: str
    nv: int
    rowptr: torch.Tensor  # int64 [nv+1]
    colidx: torch.Tensor  # int32 [ne]  (bit pattern of uint32)
    seed: int

    @property
    def ne(self) -> int:
        return int(self.colidx.numel())


def _cumsum_f64(w: torch.Tensor) -> torch.Tensor:
    """running sum of float64 weights, THE SAME BITS on every call: a device scan (decoupled look-back) associates the partial sums
    by the timing of its tiles, so two generations of the same range -- a rank's own and the one rank 0 makes for the oracle's
    global run, or the two owners of a cross block -- could differ in the last place of the cdf and with it in a handful of
    sampled edges (found by the N > 1 parity leg at 4 busy ranks on one device: 1 to 41 rows per rank off by up to 3 % of the
    largest value, in some runs only).  Sequential on the host: 14 M entries take 40 ms."""
    import numpy as np

    return torch.from_numpy(np.cumsum(w.detach().cpu().numpy().astype(np.float64, copy=False))).to(w.device)


def _weights(nv: int, mean_deg: float, max_deg: float, device) -> torch.Tensor:
    """w_r = (r + r0)^-alpha with (alpha fixed, r0 solved) so that max/mean expected degree matches."""
    alpha = 0.72
    r = torch.arange(nv, dtype=torch.float64, device=device)
    target = max_deg / mean_deg
    lo, hi = 1e-3, float(nv)
    for _ in range(60):  # bisection on r0: ratio = nv * r0^-alpha / sum (r + r0)^-alpha, decreasing in r0
        mid = (lo * hi) ** 0.5
        w = (r + mid) ** (-alpha)
        ratio = float(nv * w[0] / w.sum())
        if ratio > target:
            lo = mid
        else:
            hi = mid
    return (r + (lo * hi) ** 0.5) ** (-alpha)


def chung_lu(name: str, nv: int, nnz: int, max_deg: int, seed: int = 42, device="cuda",
             oversample: float = 1.0) -> SynthGraph:
    """symmetric CSR with ~nnz stored entries (nnz/2 undirected edges before duplicate removal)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    m = int(nnz // 2 * oversample)
    w = _weights(nv, nnz / nv, max_deg, device)
    perm = torch.randperm(nv, generator=gen, device=device)
    cdf = _cumsum_f64(w)
    cdf = cdf / cdf[-1]
    chunks = []
    step = 1 << 24
    for s in range(0, m, step):
        k = min(step, m - s)
        u = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, generator=gen, device=device))
        v = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, generator=gen, device=device))
        u = perm[u.clamp_(max=nv - 1)]
        v = perm[v.clamp_(max=nv - 1)]
        keep = u != v
        u, v = u[keep], v[keep]
        chunks.append(torch.minimum(u, v) * nv + torch.maximum(u, v))
    key = torch.unique(torch.cat(chunks))  # sorted, duplicate-free undirected edges
    del chunks
    a = key // nv
    b = key - a * nv
    del key
    key2 = torch.cat([a * nv + b, b * nv + a])
    del a, b
    key2, _ = torch.sort(key2)
    rows = key2 // nv
    cols = (key2 - rows * nv).to(torch.int32)
    del key2
    counts = torch.bincount(rows, minlength=nv)
    rowptr = torch.zeros(nv + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return SynthGraph(name, nv, rowptr, cols, seed)


def make(name: str, seed: int = 42, device="cuda", scale: float = 1.0) -> SynthGraph:
    """scale < 1 shrinks vertices and edges together (parity tests / CPU baseline samples)."""
    nv, nnz, max_deg, _, _ = SHAPES[name]
    nv_s = max(int(nv * scale), 16)
    nnz_s = max(int(nnz * scale), 32)
    max_s = max(min(max_deg, nv_s // 4), 4)
    return chung_lu(name, nv_s, nnz_s, max_s, seed, device)


# ---- a graph whose NUMBERING carries locality ------------------------------------------
def planted_locality(name: str = "ogbn-products", block: int = 16384, cut: float = 0.1, seed: int = 42, device="cuda",
                     relabel=None, selfloops: bool = True) -> SynthGraph:
    """The named shape with PLANTED communities: vertices in blocks of `block` consecutive ids, a share 1 - cut of every
    vertex's edges stays inside its block (endpoints by the same power-law weights, hubs spread over the blocks) -- what a
    dataset with community structure in its numbering looks like, where make() permutes the ids at random.  relabel: an
    optional [nv] int64 permutation applied to both endpoints (the same graph under another numbering)."""
    device = torch.device(device)
    nv, nnz, max_deg, _, _ = SHAPES[name]
    max_deg = min(max_deg, nv // 4)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    w = _weights(nv, nnz / nv, max_deg, device)
    w = w[torch.randperm(nv, generator=gen, device=device)]
    cdf = _cumsum_f64(w)
    cdf = cdf / cdf[-1]
    m = nnz // 2
    keys = []
    step = 1 << 24
    for s0 in range(0, m, step):
        k = min(step, m - s0)
        u = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, generator=gen, device=device)).clamp_(max=nv - 1)
        b0 = (u // block) * block
        b1 = torch.clamp(b0 + block, max=nv)
        lo = torch.where(b0 > 0, cdf[(b0 - 1).clamp_(min=0)], torch.zeros_like(cdf[b0]))
        hi = cdf[b1 - 1]
        r = torch.rand(k, dtype=torch.float64, generator=gen, device=device)
        local = torch.rand(k, dtype=torch.float64, generator=gen, device=device) >= cut
        v = torch.searchsorted(cdf, torch.where(local, lo + r * (hi - lo), r)).clamp_(max=nv - 1)
        keep = u != v
        u, v = u[keep], v[keep]
        keys.append(torch.minimum(u, v) * nv + torch.maximum(u, v))
    key = torch.unique(torch.cat(keys))
    del keys
    a, b = key // nv, key % nv
    if relabel is not None:
        a, b = relabel[a], relabel[b]
    parts = [a * nv + b, b * nv + a]
    if selfloops:
        i = torch.arange(nv, dtype=torch.int64, device=device)
        parts.append(i * nv + i)
    key2, _ = torch.sort(torch.cat(parts))
    rows = key2 // nv
    cols = (key2 - rows * nv).to(torch.int32)
    rowptr = torch.zeros(nv + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.bincount(rows, minlength=nv), 0, out=rowptr[1:])
    return SynthGraph(name + f" (planted locality: blocks of {block}, cut {cut})", nv, rowptr, cols, seed)


# ---- block generator for the multi-GPU bench -------------------------------------------------------
@dataclass
class BlockRows:
    """one rank's rows of a global graph made of `world` products-shaped vertex ranges"""
    rowptr: torch.Tensor         # int64 [n_local+1]
    colidx_global: torch.Tensor  # int64 [ne_local], GLOBAL column ids, rows sorted
    n_global: int
    n_local: int


def _part_sampler(nv_p: int, nnz_p: int, max_deg: int, seed: int, p: int, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 1009 + p)
    w = _weights(nv_p, nnz_p / nv_p, max_deg, device)
    perm = torch.randperm(nv_p, generator=gen, device=device)
    cdf = _cumsum_f64(w)
    return cdf / cdf[-1], perm


def _sample(cdf, perm, k: int, gen, device):
    out = []
    step = 1 << 24
    for s in range(0, k, step):
        r = torch.rand(min(step, k - s), dtype=torch.float64, generator=gen, device=device)
        out.append(perm[torch.searchsorted(cdf, r).clamp_(max=perm.numel() - 1)])
    return torch.cat(out) if out else torch.empty(0, dtype=torch.int64, device=device)


def _band_sampler(cdf, perm, lo: int, hi: int):
    """the power-law weights of one range restricted to the local ids [lo, hi): (cdf over those ids, lo)"""
    w = torch.empty_like(cdf)
    w[0] = cdf[0]
    w[1:] = cdf[1:] - cdf[:-1]
    by_id = torch.empty_like(w)
    by_id[perm] = w  # perm[r] = the id that carries the r-th largest weight
    c = _cumsum_f64(by_id[lo:hi])
    return c / c[-1], lo


def _sample_band(band, k: int, gen, device):
    cdf, lo = band
    out = []
    step = 1 << 24
    for s in range(0, k, step):
        r = torch.rand(min(step, k - s), dtype=torch.float64, generator=gen, device=device)
        out.append(torch.searchsorted(cdf, r).clamp_(max=cdf.numel() - 1) + lo)
    return torch.cat(out) if out else torch.empty(0, dtype=torch.int64, device=device)


def block_rows(name: str, rank: int, world: int, seed: int = 42, cut_fraction: float = 0.1, device="cuda",
               scale: float = 1.0, selfloops: bool = False, boundary: str = "uniform", band: float = 0.2) -> BlockRows:
    """Rows [rank*nv_p, (rank+1)*nv_p) of a symmetric global graph with world*nv_p vertices.  Each
    vertex range is a Chung-Lu graph of the named shape; a fraction `cut_fraction` of a range's
    stored entries point into the other ranges (spread evenly, endpoints drawn by the same
    power-law weights).  Block (p, q) is generated from a seed shared by p and q, so both owners
    see the same undirected edges.  cut_fraction models the partitioner: ~(world-1)/world is a
    random vertex order, small values a locality-preserving (METIS-like) order.
    boundary = "uniform": a cut edge may end at ANY vertex of the two ranges -- with 50 edges per vertex nearly every row
    then has a neighbour elsewhere, which no partitioner produces.  boundary = "clustered": the cut edges land on a
    BOUNDARY BAND, the first `band` share of every range's ids, cut into world - 1 slices, one facing each peer (what a
    partition by METIS or a breadth-first order looks like: most vertices are interior, the cut runs through a surface):
    the same number of cut edges, on band x band endpoints drawn by the same weights restricted to the slices."""
    device = torch.device(device)
    nv, nnz, max_deg, _, _ = SHAPES[name]
    nv_p = max(int(nv * scale), 16)
    nnz_p = max(int(nnz * scale), 32)
    max_p = max(min(max_deg, nv_p // 4), 4)
    n_global = nv_p * world
    lo = rank * nv_p
    cut = cut_fraction if world > 1 else 0.0
    samplers = {}

    def sampler(p):
        if p not in samplers:
            samplers[p] = _part_sampler(nv_p, nnz_p, max_p, seed, p, device)
        return samplers[p]

    keys = []
    # intra-range block
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 7919 + rank * 65537 + rank)
    cdf, perm = sampler(rank)
    m_in = int((1.0 - cut) * nnz_p / 2)
    u = _sample(cdf, perm, m_in, gen, device)
    v = _sample(cdf, perm, m_in, gen, device)
    keep = u != v
    u, v = u[keep], v[keep]
    keys.append(u * n_global + (v + lo))
    keys.append(v * n_global + (u + lo))
    del u, v
    # cross blocks
    if world > 1 and cut > 0:
        m_x = int(cut * nnz_p / (world - 1))
        for q in range(world):
            if q == rank:
                continue
            a_p, b_p = min(rank, q), max(rank, q)
            gen = torch.Generator(device=device)
            gen.manual_seed(seed * 7919 + a_p * 65537 + b_p)
            ca, pa = sampler(a_p)
            cb, pb = sampler(b_p)
            if boundary == "clustered":
                nb = max(int(band * nv_p), world - 1)

                def face(owner, peer):  # the slice of owner's band that faces peer
                    k = peer if peer < owner else peer - 1
                    return (k * nb) // (world - 1), max(((k + 1) * nb) // (world - 1), (k * nb) // (world - 1) + 1)

                a = _sample_band(_band_sampler(ca, pa, *face(a_p, b_p)), m_x, gen, device)
                b = _sample_band(_band_sampler(cb, pb, *face(b_p, a_p)), m_x, gen, device)
            else:
                assert boundary == "uniform", boundary
                a = _sample(ca, pa, m_x, gen, device)  # local id in range a_p
                b = _sample(cb, pb, m_x, gen, device)  # local id in range b_p
            if rank == a_p:
                keys.append(a * n_global + (b + b_p * nv_p))
            else:
                keys.append(b * n_global + (a + a_p * nv_p))
            del a, b
            if q != rank:
                samplers.pop(q, None)
    if selfloops:
        i = torch.arange(nv_p, dtype=torch.int64, device=device)
        keys.append(i * n_global + (i + lo))
    key = torch.unique(torch.cat(keys))  # sorted by (row, col), duplicate-free
    del keys
    rows = key // n_global
    cols = key - rows * n_global
    counts = torch.bincount(rows, minlength=nv_p)
    rowptr = torch.zeros(nv_p + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return BlockRows(rowptr, cols, n_global, nv_p)


def write_dataset(name: str, root, scale: float = 1.0, device="cuda", seed: int = 42):
    """The named shape as a dataset directory in the reference's binary format (src/gnn/reader.cpp:414-457):
    root/name/graph.{meta.txt,vertex.bin,edge.bin,vlabel.bin,feats.bin} -- what the trainer CLI reads.  Topology: make();
    features: class mean + noise (training has something to learn), seed 43; labels uniform, seed 44; masks = the contiguous
    8 % / 2 % / 90 % ranges of the meta file.  Returns dict(nv, ne, F, C, train_begin, train_end, dir)."""
    from pathlib import Path

    import numpy as np

    nv0, nnz0, maxdeg, F, C = SHAPES[name]
    sg = make(name, seed=seed, device=device, scale=scale)
    d = Path(root) / name
    d.mkdir(parents=True, exist_ok=True)
    sg.rowptr.cpu().numpy().astype(np.int64).tofile(d / "graph.vertex.bin")
    sg.colidx.cpu().numpy().view(np.uint32).tofile(d / "graph.edge.bin")
    nv, ne = sg.nv, sg.ne
    g = torch.Generator(device="cpu").manual_seed(44)
    labels = torch.randint(0, C, (nv,), generator=g, dtype=torch.int64)
    labels.numpy().astype(np.uint8).tofile(d / "graph.vlabel.bin")
    gen = torch.Generator(device=device).manual_seed(43)
    centers = torch.randn(C, F, device=device, generator=gen)
    with open(d / "graph.feats.bin", "wb") as f:
        step = 1 << 18
        for s in range(0, nv, step):
            lab = labels[s:s + step].to(device)
            x = centers[lab] * 0.5 + torch.randn(lab.numel(), F, device=device, generator=gen)
            f.write(x.float().cpu().numpy().tobytes())
    max_degree = int((sg.rowptr[1:] - sg.rowptr[:-1]).max())
    tr, va = int(0.08 * nv), int(0.10 * nv)
    meta = [nv, ne, 4, 8, 1, 2, max_degree, F, C, 0, 0, tr, tr, tr, va, va - tr, va, nv, nv - va]
    (d / "graph.meta.txt").write_text("\n".join(str(v) for v in meta) + "\n")
    return dict(nv=nv, ne=ne, F=F, C=C, train_begin=0, train_end=tr, max_degree=max_degree, dir=str(d))


def write_cora_dataset(root, golden_dir, feat_len: int = 1433, seed: int = 0):
    """BASELINE config 2's dataset: the reference's own cora topology / labels / split (golden_dir: byte copies of the files the
    reference ships under inputs/cora -- data, kept under tests/golden/cora) with seeded sparse-binary, row-normalised features
    like the real ones (the shipped dataset has none).  Returns the same dict as write_dataset."""
    import shutil
    from pathlib import Path

    import numpy as np

    gold, d = Path(golden_dir), Path(root) / "cora"
    d.mkdir(parents=True, exist_ok=True)
    for f in ("graph.vertex.bin", "graph.edge.bin", "graph.vlabel.bin", "graph.meta.txt"):
        shutil.copyfile(gold / f, d / f)
    meta = (d / "graph.meta.txt").read_text().split()
    meta[7] = str(feat_len)
    (d / "graph.meta.txt").write_text("\n".join(meta) + "\n")
    rng = np.random.default_rng(seed)
    labels = np.fromfile(d / "graph.vlabel.bin", np.uint8).astype(int)
    nv, ncls = len(labels), int(labels.max()) + 1
    proto = rng.random((ncls, feat_len)) < 0.03          # class "vocabularies"
    x = ((rng.random((nv, feat_len)) < 0.008) | (proto[labels] & (rng.random((nv, feat_len)) < 0.3))).astype(np.float32)
    x /= np.maximum(x.sum(1, keepdims=True), 1.0)
    x.tofile(d / "graph.feats.bin")
    return dict(nv=nv, ne=int(meta[1]), F=feat_len, C=ncls, train_begin=int(meta[10]), train_end=int(meta[11]),
                max_degree=int(meta[6]), dir=str(d))
