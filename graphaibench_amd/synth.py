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
    cdf = torch.cumsum(w, 0)
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
