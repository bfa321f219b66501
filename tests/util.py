"""Shared helpers for the tests: seeded graph generators and comparison utilities."""
from __future__ import annotations

import numpy as np


def csr_from_pairs(n: int, src, dst):
    """symmetric, sorted, duplicate-free, self-loop-free CSR (int64 rowptr, uint32 colidx)."""
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    a = np.concatenate([src, dst])
    b = np.concatenate([dst, src])
    key = np.unique(a * n + b)
    rows = key // n
    cols = (key % n).astype(np.uint32)
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, cols


def random_graph(n: int, avg_deg: float, seed: int, power_law: bool = False, hub_deg: int = 0):
    """seeded symmetric graph; power_law=True skews endpoint choice (Chung-Lu style);
    hub_deg > 0 additionally connects vertex 0 to hub_deg random vertices (heavy-row path)."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg / 2)
    if power_law:
        w = (np.arange(1, n + 1, dtype=np.float64)) ** -0.7
        p = w / w.sum()
        src = rng.choice(n, m, p=p)
        dst = rng.choice(n, m, p=p)
    else:
        src = rng.integers(0, n, m)
        dst = rng.integers(0, n, m)
    if hub_deg:
        hub_dst = rng.choice(np.arange(1, n), min(hub_deg, n - 1), replace=False)
        src = np.concatenate([src, np.zeros(len(hub_dst), np.int64)])
        dst = np.concatenate([dst, hub_dst])
    return csr_from_pairs(n, src, dst)


def path_graph(n: int):
    """inputs/gnn-tester topology: a path 0-1-..-(n-1)"""
    s = np.arange(n - 1)
    return csr_from_pairs(n, s, s + 1)


def rel_err(a, b) -> float:
    """max |a-b| / max(|b|_inf, tiny): the norm-wise relative error the north star bounds by 1e-4"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def dense_adj(rowptr, colidx, w=None):
    n = len(rowptr) - 1
    A = np.zeros((n, n), np.float64)
    for i in range(n):
        for e in range(rowptr[i], rowptr[i + 1]):
            A[i, colidx[e]] += 1.0 if w is None else w[e]
    return A
