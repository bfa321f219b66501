"""CPU suite: the GraphSAINT-style frontier sampler of the host mirror (include/gnn/sampler.h,
SURVEY 8f rank 4) -- host-only code, no GPU: sampled sets stay inside the training set, the
subgraph is exactly the one the full graph induces on the set (re-indexed, rows sorted), sampling
is seeded-deterministic and degree-biased."""
import numpy as np
import pytest

from graphaibench_amd import layers as L
from util import random_graph


def _induced(rp, ci, ids):
    pos = {int(v): k for k, v in enumerate(ids)}
    rows = []
    for v in ids:
        rows.append([pos[int(c)] for c in ci[rp[v]:rp[v + 1]] if int(c) in pos])
    return rows


@pytest.mark.parametrize("n,m", [(400, 50), (300, 300), (1000, 200)])
def test_sampler_subgraph_is_induced_and_inside_training_set(n, m):
    rp, ci = random_graph(4000, 10, seed=7, power_law=True)
    nv = len(rp) - 1
    masks = np.zeros(nv, np.uint8)
    masks[:2500] = 1
    srp, sci, ids = L.sample_subgraph(rp, ci, masks, n, m, seed=3)
    assert 0 < len(ids) <= n and np.all(np.diff(ids.astype(np.int64)) > 0)
    assert masks[ids].all(), "sampled vertices must be training vertices"
    if n > m:
        assert len(ids) > m  # the frontier walk added vertices beyond the initial frontier
    want = _induced(rp, ci, ids)
    for k in range(len(ids)):
        got = list(sci[srp[k]:srp[k + 1]])
        assert got == want[k] and got == sorted(got)
    # deterministic for a seed, different for another
    srp2, sci2, ids2 = L.sample_subgraph(rp, ci, masks, n, m, seed=3)
    assert np.array_equal(ids, ids2) and np.array_equal(sci, sci2)
    _, _, ids3 = L.sample_subgraph(rp, ci, masks, n, m, seed=4)
    assert not np.array_equal(ids, ids3)


def test_sampler_prefers_high_degree_vertices():
    rp, ci = random_graph(6000, 12, seed=11, power_law=True)
    nv = len(rp) - 1
    masks = np.ones(nv, np.uint8)
    deg = np.diff(rp)
    picked = np.zeros(nv)
    for seed in range(20):
        _, _, ids = L.sample_subgraph(rp, ci, masks, 600, 100, seed=seed)
        picked[ids] += 1
    # reached-by-walk vertices are neighbours of frontier vertices: mean degree well above average
    assert (deg * picked).sum() / picked.sum() > 1.5 * deg.mean()


def test_sampler_handles_isolated_training_graph():
    rp = np.zeros(101, np.int64)
    ci = np.zeros(0, np.uint32)
    srp, sci, ids = L.sample_subgraph(rp, ci, np.ones(100, np.uint8), 50, 10, seed=1)
    assert 0 < len(ids) <= 10 and srp[-1] == 0
