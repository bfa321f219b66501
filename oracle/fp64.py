"""fp64 evaluations of the path's formulas on the device (torch), for ONE purpose: to measure how far the fp32 oracle
itself is from the exact result where a comparison at the default floor is not meaningful -- sums of 10^6..10^9 fp32
terms, differences of nearly cancelling dot products, the jump of leaky_relu' at zero.  A tolerance in tests/ or in
bench.py's `parity` that is looser than 1e-4 at the default floor is always "twice the ORACLE's own measured distance
from fp64" (never the GPU's), computed here, in the run that uses it.

TEST INFRASTRUCTURE ONLY, like everything under oracle/: imported by tests/, bench.py's parity leg and scripts/."""
from __future__ import annotations

import numpy as np
import torch


def _rows_of(rowptr: torch.Tensor) -> torch.Tensor:
    n = rowptr.numel() - 1
    return torch.repeat_interleave(torch.arange(n, device=rowptr.device), rowptr[1:] - rowptr[:-1])


def gat_alpha_grads_fp64(rowptr, colidx, hfeat, alpha_l, alpha_r, g_act, heads: int, temp_fp32=None, signs=None):
    """alpha gradients of GAT_Aggregator::d_aggregate (gat_aggregator.cpp:99-165) in fp64, head by head.
    rowptr int64 [n+1], colidx [ne] (with self loops), hfeat / g_act fp32 [n x d] (g_act = the gradient after d_relu),
    alpha_* fp32 [d]: numpy or device tensors.  Returns (alpha_l grad [d] f64 numpy, alpha_r grad [d] f64 numpy, info);
    info counts the leaky-relu sign flips between temp_fp32 ([heads] list of [ne] fp32 score arrays, e.g. the oracle's)
    and the fp64 scores, and bounds what those flips are worth per entry (sum of 0.8 |ds_e| |h| over the flipped edges).
    signs: [ne x heads] uint8 device tensor / array (t_e > 0) to IMPOSE on leaky_relu' (the softmax itself keeps the fp64
    scores: leaky_relu is continuous, only its derivative jumps) -- with an implementation's own signs the result is what
    that implementation's arithmetic should reproduce; info then also counts how many signs differ from fp64's and the
    largest |t| (relative to max |t|) among them."""
    dev = "cuda"
    t_ = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(dev)
    rowptr = t_(rowptr).to(torch.int64)
    col = t_(colidx.astype(np.int64) if isinstance(colidx, np.ndarray) else colidx).to(torch.int64)
    rows = _rows_of(rowptr)
    n = rowptr.numel() - 1
    hf, ga = t_(hfeat), t_(g_act)
    al, ar = t_(alpha_l).double(), t_(alpha_r).double()
    d = hf.shape[1]
    dh = d // heads
    lg, rg = np.empty(d), np.empty(d)
    flips, worth_l, worth_r = 0, np.zeros(d), np.zeros(d)
    imposed_flips, imposed_worst = 0, 0.0
    step = 1 << 24
    for k in range(heads):
        sl = slice(k * dh, (k + 1) * dh)
        hk = hf[:, sl].double()
        s_l, s_r = hk @ al[sl], hk @ ar[sl]
        t = s_l[rows] + s_r[col]
        s = torch.where(t > 0, t, 0.2 * t)
        M = torch.full((n,), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, rows, s, "amax")
        e = torch.exp(s - M[rows])
        S = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, e)
        p = e / S[rows]
        del e, s
        gk = ga[:, sl].double()
        dp = torch.empty_like(p)
        for a in range(0, p.numel(), step):  # the gathered rows chunk by chunk
            dp[a:a + step] = (gk[rows[a:a + step]] * hk[col[a:a + step]]).sum(1)
        rowdot = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, p * dp)
        ds = p * (dp - rowdot[rows])
        del dp
        pos = t > 0
        if signs is not None:
            sg = t_(signs)[:, k].to(torch.bool)
            diff = sg != pos
            imposed_flips += int(diff.sum().item())
            if diff.any():
                imposed_worst = max(imposed_worst, float((t[diff].abs().max() / t.abs().max()).item()))
            pos = sg
        ge = ds * torch.where(pos, 1.0, 0.2)
        cs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, col, ge)
        rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, ge)
        lg[sl], rg[sl] = (rs @ hk).cpu().numpy(), (cs @ hk).cpu().numpy()
        if temp_fp32 is not None:
            fl = (t_(temp_fp32[k]) > 0) != (t > 0)
            flips += int(fl.sum().item())
            if fl.any():
                w = 0.8 * ds[fl].abs()
                worth_r[sl] += (w[:, None] * hk[col[fl]].abs()).sum(0).cpu().numpy()
                worth_l[sl] += (w[:, None] * hk[rows[fl]].abs()).sum(0).cpu().numpy()
            del fl
        del t, p, ds, ge, cs, rs
        torch.cuda.empty_cache()
    return lg, rg, {"sign_flips": flips, "flips_worth_l": worth_l, "flips_worth_r": worth_r,
                    "imposed_sign_flips_vs_fp64": imposed_flips, "imposed_flips_max_abs_t_over_scale": imposed_worst}


def inf_dist(a, b) -> float:
    """max|a - b| / max|b| in fp64 (numpy)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def sage_grad_out_rows_fp64(rowptr, colidx, masked_grad, W_neigh, W_self, sample_rows):
    """rows `sample_rows` of SAGE_layer::backward's input gradient (sage_layer.cpp:44-50, level > 0, din <= dout) in
    fp64:  grad_out[r] = sum_{e in row r} w_e (g W_neigh^T)[col_e] + (g W_self^T)[r],  w_e = float(1.0 / float(deg(col_e)))
    -- the SAME fp32 edge weights the fp32 paths use (sage_aggregator.cpp:44), so that what is measured is the
    summation, not the weights.  masked_grad = the gradient after d_relu, fp32 [n x dout].  Returns f64 device tensor."""
    dev = "cuda"
    t_ = lambda a: (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(dev)
    rowptr = t_(rowptr).to(torch.int64)
    col = t_(colidx.astype(np.int64) if isinstance(colidx, np.ndarray) else colidx).to(torch.int64)
    g = t_(masked_grad)
    Wn, Ws = t_(W_neigh).double(), t_(W_self).double()
    n = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
    w_v = (1.0 / deg.double()).to(torch.float32).double()  # 1.0 / float(deg), narrowed (sage_aggregator.cpp:44)
    T = torch.empty(n, Wn.shape[0], dtype=torch.float64, device=dev)
    step = 1 << 18
    for a in range(0, n, step):
        T[a:a + step] = g[a:a + step].double() @ Wn.t()
    sr = t_(sample_rows).to(torch.int64)
    out = g[sr].double() @ Ws.t()
    for a in range(0, sr.numel(), 8192):  # edges of a slice of the sampled rows
        r = sr[a:a + 8192]
        cnt = rowptr[r + 1] - rowptr[r]
        eidx = torch.repeat_interleave(rowptr[r], cnt) + (torch.arange(int(cnt.sum()), device=dev) -
                                                         torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
        c = col[eidx]
        local = torch.repeat_interleave(torch.arange(r.numel(), device=dev), cnt)
        out[a:a + 8192].index_add_(0, local, T[c] * w_v[c][:, None])
    return out
