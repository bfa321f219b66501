"""profiles/hbm_traffic.json from the PMC summaries of scripts/profile_round.sh (FETCH_SIZE / WRITE_SIZE passes).

    python scripts/make_hbm_traffic.py gpurun_out/prof_r02 <commit the profile was taken at> [gpurun_out/prof_shard_r04]

Counters are in KB (1024 B).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
coalesced read -- re-calibrated for THIS access pattern (64 lanes x 8 B gathers of 512-B rows) with
scripts/calibrate_fetch.py on a permutation graph whose byte count is known.  FETCH_SIZE counts L2 -> fabric requests:
Infinity-Cache hits are INCLUDED, so the figures are an upper bound on the HBM share."""
import hashlib
import json
import sys
from pathlib import Path

src, commit = Path(sys.argv[1]), sys.argv[2]
ROOT = Path(__file__).resolve().parent.parent


def blob_hashes(rels):
    """path -> `git hash-object`: bench.py recomputes these and drops `traffic` when a kernel source has changed"""
    out = {}
    for rel in rels:
        data = (ROOT / rel).read_bytes()
        out[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


pmc = json.loads((src / "gcn_pmc_summary.json").read_text())
cal = json.loads((src / "calib_pmc.json").read_text())
NV, D = 2_449_029, 128
known_read, known_write = NV * 4 * D + NV * 16, NV * 4 * D


def find(d, counter, frag):
    for k, v in d[counter].items():
        if frag in k:
            return v["mean"]
    raise SystemExit(f"{frag} not in {counter}")


cf, cw = find(cal, "FETCH_SIZE", "spmm_w64_kernel"), find(cal, "WRITE_SIZE", "spmm_w64_kernel")
ratio = cf * 1024 / known_read
out = {
    "workload": "bench.py N=1: products-shaped graph nv=2449029 ne=125915443 D=128, GCN layer fwd+bwd",
    "commit": commit,
    "method": __doc__.split("Counters are", 1)[1].strip().replace("\n", " "),
    "calibration": {"known_read_bytes": known_read, "FETCH_SIZE_KB": cf, "ratio_counter_to_known": ratio,
                    "known_write_bytes": known_write, "WRITE_SIZE_KB": cw, "write_ratio": cw * 1024 / known_write},
    "fetch_correction": 2.0,
    "sources": blob_hashes(["graphaibench_amd/csrc/spmm.hip", "graphaibench_amd/csrc/spmm_kernels.h", "graphaibench_amd/csrc/spmm_core.h",
                            "graphaibench_amd/csrc/common.h", "graphaibench_amd/csrc/sgemm.hip"]),
}
for name, frag in (("spmm_gemm_kernel", "spmm_gemm_kernel"), ("spmm_heavy_kernel", "spmm_heavy_kernel"),
                   ("sgemm_tn_reg_kernel_masked", "sgemm_tn_reg_kernel<true")):
    f, w = find(pmc, "FETCH_SIZE", frag), find(pmc, "WRITE_SIZE", frag)
    out[name] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w}
    out[f"{name}_bytes_per_launch"] = f * 1024 * 2.0 + w * 1024
# the GAT bench line (bench.py --workload gat-reddit): its one-sweep kernels.  The 64-B-per-lane-group gathers of 256-B
# rows use the same x2 correction (the guide's rule for wide reads; not separately calibrated)
gp = src / "gat_pmc_summary.json"
if gp.exists():
    gat = json.loads(gp.read_text())
    rec = {"workload": "bench.py --workload gat-reddit: reddit-shaped graph, 8-head GAT layer 64->64 fwd+bwd", "commit": commit,
           "sources": blob_hashes(["graphaibench_amd/csrc/gat.hip", "graphaibench_amd/csrc/common.h"])}
    for key, frag in (("gat_fwd_fused", "gat_fwd_fused_chunk_kernel"), ("gat_bwd_fused", "gat_bwd_fused_chunk_kernel")):
        try:
            f, w = find(gat, "FETCH_SIZE", frag), find(gat, "WRITE_SIZE", frag)
        except SystemExit:
            continue
        rec[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w}
        rec[f"{key}_bytes_per_launch"] = f * 1024 * 2.0 + w * 1024
    out["gat_reddit"] = rec
# the planted-locality graph of bench.py's `roofline.planted_locality` leg: FETCH_SIZE of the fused kernel from
# `rocprofv3 --pmc FETCH_SIZE -- python3 scripts/locality_study.py --kernel fused` (+ --parse), natural order; the stored
# rows (2 N x 4D per launch) are added as written bytes
lp = src / "locality_fused_fetch.jsonl"
if lp.exists():
    nat = [json.loads(l) for l in lp.read_text().splitlines() if l.startswith("{") and '"FETCH_SIZE_KB"' in l and '"natural"' in l]
    if nat:
        fkb = nat[-1]["FETCH_SIZE_KB"]
        out["planted_locality"] = {"workload": "scripts/locality_study.py --kernel fused, natural order (= synth.planted_locality)",
                                   "commit": commit, "sources": out["sources"], "spmm_gemm_kernel": {"FETCH_SIZE_KB": fkb},
                                   "spmm_gemm_kernel_bytes_per_launch": fkb * 1024 * 2.0 + 2 * NV * 4 * D}
# the N > 1 bench record's dominant kernel on the uniform generator: the owned-column pass of the split (spmm_w64_kernel over
# rank 0's own rows; the same 113.6 M own-column edges at every N), from scripts/profile_shard.sh's PMC passes
sp = Path(sys.argv[3]) / "shard_products_uniform_pmc_summary.json" if len(sys.argv) > 3 else None
if sp is not None and sp.exists():
    sh = json.loads(sp.read_text())
    f, w = find(sh, "FETCH_SIZE", "spmm_w64_kernel"), find(sh, "WRITE_SIZE", "spmm_w64_kernel")
    out["partitioned_products_uniform"] = {
        "workload": "scripts/papers_shard.py --shape ogbn-products --cut 0.1 --boundary uniform --mode auto: rank 0 of 8, split mode",
        "commit": commit,
        "sources": blob_hashes(["graphaibench_amd/csrc/spmm.hip", "graphaibench_amd/csrc/spmm_kernels.h", "graphaibench_amd/csrc/spmm_core.h",
                                "graphaibench_amd/csrc/spmm_part.hip", "graphaibench_amd/csrc/common.h"]),
        "spmm_w64_kernel": {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w},
        "spmm_w64_kernel_owned_pass_bytes_per_launch": f * 1024 * 2.0 + w * 1024}
# round 5: the N > 1 HEADLINE's shard (the N = 1 bench graph in 8 vertex ranges): its dominant kernel is the halo-column half --
# spmm_gemm_kernel with the accumulate form (it continues the owned-column partial sums); the launches of that kernel in the run
# are the two halves of a step, so the mean over launches is the per-launch figure
sp2 = Path(sys.argv[3]) / "shard_products_strong_pmc_summary.json" if len(sys.argv) > 3 else None
if sp2 is not None and sp2.exists():
    sh = json.loads(sp2.read_text())
    f, w = find(sh, "FETCH_SIZE", "spmm_gemm_kernel"), find(sh, "WRITE_SIZE", "spmm_gemm_kernel")
    out["partitioned_products_strong"] = {
        "workload": "scripts/papers_shard.py --strong --mode auto: rank 0 of 8 of the N = 1 bench graph (random order), split mode",
        "commit": commit,
        "sources": blob_hashes(["graphaibench_amd/csrc/spmm.hip", "graphaibench_amd/csrc/spmm_kernels.h", "graphaibench_amd/csrc/spmm_core.h",
                                "graphaibench_amd/csrc/spmm_part.hip", "graphaibench_amd/csrc/common.h"]),
        "spmm_gemm_kernel": {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w},
        "spmm_gemm_kernel_halo_half_bytes_per_launch": f * 1024 * 2.0 + w * 1024}
out["note"] = ("FETCH_SIZE counts L2 -> fabric requests; Infinity-Cache hits are included (MI355X_MICROARCH.md), so "
               "these are upper bounds on the HBM bytes.")
Path("profiles/hbm_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps({k: v for k, v in out.items() if k.endswith("per_launch")}, indent=1), "calibration ratio", ratio)
