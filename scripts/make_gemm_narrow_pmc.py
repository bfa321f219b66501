"""profiles/rNN/gemm_narrow_pmc*.json from one run of scripts/profile_gemm_narrow.sh: per shape of scripts/gemm_narrow.py the main
kernel, ms per launch (the unprofiled timing leg), L2 -> fabric traffic and the SQ counters per launch.
    python scripts/make_gemm_narrow_pmc.py gpurun_out/prof_r06 /tmp/after.json   (profiles/r06/gemm_narrow_pmc.json holds "before" and "after")
Counters are in KB (1024 B); MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
coalesced streaming read -- doubled here; WRITE_SIZE is exact.  SQ_WAIT_INST_ANY counts quad-cycles summed over waves."""
import json
import sys

src, out = sys.argv[1], sys.argv[2]
timing = [json.loads(l) for l in open(f"{src}/gemm_narrow_timing.jsonl")]
groups = json.load(open(f"{src}/gemm_narrow_pmc_by_shape.json"))
assert len(groups) == len(timing), (len(groups), len(timing))
shapes = {}
for i, t in enumerate(timing):
    g = groups[str(i)]
    c = {k: v["mean"] for k, v in g["counters"].items()}
    fetch = c.get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e9
    write = c.get("WRITE_SIZE", 0.0) * 1024 / 1e9
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    waves, busy, wait = c.get("SQ_WAVES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAIT_INST_ANY", 0.0)
    rec = {"key": t["key"], "kernel": g["kernel"].replace("void (anonymous namespace)::", "")[:80], "ms": t["ms"], "roof_ms": t["roof_ms"],
           "frac": t["frac"], "alg_gb": t["alg_gb"], "fetch_gb_x2": round(fetch, 3), "write_gb": round(write, 3),
           "traffic_over_alg": round((fetch + write) / t["alg_gb"], 3), "traffic_tb_s": round((fetch + write) / t["ms"], 3),
           "waves": waves, "valu_insts": c.get("SQ_INSTS_VALU"), "wait_inst_any_quadcycles": wait, "busy_cycles": busy,
           "l2_hit": round(hit / (hit + miss), 3) if hit + miss else None}
    if "mixed_kernels" in g:
        rec["mixed_kernels"] = g["mixed_kernels"]
    shapes[t["shape"]] = rec
json.dump({"what": "the dense products of the products-shaped epochs at 2.45 M rows (scripts/gemm_narrow.py), one rocprofv3 --pmc pass per "
                   "counter group (scripts/profile_gemm_narrow.sh): per shape the main kernel, ms per launch (in-stream events, unprofiled "
                   "run of the same call), L2 -> fabric traffic = FETCH_SIZE x 2 (gfx950 half count) + WRITE_SIZE in GB, over the "
                   "algorithmic bytes; SQ counters per launch", "shapes": shapes}, open(out, "w"), indent=1)
print(f"wrote {out}: {len(shapes)} shapes")
