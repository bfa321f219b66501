"""Summarise rocprofv3 output directories into small JSON/CSV files for profiles/.

    python scripts/summarize_rocprof.py stats <dir> <out.csv>        # *_kernel_stats.csv, our kernels only
    python scripts/summarize_rocprof.py pmc <out.json> NAME=<dir> ...  # mean counter per kernel per pass
    python scripts/summarize_rocprof.py pmcseq <out.json> <group> <skip-regex> NAME=<dir> ...
        # a driver that launches shape after shape through the SAME kernels (scripts/gemm_narrow.py): our dispatches in
        # dispatch order, those matching <skip-regex> dropped (probes, the split-K reduce), cut into consecutive groups of
        # <group> launches = one shape each -> per group the kernel name and the mean of every counter
"""
import csv
import glob
import json
import sys
from collections import defaultdict



def ours(name: str) -> bool:
    return "(anonymous namespace)::" in name and "at::" not in name



def find(d, suffix):
    hits = glob.glob(f"{d}/**/*{suffix}", recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "_kernel_stats.csv"))))
    keep = [r for r in rows if ours(r["Name"])]
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    for r in keep[:12]:
        print(f'{r["Calls"]:>5} x {float(r["AverageNs"]) / 1e6:8.3f} ms  {r["Name"][:110]}')


def pmc(out, pairs):
    res = {}
    for pair in pairs:
        name, d = pair.split("=", 1)
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
            if ours(r["Kernel_Name"]):
                acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for counter, kernels in acc.items():
            res.setdefault(counter, {})
            for k, v in kernels.items():
                res[counter][k] = {"dispatches": len(v), "mean": sum(v) / len(v), "pass": name}
    json.dump(res, open(out, "w"), indent=1)
    print(f"wrote {out}: counters {sorted(res)}")


def pmcseq(out, group, skip, pairs):
    import re

    rx = re.compile(skip)
    res = {}
    for pair in pairs:
        name, d = pair.split("=", 1)
        by_id = {}
        for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
            if ours(r["Kernel_Name"]) and not rx.search(r["Kernel_Name"]):
                e = by_id.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "grid": r.get("Grid_Size"), "ctr": {}})
                e["ctr"][r["Counter_Name"]] = e["ctr"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seq = [by_id[k] for k in sorted(by_id)]
        for gi in range(0, len(seq) // group):
            chunk = seq[gi * group:(gi + 1) * group]
            g = res.setdefault(str(gi), {"kernel": chunk[0]["kernel"], "grid": chunk[0]["grid"], "launches": len(chunk), "counters": {}})
            if any(c["kernel"] != chunk[0]["kernel"] for c in chunk):
                g["mixed_kernels"] = sorted({c["kernel"] for c in chunk})
            for cname in chunk[0]["ctr"]:
                vals = [c["ctr"].get(cname, 0.0) for c in chunk]
                g["counters"][cname] = {"mean": sum(vals) / len(vals), "pass": name}
    json.dump(res, open(out, "w"), indent=1)
    print(f"wrote {out}: {len(res)} groups")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmcseq":
        pmcseq(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5:])
    else:
        pmc(sys.argv[2], sys.argv[3:])
