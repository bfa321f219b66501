"""Summarise rocprofv3 output directories into small JSON/CSV files for profiles/.

    python scripts/summarize_rocprof.py stats <dir> <out.csv>        # *_kernel_stats.csv, our kernels only
    python scripts/summarize_rocprof.py pmc <out.json> NAME=<dir> ...  # mean counter per kernel per pass
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
