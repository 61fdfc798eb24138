"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats", "*kernel_stats.csv"):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            print(",".join(row))
            if i > 12:
                break
print()
print("== per-dispatch durations of the PairHMM fp32 kernel (kernel trace) ==")
for f in find("stats", "*kernel_trace.csv"):
    d = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = v[len(v) // 3:] if len(v) > 6 else v
        print(f"{k[:90]:90s} n={len(v):4d} avg_ms={sum(v)/len(v):9.4f} steady_avg_ms={sum(v2)/len(v2):9.4f} min_ms={min(v):9.4f}")
print()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"== {sub}: per-launch counter averages ==")
        for k, cs in agg.items():
            if "pairhmm" not in k and "sortdedup" not in k and "radix" not in k:
                continue
            for c, v in cs.items():
                print(f"{k[:70]:70s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
        print()
