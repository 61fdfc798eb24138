"""Condenses rocprofv3 CSV output (kernel stats + PMC passes, written by tools/profile_bench.sh) into a small text
summary.  Launches of one kernel are grouped by grid size: bench.py launches the same PairHMM kernel on the resident
1 M-test-case batch (the launch `roofline` is quoted on), on 65 536-test-case queue batches and on ragged classes."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
KEEP = ("pairhmm", "radix", "k_build", "k_find", "k_mark", "k_indicator", "k_expand", "k_order", "k_hist", "k_sw")


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


print("== kernel stats (rocprofv3 --kernel-trace --stats), top rows ==")
for f in find("stats", "*kernel_stats.csv"):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            print(",".join([short(row[0])] + row[1:]))
            if i > 14:
                break
print()
print("== per-dispatch durations from the kernel trace, grouped by (kernel, grid size in workgroups) ==")
for f in find("stats", "*kernel_trace.csv"):
    d = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            wg = int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)
            d[(short(row["Kernel_Name"]), wg)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for (k, wg), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if not any(x in k for x in KEEP) or sum(v) < 1.0:
            continue
        print(f"{k[:58]:58s} workgroups={wg:8d} n={len(v):4d} avg_ms={sum(v)/len(v):9.4f} min_ms={min(v):9.4f} max_ms={max(v):9.4f}")
print()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):   # written by tools/profile_bench.sh
    for f in find(sub, "*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                wg = int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0) // max(int(row.get("Workgroup_Size", row.get("Workgroup_Size_X", 1)) or 1), 1)
                agg[(short(row["Kernel_Name"]), wg)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"== {sub}: per-launch counter averages, grouped by (kernel, workgroups) ==")
        for (k, wg), cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            if not any(x in k for x in KEEP):
                continue
            for c, v in cs.items():
                if sum(v) / len(v) < 1000:
                    continue
                print(f"{k[:58]:58s} workgroups={wg:8d} {c:16s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
        print()
