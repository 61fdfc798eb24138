#!/bin/bash
# SQ counters of the packed and scalar PairHMM kernels side by side (two rocprofv3 --pmc passes of tools/dev_pk_one.py)
set -o pipefail
OUT=${1:-$PWD/gpurun_out/prof_pk}
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/p1" -o pk --output-format csv -- python3 "$REPO/tools/dev_pk_one.py" > "$OUT/p1.log" 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA -d "$OUT/p2" -o pk --output-format csv -- python3 "$REPO/tools/dev_pk_one.py" > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
rocprofv3 --kernel-trace --stats -d "$OUT/p3" -o pk --output-format csv -- python3 "$REPO/tools/dev_pk_one.py" > "$OUT/p3.log" 2>&1 || echo "pass 3 failed"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p[12]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pairhmm_fwd" not in k: continue
        k = "pk" if "pairhmm_fwd_pk" in k else ("scalar" if "float" in k else "f64")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
print(f"{'counter':24s} {'packed':>16s} {'scalar':>16s}")
for c in names:
    a = agg["pk"].get(c, [0]); b = agg["scalar"].get(c, [0])
    print(f"{c:24s} {sum(a)/len(a):16.6g} {sum(b)/len(b):16.6g}")
for f in glob.glob(out + "/p3/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 6: print(",".join(row)[:200])
PY
