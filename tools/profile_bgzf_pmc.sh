#!/bin/bash
# HBM traffic of the BGZF kernel alone: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes over the bench's bgzf leg
set -o pipefail
OUT=${1:-$PWD/gpurun_out/prof_bgzf}
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/$c" -o bench --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --pairs 65536 --sort-records 0 --sw-pairs 0 --no-queue --no-ragged --no-regions --no-cpu-baseline --bgzf-mb 512 --cli-records 0 --no-mixed > "$OUT/$c.json" 2> "$OUT/$c.err" || echo "$c run failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, statistics as st, sys
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{sys.argv[1]}/{c}/bench_counter_collection.csv")) if "k_bgzf_deflate" in r["Kernel_Name"]]
    print(c, "k_bgzf_deflate launches", len(v), "avg KB", st.mean(v) if v else None)
PY
