#!/bin/bash
# HBM traffic of the headline PairHMM kernel alone: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
set -o pipefail
OUT=${1:-$PWD/gpurun_out/prof_hmm}
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/$c" -o bench --output-format csv -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --sort-records 0 --sw-pairs 0 --bgzf-mb 0 --cli-records 0 --no-queue --no-ragged --no-regions --no-cpu-baseline --no-mixed > "$OUT/$c.json" 2> "$OUT/$c.err" || echo "$c run failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, statistics as st, sys
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{sys.argv[1]}/{c}/bench_counter_collection.csv")) if "pairhmm_fwd<float, 16, 8>" in r["Kernel_Name"] and int(r["Grid_Size"]) == 262144 * 64]
    print(c, "pairhmm_fwd<float, 16, 8> launches", len(v), "avg KB", st.mean(v) if v else None)
PY
