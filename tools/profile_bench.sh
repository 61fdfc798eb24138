#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box from the repo root):
#   1. kernel trace + stats of the bench command itself
#   2. counter passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU in separate runs, no trace domains next to --pmc)
# Summaries are condensed by tools/summarize_prof.py; copy what is to be judged into profiles/.
set -o pipefail
OUT=${1:-$PWD/gpurun_out/prof_r03}
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || echo "stats run failed"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  sub=pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//; s/sq_insts_valu/sq/')
  rocprofv3 --pmc $c -d "$OUT/$sub" -o bench --output-format csv -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --sort-steps 1 --cli-records 0 --no-cpu-baseline --no-regions --no-queue --no-mixed > "$OUT/$sub.json" 2> "$OUT/$sub.err" || echo "$c run failed"
done
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -n 60 "$OUT/summary.txt"
