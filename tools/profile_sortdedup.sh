#!/bin/bash
# rocprofv3 kernel stats of the sort / mark-duplicate pipeline (run through gpurun).
set -o pipefail
TAG=${1:-r01}; N=${2:-200000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_sd_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/dev_sortdedup.py $N > $OUT/run.log 2>&1
cat $OUT/run.log
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
