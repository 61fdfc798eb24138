#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
# usage: tools/profile_bench.sh <round-tag>
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# per-kernel durations: the sort leg's three streams are serialised for this pass so that every launch
# runs alone (overlapped launches share the device and their durations say little about the kernel)
MGX_SORTDEDUP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_sq2.log 2>&1
cd $OUT && find . -name "*.csv" | head -50
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
