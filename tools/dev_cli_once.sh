# usage: dev_cli_once.sh [n_records] [extra CLI args...] : one traced run of the CLI on n synthetic records (default 20 M)
N=${1:-20000000}; shift
python - $N <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("fast-genomic-data-processing_amd")
recs, L = pkg.synth.gen_sortdedup_packed_fast(int(sys.argv[1]), 0x5EED0004)
pkg.synth.write_sam_from_packed("/dev/shm/mgx_once.sam", recs)
PY
for i in 1 2; do MGX_CLI_TRACE=1 MGX_BGZF_TRACE=1 fast-genomic-data-processing_amd/bin/sortmardup -I /dev/shm/mgx_once.sam -O /dev/shm/mgx_once.bam -t 16 "$@" 2>&1 | grep -v "^program\|^double"; done
rm -f /dev/shm/mgx_once.sam /dev/shm/mgx_once.bam /dev/shm/mgx_once.bam.bai
