# usage: dev_cli_writers.sh : the CLI's output stage against the number of file helpers (needs /dev/shm/mgx_scale.sam from dev_cli_scale.py)
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("fast-genomic-data-processing_amd")
recs, L = pkg.synth.gen_sortdedup_packed_fast(20_000_000, 0x5EED0004)
pkg.synth.write_sam_from_packed("/dev/shm/mgx_scale.sam", recs)
PY
for w in 0 1 2 4 8 16; do echo "writers $w: $(MGX_CLI_WRITERS=$w MGX_BGZF_TRACE=1 fast-genomic-data-processing_amd/bin/sortmardup -I /dev/shm/mgx_scale.sam -O /dev/shm/mgx_scale.bam -t 16 2>&1 | grep 'windows done\|output done' | tr '\n' ' ')"; done
rm -f /dev/shm/mgx_scale.sam /dev/shm/mgx_scale.bam /dev/shm/mgx_scale.bam.bai
