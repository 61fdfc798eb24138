timeout -k 10 300 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q -s 2>&1 | grep "^E \|passed\|failed\|bgzf bytes"
for cfg in "8 5 1" "10 6 1" "12 8 1"; do set -- $cfg; echo "base $1 rle $2 lazy $3: $(MGX_BGZF_PROF=1 MGX_BGZF_COST_BASE=$1 MGX_BGZF_COST_RLE=$2 MGX_BGZF_LAZY=$3 timeout -k 10 100 python tools/dev_bgzf.py 300 1024 2>&1 | grep 'ratio\|cycles')"; done
