timeout -k 10 300 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q -s 2>&1 | grep "^E \|passed\|failed\|bgzf bytes"
MGX_BGZF_PROF=1 timeout -k 10 100 python tools/dev_bgzf.py 1000 1024 2>&1 | grep 'ratio\|cycles\|kernels\|code:\|parse alone'
timeout -k 10 100 python tools/dev_bgzf.py 1000 1024 2>&1 | grep 'ratio\|cycles\|kernels\|code:\|parse alone'
timeout -k 10 100 python tools/dev_bgzf.py 1000 4096 2>&1 | grep 'ratio\|cycles\|kernels\|code:\|parse alone'
