"""In-process A/B of sort pipeline variants selected by environment variables (same device, interleaved)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
recs, L = synth.gen_sortdedup_packed(n, 0x5EED0004)
eng = pkg.SortDedupEngine(0); eng.upload(L, recs)
variants = {"one_stream": {"MGX_SORTDEDUP_STREAMS": "1"}, "three_streams": {"MGX_SORTDEDUP_STREAMS": "3"}}
res = {k: [] for k in variants}
for rnd in range(6):
    for name, env in variants.items():
        os.environ.update(env)
        eng.run(); st = eng.stats()
        if rnd: res[name].append(st["ms_total"])
for k, v in res.items():
    print(f"{k:16s} median {np.median(v):.2f} ms  min {min(v):.2f}  all {['%.2f' % x for x in v]}")
