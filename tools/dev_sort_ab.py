"""In-process A/B of sort pipeline variants selected by environment variables (same device, same
context and buffers, interleaved).  usage: dev_sort_ab.py [n_records] [VAR=a,b ...]"""
import importlib, itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
knobs = [a.split("=") for a in sys.argv[2:]] or [["MGX_SORTDEDUP_STREAMS", "1,3"]]
names = [k for k, _ in knobs]
variants = {}
for combo in itertools.product(*[v.split(",") for _, v in knobs]):
    variants[" ".join(f"{k.replace('MGX_SORTDEDUP_', '')}={v}" for k, v in zip(names, combo))] = dict(zip(names, combo))
recs, L = synth.gen_sortdedup_packed_fast(n, 0x5EED0004, threads=16)
eng = pkg.SortDedupEngine(0); eng.upload(L, recs)
res = {k: [] for k in variants}; rs = {k: [] for k in variants}; sc = {k: [] for k in variants}
ref = None
for rnd in range(7):
    for name, env in variants.items():
        os.environ.update(env)
        eng.run(); out = eng.results(); st = eng.stats()
        if rnd == 0:
            if ref is None: ref = out
            assert all(np.array_equal(a, b) for a, b in zip(ref, out)), name
        else:
            res[name].append(st["ms_total"]); sc[name].append(st["ms_radix_scatter"])
            rs[name].append(st["ms_scatter_records"] / max(1, st["n_scatter_records"]))
print("outputs identical across variants")
for k in variants:
    print(f"{k:28s} total median {np.median(res[k]):.2f} ms (min {min(res[k]):.2f}); all scatters {np.median(sc[k]):.2f} ms; record scatter {np.median(rs[k]):.3f} ms/launch (min {min(rs[k]):.3f})")
