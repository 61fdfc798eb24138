"""Development driver: PairHMM throughput on ragged workloads (BASELINE configs[1] sub-run 2b)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
eng = pkg.PairHMMEngine(0, flags=2)
for name, kw in [("2b ragged R[32,128] H[64,256]", dict(r_range=(32, 128), h_range=(64, 256))),
                 ("short R[20,60] H[40,120]", dict(r_range=(20, 60), h_range=(40, 120))),
                 ("R=100 H=200", dict(r_range=(100, 100), h_range=(200, 200))),
                 ("R=151 H[200,400]", dict(r_range=(151, 151), h_range=(200, 400))),
                 ("R=250 H[300,500]", dict(r_range=(250, 250), h_range=(300, 500)))]:
    n = 1 << 19
    d = synth.gen_pairhmm_pairs(n, 5, **kw)
    b = eng.batch(d)
    for _ in range(3): b.run()
    eng.sync()
    t = time.perf_counter()
    for _ in range(5): b.run()
    eng.sync(); dt = (time.perf_counter() - t) / 5
    st = b.stats()
    print(f"{name:32s} {n} pairs {d['cells']/1e9:.2f} Gcells: {dt*1e3:.2f} ms/step => {d['cells']/dt/1e9:.0f} GCUPS; launches f32 {st['n_launches_f32']} rerun {st['n_rerun_f64']} ms_f32 {st['ms_f32']:.2f} ms_f64 {st['ms_f64']:.2f}", flush=True)
    b.close()
