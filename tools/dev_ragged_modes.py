"""Ragged sub-run 2b and region batches under the multi-class launch modes (development aid): run once per mode,
MGX_PAIRHMM_MULTI is read at first use."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-genomic-data-processing_amd"); synth = pkg.synth
eng = pkg.PairHMMEngine(0, flags=pkg.pairhmm.TIMING)
for name, kw, n in (("ragged 2b", dict(r_range=(32, 128), h_range=(64, 256)), 1 << 20), ("short U[20,60]", dict(r_range=(20, 60), h_range=(40, 120)), 1 << 20),
                    ("R U[90,151]", dict(r_range=(90, 151), h_range=(150, 400)), 1 << 19)):
    d = synth.gen_pairhmm_pairs_fast(n, 0x5EED0002, threads=8, **kw)
    b = eng.batch(d)
    for _ in range(3): b.run()
    eng.sync(); b.stats()
    t0 = time.perf_counter()
    for _ in range(10): b.run()
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    st = b.stats(); b.close()
    print(f"MULTI={os.environ.get('MGX_PAIRHMM_MULTI', '1')} {name:16s} {dt*1e3:7.3f} ms/step {d['cells']/dt/1e9:7.0f} GCUPS  f32 {st['ms_f32']:.3f} ms in {st['n_launches_f32']} launches, f64 {st['ms_f64']:.3f} ms", flush=True)
