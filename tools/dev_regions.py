"""Row F1: many small active regions -- one synchronous call per region vs all of them in one batch."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
eng = pkg.PairHMMEngine(0)
for n_regions, nr, nh in ((1000, 40, 25), (1000, 100, 50), (200, 300, 100)):
    regs = []
    for k in range(n_regions):
        d = synth.gen_pairhmm_region(nr, nh, 1000 + k, r_range=(90, 151), h_range=(200, 400))
        d["pair_read"] = None; d["pair_hap"] = None
        regs.append(d)
    cells = sum(float(d["cells"]) for d in regs)
    eng.compute(regs[0]); eng.compute_regions(regs[:2])
    t0 = time.perf_counter(); single = [eng.compute(d) for d in regs]; t1 = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); multi = eng.compute_regions(regs); best = min(best, time.perf_counter() - t0)
    same = all(np.array_equal(a.reshape(b.shape), b) for a, b in zip(single, multi))
    print(f"{n_regions} regions of {nr} reads x {nh} haplotypes ({cells/1e9:.2f} Gcells): one call per region {t1*1e3:.1f} ms "
          f"({t1/n_regions*1e6:.0f} us each, {cells/t1/1e9:.0f} GCUPS); one batch {best*1e3:.1f} ms ({cells/best/1e9:.0f} GCUPS); identical: {same}", flush=True)
