"""GCUPS of resident PairHMM batches over several read/haplotype shapes (kernel events, mean of 10 runs).
MGX_PAIRHMM_MIN_G=16 python tools/dev_shapes.py   -> without the narrow lane groups (A/B)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
eng = pkg.PairHMMEngine(0, flags=pkg.pairhmm.TIMING)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
shapes = [("2a 128x256", (128, 128), (256, 256)), ("ragged 32-128 x 64-256", (32, 128), (64, 256)), ("short 20-60 x 64-256", (20, 60), (64, 256)),
          ("R=100 H=200", (100, 100), (200, 200)), ("R=151 H 200-400", (151, 151), (200, 400)), ("R 20-32 H 40-100", (20, 32), (40, 100)),
          ("R 33-64 H 100-200", (33, 64), (100, 200)), ("R=64 H=128", (64, 64), (128, 128)), ("R=32 H=64", (32, 32), (64, 64)), ("R 65-96 x 100-300", (65, 96), (100, 300))]
print("MIN_G =", os.environ.get("MGX_PAIRHMM_MIN_G", "4"))
for name, rr, hr in shapes:
    d = synth.gen_pairhmm_pairs_fast(n, 0x5EED0002, r_range=rr, h_range=hr)
    b = eng.batch(d)
    for _ in range(3):
        b.run()
    eng.sync(); b.stats()
    t0 = time.perf_counter()
    for _ in range(10):
        b.run()
    eng.sync()
    dt = (time.perf_counter() - t0) / 10
    st = b.stats()
    print(f"{name:28s} {d['cells'] / dt / 1e9:8.0f} GCUPS wall  {d['cells'] / (st['ms_f32'] * 1e-3) / 1e9:8.0f} GCUPS fp32 kernels  launches {st['n_launches_f32']:2d}  "
          f"step {dt * 1e3:7.3f} ms  f32 {st['ms_f32']:7.3f}  f64 {st['ms_f64']:6.3f}", flush=True)
    b.close()
