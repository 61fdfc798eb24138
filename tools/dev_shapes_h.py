import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
eng = pkg.PairHMMEngine(0, flags=pkg.pairhmm.TIMING)
n = 1 << 19
for name, rr, hr in [("R=32 H=64", (32, 32), (64, 64)), ("R=32 H=160", (32, 32), (160, 160)), ("R=32 H=640", (32, 32), (640, 640)),
                     ("R=64 H=64", (64, 64), (64, 64)), ("R=64 H=160", (64, 64), (160, 160)), ("R=64 H=640", (64, 64), (640, 640)),
                     ("R=128 H=64", (128, 128), (64, 64)), ("R=128 H=160", (128, 128), (160, 160)), ("R=128 H=256", (128, 128), (256, 256)), ("R=128 H=1024", (128, 128), (1024, 1024))]:
    m = n if hr[0] <= 256 else n // 4
    d = synth.gen_pairhmm_pairs_fast(m, 0x5EED0002, r_range=rr, h_range=hr)
    b = eng.batch(d)
    for _ in range(3): b.run()
    eng.sync(); b.stats()
    t0 = time.perf_counter()
    for _ in range(10): b.run()
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    st = b.stats()
    G = 4 if rr[1] <= 32 else 8 if rr[1] <= 64 else 16
    steps = hr[0] + G - 1
    print(f"{name:14s} {d['cells'] / (st['ms_f32'] * 1e-3) / 1e9:7.0f} GCUPS fp32 kernels; per test case {st['ms_f32'] * 1e-3 / m * 1e9 * 1024 * 1 :8.1f} SIMD-ns ({steps} steps)", flush=True)
    b.close()
