"""One shape, packed and scalar kernels, a few runs each (for rocprofv3 --pmc passes; development aid)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-genomic-data-processing_amd"); synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d = synth.gen_pairhmm_pairs_fast(n, 0x5EED0002, threads=8)
for flags in (pkg.pairhmm.PACKED_FP32, 0):
    eng = pkg.PairHMMEngine(0, flags=flags)
    b = eng.batch(d)
    for _ in range(3): b.run()
    eng.sync(); b.close(); eng.close()
