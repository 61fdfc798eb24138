import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("fast-genomic-data-processing_amd"); synth = pkg.synth
from conftest import PairHMMOracle, _ensure_oracle
orc = PairHMMOracle(_ensure_oracle()); eng = pkg.PairHMMEngine(0)
n = 300
d = synth.gen_pairhmm_pairs(n, 0x5EED0002 ^ n, r_range=(513, 1024), h_range=(300, 1200), hap_n_rate=0.01)
want, wu = orc.batch(d)
b = eng.batch(d); b.run(); out, used = b.results(True)
bad = np.nonzero(np.isinf(out) != np.isinf(want))[0]
print("mismatch count", len(bad))
for i in bad[:10]: print(i, d["R"][i], d["H"][i], out[i], want[i], used[i], wu[i])
fin = ~np.isinf(want) & ~np.isinf(out)
print("max abs diff finite", np.abs(out[fin]-want[fin]).max())
