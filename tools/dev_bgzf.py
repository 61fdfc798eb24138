"""dev: throughput of the device BGZF compressor on BAM-like bytes.  usage: dev_bgzf.py [MB] [blocks_per_batch]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("fast-genomic-data-processing_amd")
from test_bgzf_gpu import bam_like
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(7)
base = np.frombuffer(bam_like(rng, 8_000_000), dtype=np.uint8)
comp = pkg.BgzfCompressor(0)
B = 0xff00
batches = [pkg.BgzfBatch(comp, per * B, per) for _ in range(3)]
n_batches = max(3, mb * 1000000 // (per * B))
for b in batches:
    sh = int(rng.integers(0, len(base)))
    b.input[:] = np.resize(np.roll(base, -sh), per * B)
    b.offsets[:] = np.arange(per + 1, dtype=np.uint64) * B
# warm
for b in batches: b.submit(per)
for b in batches: b.wait()
t0 = time.time(); out_bytes = 0; kms = []
for i in range(n_batches):
    b = batches[i % 3]
    if i >= 3:
        o, oo = b.wait(); out_bytes += int(oo[-1]); kms.append(comp.lib and comp.stats()["ms_kernels"])
    b.submit(per)
for i in range(n_batches, n_batches + 3):
    o, oo = batches[i % 3].wait(); out_bytes += int(oo[-1])
dt = time.time() - t0
tot = n_batches * per * B
print(f"{n_batches} batches x {per} blocks: {tot/1e9:.2f} GB in {dt:.3f} s = {tot/dt/1e9:.2f} GB/s end to end (pinned in -> pinned out), ratio {out_bytes/tot:.3f}")
st = comp.stats()
for b in batches: b.close()
print("kernels of the last batch: %.3f ms = %.1f GB/s" % (st["ms_kernels"], per * B / st["ms_kernels"] / 1e6), st)
comp.close()
