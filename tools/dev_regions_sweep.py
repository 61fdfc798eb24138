"""Row F1 through the queue: lanes x batch size sweep on 1000 regions of 40 x 25 (development aid).
usage: dev_regions_sweep.py [n_regions reads haps]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-genomic-data-processing_amd"); synth = pkg.synth
n_reg, nr, nh = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 40, 25)
distinct = [synth.gen_pairhmm_region(nr, nh, 1000 + g, r_range=(20, 128), h_range=(64, 256)) for g in range(min(50, n_reg))]
regions = [distinct[g % len(distinct)] for g in range(n_reg)]
cells = sum(r["cells"] for r in regions)
prep = pkg.pairhmm.prepare_regions(regions)
eng = pkg.PairHMMEngine(0)
eng.compute_regions(prepared=prep)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); eng.compute_regions(prepared=prep); ts.append(time.perf_counter() - t0)
print(f"one batch: median {np.median(ts)*1e3:.2f} ms = {cells/np.median(ts)/1e9:.0f} GCUPS (min {min(ts)*1e3:.2f})", flush=True)
big = n_reg * nr * nh > (2 << 20)
for lanes in ((2, 3, 4) if big else (1, 2, 3, 4)):
    for bp in ((131072, 262144, 393216, 524288) if big else (65536, 131072, 262144)):
        q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=lanes, depth=2, batch_pairs=bp)
        q.run_regions(prepared=prep)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); q.run_regions(prepared=prep); ts.append(time.perf_counter() - t0)
        st = q.stats(); q.close()
        print(f"queue lanes={lanes} batch={bp:6d}: median {np.median(ts)*1e3:.2f} ms = {cells/np.median(ts)/1e9:.0f} GCUPS (min {min(ts)*1e3:.2f}; {st['n_batches']} batches, pack {st['pack_seconds']/lanes*1e3:.2f} ms/lane, wait {st['wait_seconds']/lanes*1e3:.2f})", flush=True)
