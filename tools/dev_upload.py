"""Upload rate of packed records (pinned staging) for several staging-thread counts: run once per setting."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("fast-genomic-data-processing_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
recs, L = pkg.synth.gen_sortdedup_packed(n, 3)
eng = pkg.SortDedupEngine(0)
for it in range(3):
    t0 = time.perf_counter(); eng.upload(L, recs); dt = time.perf_counter() - t0
    print(f"threads {os.environ.get('MGX_UPLOAD_THREADS', '4')}: upload {recs.nbytes/1e9:.2f} GB in {dt*1e3:.0f} ms = {recs.nbytes/dt/1e9:.1f} GB/s")
