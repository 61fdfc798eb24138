"""Upload rate of packed records host -> HBM (warm: second upload), for the staging thread count in MGX_UPLOAD_THREADS
and the wire form in MGX_SORTDEDUP_WIRE."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-genomic-data-processing_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
recs, L = pkg.synth.gen_sortdedup_packed_fast(n, 0x5EED0004)
eng = pkg.SortDedupEngine(0)
eng.upload(L, recs)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.upload(L, recs); ts.append(time.perf_counter() - t0)
print(f"threads {os.environ.get('MGX_UPLOAD_THREADS', 'default')} wire {os.environ.get('MGX_SORTDEDUP_WIRE', '24')}: "
      f"{min(ts) * 1e3:.1f} ms = {n / min(ts) / 1e6:.0f} Mrecords/s, {n * 32 / min(ts) / 1e9:.1f} GB/s of records", flush=True)
