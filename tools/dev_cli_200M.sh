# BASELINE.json configs[3] through the tool: 200 M records as SAM text (72.7 GB, in /dev/shm), one traced run, peak RSS of the tool
python - <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("fast-genomic-data-processing_amd")
t = time.time()
recs, L = pkg.synth.gen_sortdedup_packed_fast(200_000_000, 0x5EED0004)
size = pkg.synth.write_sam_from_packed("/dev/shm/mgx_200M.sam", recs)
print(f"200000000 records, SAM text {size / 1e9:.2f} GB written in {time.time() - t:.1f} s", flush=True)
PY
python - <<'PY'
import resource, subprocess, time, os
t = time.time()
res = subprocess.run(["fast-genomic-data-processing_amd/bin/sortmardup", "-I", "/dev/shm/mgx_200M.sam", "-O", "/dev/shm/mgx_200M.bam", "-t", "16"],
                     capture_output=True, text=True, env=dict(os.environ, MGX_CLI_TRACE="1", MGX_BGZF_TRACE="1"))
wall = time.time() - t
print(res.stdout.strip()); print(res.stderr.strip()[-2500:])
rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
print(f"rc {res.returncode}, wall {wall:.2f} s = {200 / wall:.2f} Mrecords/s end to end, peak RSS {rss:.1f} GB, BAM {os.path.getsize('/dev/shm/mgx_200M.bam') / 1e9:.2f} GB")
PY
python - <<'PY'
# spot check: the BAM's first and last blocks inflate, the record count from the index metadata
import gzip, struct
with open("/dev/shm/mgx_200M.bam", "rb") as f:
    head = f.read(1 << 20)
bs = struct.unpack_from("<H", head, 16)[0] + 1
print("first block inflates to", len(gzip.decompress(head[:bs])), "bytes; magic", gzip.decompress(head[:bs])[:4])
bai = open("/dev/shm/mgx_200M.bam.bai", "rb").read()
n_ref = struct.unpack_from("<i", bai, 4)[0]; p = 8; mapped = unmapped = 0
for _ in range(n_ref):
    n_bin = struct.unpack_from("<i", bai, p)[0]; p += 4
    for _ in range(n_bin):
        b, nc = struct.unpack_from("<Ii", bai, p); p += 8
        if b == 37450:
            mapped += struct.unpack_from("<Q", bai, p + 16)[0]; unmapped += struct.unpack_from("<Q", bai, p + 24)[0]
        p += 16 * nc
    n_intv = struct.unpack_from("<i", bai, p)[0]; p += 4 + 8 * n_intv
print("index: references", n_ref, "mapped", mapped, "unmapped placed", unmapped, "no coordinate", struct.unpack_from("<Q", bai, p)[0])
PY
rm -f /dev/shm/mgx_200M.sam /dev/shm/mgx_200M.bam /dev/shm/mgx_200M.bam.bai
