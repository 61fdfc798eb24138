"""Development driver: stage timings of the sortmardup CLI on a synthetic SAM file."""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cli_gpu import build_cli
n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
rng = np.random.RandomState(1)
L, nc = 50_000_000, 4
t0 = time.time()
path = "/tmp/dev_cli.sam"
with open(path, "w") as f:
    f.write("@HD\tVN:1.6\tSO:queryname\n" + "".join(f"@SQ\tSN:chr{i+1}\tLN:{L}\n" for i in range(nc)))
    seq = "ACGT" * 37 + "AC"; 
    start = rng.randint(1000, L - 2000, n_t); ins = rng.randint(200, 600, n_t); tid = rng.randint(0, nc, n_t)
    dup = rng.rand(n_t) < 0.1
    src = np.maximum(np.arange(n_t) - rng.randint(1, 1000, n_t), 0)
    start = np.where(dup, start[src], start); ins = np.where(dup, ins[src], ins); tid = np.where(dup, tid[src], tid)
    quals = ["".join(chr(33 + q) for q in rng.randint(2, 42, 150)) for _ in range(64)]
    buf = []
    for t in range(n_t):
        qn = f"SYN:1:FC:1:{t >> 32}:{(t >> 16) & 65535}:{t & 65535}"
        p1, p2 = int(start[t]), int(start[t] + ins[t]); c = f"chr{tid[t]+1}"
        buf.append(f"{qn}\t99\t{c}\t{p1}\t60\t150M\t=\t{p2}\t{p2-p1+150}\t{seq}\t{quals[t & 63]}\tNM:i:1\tMD:Z:150\tRG:Z:g\n")
        buf.append(f"{qn}\t147\t{c}\t{p2}\t60\t150M\t=\t{p1}\t{-(p2-p1+150)}\t{seq}\t{quals[(t + 7) & 63]}\tNM:i:0\tMD:Z:150\tRG:Z:g\n")
        if len(buf) >= 20000: f.write("".join(buf)); buf = []
    f.write("".join(buf))
print(f"generated {2*n_t} records, {os.path.getsize(path)/1e6:.0f} MB in {time.time()-t0:.1f}s", flush=True)
for threads in (16, 4):
    t0 = time.time()
    res = subprocess.run([build_cli(), "-I", path, "-O", "/tmp/dev_cli.bam", "-t", str(threads), "-l", "1"], capture_output=True, text=True)
    print(f"--- -t {threads}: rc {res.returncode} wall {time.time()-t0:.2f}s; out {os.path.getsize('/tmp/dev_cli.bam')/1e6:.0f} MB")
    print(res.stdout, res.stderr[-300:])
