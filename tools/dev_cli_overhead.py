import subprocess, time, os, sys
exe = "fast-genomic-data-processing_amd/bin/sortmardup"
open("/dev/shm/h.sam", "w").write("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:1000000\nr1\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n")
for env in ({}, {"LD_BIND_NOW": "1"}):
    for _ in range(3):
        t = time.time(); r = subprocess.run([exe, "-I", "/dev/shm/h.sam", "-O", "/dev/shm/h.bam", "-t", "16"], capture_output=True, text=True, env=dict(os.environ, MGX_CLI_TRACE="1", **env)); dt = time.time() - t
        last = [l for l in r.stdout.splitlines() if "output done" in l]
        print(env, "wall %.3f s;" % dt, last[-1] if last else r.stderr[-200:])
print(r.stderr[-600:])
