"""How fast the box's page cache takes a large file: N threads pwrite()-ing disjoint ranges of ONE file against N files
(is the single file's inode lock the limit?), and the same with some tens of GB of other tmpfs data already resident.
usage: dev_tmpfs_write.py [GB per test] [resident ballast GB]"""
import os, sys, threading, time
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ballast = int(sys.argv[2]) if len(sys.argv) > 2 else 0
buf = os.urandom(1 << 20) * 64            # 64 MB
def run(n_threads, one_file):
    paths = ["/dev/shm/mgx_wtest_%d" % (0 if one_file else t) for t in range(n_threads)]
    fds = [os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) for p in (paths[:1] if one_file else paths)]
    per = gb * (1 << 30) // n_threads // len(buf) * len(buf)
    def work(t):
        fd = fds[0] if one_file else fds[t]
        base = t * per if one_file else 0
        for off in range(0, per, len(buf)):
            os.pwrite(fd, buf, base + off)
    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    t0 = time.time()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.time() - t0
    for fd in fds: os.close(fd)
    t1 = time.time()
    for p in set(paths): os.unlink(p)
    print(f"{n_threads} threads, {'one file' if one_file else 'a file each'}: {per * n_threads / dt / 1e9:.2f} GB/s  (unlink {time.time() - t1:.2f} s)", flush=True)
if ballast:
    t0 = time.time()
    with open("/dev/shm/mgx_wtest_ballast", "wb") as f:
        for _ in range(ballast * 16): f.write(buf)
    print(f"ballast {ballast} GB written in {time.time() - t0:.1f} s", flush=True)
for n, one in ((1, True), (4, True), (4, False), (8, True), (8, False), (16, False)):
    run(n, one)
if ballast: os.unlink("/dev/shm/mgx_wtest_ballast")
