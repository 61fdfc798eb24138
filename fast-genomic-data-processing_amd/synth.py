"""Deterministic synthetic workloads for the two hot paths (tests, golden fixtures, bench).

Everything is derived from splitmix64 so a (seed, index) pair names the same bytes on every
machine: pair ``i`` of a workload with seed ``S`` draws its random words from the splitmix64
stream whose state starts at ``S ^ (i * 0xD1B54A32D192ED03)``  (SURVEY.md section 8d).

The layouts returned are exactly the packed host layouts the C-ABI takes
(include/mgx_pairhmm.h, include/mgx_sortdedup.h).
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_STREAM = np.uint64(0xD1B54A32D192ED03)


class SplitMix:
    """Vectorised splitmix64: one independent stream per element of ``state``."""

    def __init__(self, seed, n):
        with np.errstate(over="ignore"):
            self.s = np.uint64(seed) ^ (np.arange(n, dtype=np.uint64) * _STREAM)

    def next(self):
        with np.errstate(over="ignore"):
            self.s = self.s + _GOLD
            z = self.s.copy()
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def bytes(self, nbytes):
        """[n, nbytes] uint8 matrix, 8 bytes per draw."""
        ndraw = (nbytes + 7) // 8
        cols = [self.next() for _ in range(ndraw)]
        m = np.stack(cols, axis=1).view(np.uint8)  # little-endian host
        return m[:, :nbytes]


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def gen_pairhmm_pairs(n_pairs, seed, r_range=(128, 128), h_range=(256, 256),
                      sub_rate=1 / 64, n_rate=1 / 512, random_read_rate=1 / 128,
                      qual_range=(6, 41), gap_range=(10, 45), gcp=10, hap_n_rate=0.0):
    """Independent (read, haplotype) pairs: pair i uses read i and haplotype i.

    hap: uniform ACGT (optionally 'N' at hap_n_rate); read: a window of the haplotype at a
    random offset with substitutions and 'N's, or (at random_read_rate) a completely random
    read, which drives the likelihood below 1e-28f and forces the fp64 re-run.
    """
    rmin, rmax = r_range
    hmin, hmax = h_range
    g = SplitMix(seed, n_pairs)
    w = g.next()
    R = (rmin + (w & np.uint64(0xFFFF)) % np.uint64(rmax - rmin + 1)).astype(np.int64)
    H = (hmin + ((w >> np.uint64(16)) & np.uint64(0xFFFF)) % np.uint64(hmax - hmin + 1)).astype(np.int64)
    offw = ((w >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    is_random = (((w >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64)
                 < random_read_rate * 65536.0)
    # a read longer than its haplotype is legal (R > H) but then it cannot be a window
    span = np.maximum(H - R, 0)
    off = offw % (span + 1)

    hb = g.bytes(hmax)
    hap = _ACGT[hb & 3]
    if hap_n_rate > 0:
        hn = g.bytes(hmax)
        hap = np.where(hn < int(hap_n_rate * 256), np.uint8(ord("N")), hap)

    col = np.arange(rmax, dtype=np.int64)[None, :]
    src = np.minimum(off[:, None] + col, hmax - 1)
    read = np.take_along_axis(hap, src, axis=1)
    ev = g.bytes(2 * rmax).reshape(n_pairs, rmax, 2)
    ev16 = ev[:, :, 0].astype(np.uint32) | (ev[:, :, 1].astype(np.uint32) << 8)
    rnd_base = _ACGT[g.bytes(rmax) & 3]
    sub = ev16 < int(sub_rate * 65536)
    isn = (ev16 >= int(sub_rate * 65536)) & (ev16 < int((sub_rate + n_rate) * 65536))
    read = np.where(sub | is_random[:, None], rnd_base, read)
    read = np.where(isn, np.uint8(ord("N")), read)

    ql, qh = qual_range
    gl, gh = gap_range
    qual = (ql + g.bytes(rmax) % (qh - ql + 1)).astype(np.uint8)
    ins = (gl + g.bytes(rmax) % (gh - gl + 1)).astype(np.uint8)
    dele = (gl + g.bytes(rmax) % (gh - gl + 1)).astype(np.uint8)

    rmask = col < R[:, None]
    hmask = np.arange(hmax, dtype=np.int64)[None, :] < H[:, None]
    read_off = np.zeros(n_pairs + 1, dtype=np.uint64)
    hap_off = np.zeros(n_pairs + 1, dtype=np.uint64)
    read_off[1:] = np.cumsum(R).astype(np.uint64)
    hap_off[1:] = np.cumsum(H).astype(np.uint64)
    out = dict(
        n_reads=n_pairs, n_haps=n_pairs, n_pairs=n_pairs,
        read_off=read_off, hap_off=hap_off,
        bases=np.ascontiguousarray(read[rmask]), qual=np.ascontiguousarray(qual[rmask]),
        ins=np.ascontiguousarray(ins[rmask]), dele=np.ascontiguousarray(dele[rmask]),
        gcp=np.full(int(R.sum()), gcp, dtype=np.uint8),
        hap_bases=np.ascontiguousarray(hap[hmask]),
        pair_read=np.arange(n_pairs, dtype=np.uint32), pair_hap=np.arange(n_pairs, dtype=np.uint32),
        R=R, H=H,
    )
    out["cells"] = int((R * H).sum())
    # algorithmic bytes per pair: 5R + H + 4
    out["alg_bytes"] = int((5 * R + H + 4).sum())
    return out


def gen_pairhmm_region(n_reads, n_haps, seed, r_range=(20, 128), h_range=(64, 256), dup_reads=0,
                       **kw):
    """An active region: every read against every haplotype (the shape
    VectorLoglessPairHMM::computeLog10Likelihoods builds, VectorLoglessPairHMM.cpp:71-104).
    Haplotypes are mutated copies of one backbone so reads match several of them;
    ``dup_reads`` trailing reads are exact copies of earlier ones (exercises read de-dup)."""
    base = gen_pairhmm_pairs(n_reads, seed, r_range=r_range, h_range=(h_range[1], h_range[1]), **kw)
    g = SplitMix(seed ^ 0xABCDEF, n_haps)
    w = g.next()
    hmin, hmax = h_range
    H = (hmin + (w & np.uint64(0xFFFF)) % np.uint64(hmax - hmin + 1)).astype(np.int64)
    backbone = _ACGT[SplitMix(seed ^ 0x1234, 1).bytes(hmax)[0] & 3]
    mut = g.bytes(hmax)
    alt = _ACGT[g.bytes(hmax) & 3]
    haps = np.where(mut < 6, alt, backbone[None, :])
    # reads: windows of the backbone
    R = base["R"]
    col = np.arange(base["R"].max(), dtype=np.int64)[None, :]
    rmask = col < R[:, None]
    g2 = SplitMix(seed ^ 0x77, n_reads)
    off = (g2.next() % np.uint64(max(hmin - int(R.max()), 1))).astype(np.int64)
    win = backbone[np.minimum(off[:, None] + col, hmax - 1)]
    noise = g2.bytes(col.shape[1])
    alt2 = _ACGT[g2.bytes(col.shape[1]) & 3]
    reads = np.where(noise < 4, alt2, win)
    bases = np.ascontiguousarray(reads[rmask])
    d = dict(base)
    d["bases"] = bases
    if dup_reads:
        ro = base["read_off"].astype(np.int64)
        for k in range(dup_reads):
            src, dst = k % (n_reads - dup_reads), n_reads - dup_reads + k
            if R[src] == R[dst]:
                for key in ("bases", "qual", "ins", "dele", "gcp"):
                    d[key][ro[dst]:ro[dst + 1]] = d[key][ro[src]:ro[src + 1]]
    hmask = np.arange(hmax, dtype=np.int64)[None, :] < H[:, None]
    hap_off = np.zeros(n_haps + 1, dtype=np.uint64)
    hap_off[1:] = np.cumsum(H).astype(np.uint64)
    d.update(n_reads=n_reads, n_haps=n_haps, n_pairs=n_reads * n_haps, hap_off=hap_off,
             hap_bases=np.ascontiguousarray(haps[hmask]), H=H,
             pair_read=np.repeat(np.arange(n_reads, dtype=np.uint32), n_haps),
             pair_hap=np.tile(np.arange(n_haps, dtype=np.uint32), n_reads))
    d["cells"] = int(R.sum() * H.sum())
    d["alg_bytes"] = int((5 * R[:, None] + H[None, :] + 4).sum())
    return d
