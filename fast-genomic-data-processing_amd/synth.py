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


class _PairHMMParams(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("seed", _C.c_uint64), ("first_pair", _C.c_uint64), ("rmin", _C.c_int), ("rmax", _C.c_int),
                ("hmin", _C.c_int), ("hmax", _C.c_int), ("sub_thr", _C.c_int), ("n_thr", _C.c_int),
                ("random_thr", _C.c_double), ("ql", _C.c_int), ("qh", _C.c_int), ("gl", _C.c_int), ("gh", _C.c_int),
                ("gcp", _C.c_int), ("hap_n_thr", _C.c_int)]


_SYNTH_LIB = None


def _synth_lib():
    """libmgx_synth.so: the same streams generated by threaded C++ (csrc/synth/synth_gen.cpp)."""
    global _SYNTH_LIB
    if _SYNTH_LIB is None:
        import ctypes
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "libmgx_synth.so")
        if not os.path.exists(path):
            from . import build
            build.build_synth()
        _SYNTH_LIB = ctypes.CDLL(path)
    return _SYNTH_LIB


def gen_pairhmm_pairs_fast(n_pairs, seed, r_range=(128, 128), h_range=(256, 256), sub_rate=1 / 64, n_rate=1 / 512,
                           random_read_rate=1 / 128, qual_range=(6, 41), gap_range=(10, 45), gcp=10, hap_n_rate=0.0,
                           first_pair=0, threads=0, with_pairs=True):
    """gen_pairhmm_pairs, byte for byte, by the threaded C++ generator: pairs
    [first_pair, first_pair + n_pairs) of the workload ``seed`` (so a rank can generate only its
    shard of a long stream).  ``with_pairs=False`` leaves the pair arrays out (pair i = read i x
    haplotype i is then implied by the caller)."""
    import ctypes as C
    import os
    lib = _synth_lib()
    threads = threads or min(len(os.sched_getaffinity(0)), 32)
    prm = _PairHMMParams(seed=seed, first_pair=first_pair, rmin=r_range[0], rmax=r_range[1], hmin=h_range[0], hmax=h_range[1],
                         sub_thr=int(sub_rate * 65536), n_thr=int((sub_rate + n_rate) * 65536),
                         random_thr=random_read_rate * 65536.0, ql=qual_range[0], qh=qual_range[1], gl=gap_range[0],
                         gh=gap_range[1], gcp=gcp, hap_n_thr=int(hap_n_rate * 256) if hap_n_rate > 0 else 0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    R = np.empty(n_pairs, dtype=np.int64); H = np.empty(n_pairs, dtype=np.int64)
    lib.synth_pairhmm_lengths(C.byref(prm), C.c_uint64(n_pairs), P(R), P(H), C.c_int(threads))
    read_off = np.zeros(n_pairs + 1, dtype=np.uint64); hap_off = np.zeros(n_pairs + 1, dtype=np.uint64)
    np.cumsum(R, out=read_off[1:].view(np.int64)); np.cumsum(H, out=hap_off[1:].view(np.int64))
    rb, hb = int(read_off[-1]), int(hap_off[-1])
    arrs = {k: np.empty(rb, dtype=np.uint8) for k in ("bases", "qual", "ins", "dele", "gcp")}
    hap = np.empty(hb, dtype=np.uint8)
    lib.synth_pairhmm_fill(C.byref(prm), C.c_uint64(n_pairs), P(read_off), P(hap_off), P(arrs["bases"]), P(arrs["qual"]),
                           P(arrs["ins"]), P(arrs["dele"]), P(arrs["gcp"]), P(hap), C.c_int(threads))
    out = dict(n_reads=n_pairs, n_haps=n_pairs, n_pairs=n_pairs, read_off=read_off, hap_off=hap_off, hap_bases=hap, R=R, H=H, **arrs)
    if with_pairs:
        out["pair_read"] = np.arange(n_pairs, dtype=np.uint32); out["pair_hap"] = np.arange(n_pairs, dtype=np.uint32)
    out["cells"] = int((R * H).sum())
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


# ------------------------------------------------------------------------------------------------
# sortmardup workloads
# ------------------------------------------------------------------------------------------------
REC_DTYPE = np.dtype([("coord", "<u8"), ("prime5", "<u8"), ("mate", "<u4"), ("flag", "<u2"),
                      ("score", "<u2"), ("tile", "<u2"), ("x", "<u2"), ("y", "<u2"), ("pad_", "<u2")])
assert REC_DTYPE.itemsize == 32
NO_MATE = 0xFFFFFFFF
_CIGAR_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def cigar_encode(s):
    """'5S95M' -> list of BAM uint32 (len << 4 | op)."""
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | _CIGAR_OPS[ch])
            num = ""
    return out


class RawRecords:
    """Parsed alignment records in input order (the SoA layout of mgx_raw_records_t)."""

    def __init__(self, target_len):
        self.target_len = np.asarray(target_len, dtype=np.uint64)
        self.flag, self.tid, self.pos, self.cigar, self.qual, self.qname = [], [], [], [], [], []

    def add(self, qname, flag, tid, pos, cigar, qual):
        self.qname.append(qname.encode() if isinstance(qname, str) else qname)
        self.flag.append(flag); self.tid.append(tid); self.pos.append(pos)
        self.cigar.append(cigar_encode(cigar) if isinstance(cigar, str) else list(cigar))
        self.qual.append(np.asarray(qual, dtype=np.uint8))

    def arrays(self):
        n = len(self.flag)
        off = lambda lens: np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)  # noqa: E731
        cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if n and sum(len(x) for x in xs) else np.zeros(0, dt))  # noqa: E731
        return dict(
            n_records=n, flag=np.asarray(self.flag, dtype=np.uint16), tid=np.asarray(self.tid, dtype=np.int32),
            pos=np.asarray(self.pos, dtype=np.int64),
            cigar_off=off([len(c) for c in self.cigar]), cigar=cat([np.asarray(c, dtype=np.uint32) for c in self.cigar], np.uint32),
            qual_off=off([len(q) for q in self.qual]), qual=cat(self.qual, np.uint8),
            qname_off=off([len(q) for q in self.qname]),
            qname=np.frombuffer(b"".join(self.qname), dtype=np.uint8).copy(),
            n_targets=len(self.target_len), target_len=self.target_len)


def gen_sortdedup_raw(n_templates, seed, n_contigs=5, contig_len=200_000, read_len=100, dup_rate=0.15,
                      frag_rate=0.05, supp_rate=0.03, clip_rate=0.2, unmapped_pair_rate=0.01,
                      cross_contig_rate=0.02, qname_style="illumina7", ins_range=(200, 600)):
    """A queryname-grouped synthetic SAM body in parsed form (BASELINE.json configs[3] in miniature):
    proper pairs in all four orientations, soft/hard clips on both strands, fragments whose mate
    is unmapped, supplementary records between mates, both-unmapped pairs, cross-contig pairs and
    duplicate families that copy the 5' ends of an earlier template.  (tile, x, y) are the base-65536
    digits of the template index, so no two pairs tie on every comparison field."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    rr = RawRecords([contig_len] * n_contigs)
    history = []          # (tid1, start1, rev1, tid2, start2, rev2) of earlier pairs / (tid, start, rev) frags
    for t in range(n_templates):
        tile, x, y = (t >> 32) & 0xFFFF, (t >> 16) & 0xFFFF, t & 0xFFFF
        if qname_style == "illumina7":
            qn = f"SYN:1:FC:1:{tile}:{x}:{y}"
        elif qname_style == "illumina6":
            qn = f"SYN:FC:1:{tile}:{x}:{y}"
        else:
            qn = f"read{t}"
        q1 = rng.randint(2, 42, read_len).astype(np.uint8)
        q2 = rng.randint(2, 42, read_len).astype(np.uint8)

        def clipped(start, rev):
            """(pos, cigar) of a read whose UNCLIPPED 5' end is fixed by (start, rev)."""
            lead = int(rng.randint(1, 21)) if rng.rand() < clip_rate else 0
            trail = int(rng.randint(1, 21)) if rng.rand() < clip_rate else 0
            kind = "H" if rng.rand() < 0.2 else "S"
            core = read_len - lead - trail
            if rng.rand() < 0.1 and core > 20:       # an indel inside
                a = core // 2
                body = f"{a}M2D{core - a - 3}M3I" if rng.rand() < 0.5 else f"{a}M1I{core - a - 1}M"
                reflen = (core - 3 + 2) if "D" in body else (core - 1)
            else:
                body, reflen = f"{core}M", core
            cig = (f"{lead}{kind}" if lead else "") + body + (f"{trail}{kind}" if trail else "")
            if not rev:
                pos = start + lead           # prime5 = pos - lead = start
            else:
                pos = start - trail - reflen + 1   # prime5 = pos + trail + reflen - 1 = start
            return max(pos, 0), cig

        u = rng.rand()
        if u < unmapped_pair_rate:
            rr.add(qn, 77, -1, -1, "", q1); rr.add(qn, 141, -1, -1, "", q2)
            continue
        if u < unmapped_pair_rate + frag_rate:
            # fragment: one mapped read, mate unmapped (placed at the mate's coordinate)
            if history and rng.rand() < 0.5:
                h = history[rng.randint(len(history))]
                tid, start, rev = h[0], h[1], h[2]          # collides with an earlier 5' end
            else:
                tid, start, rev = int(rng.randint(n_contigs)), int(rng.randint(1000, contig_len - 1000)), bool(rng.rand() < 0.5)
            pos, cig = clipped(start, rev)
            rr.add(qn, 73 | (16 if rev else 0), tid, pos, cig, q1)
            rr.add(qn, 133, tid, pos, "", q2)
            history.append((tid, start, rev))
            continue
        if history and len(history[-1]) == 6 and rng.rand() < dup_rate:
            k = rng.randint(len(history))
            while len(history[k]) != 6:
                k = rng.randint(len(history))
            tid1, s1, r1, tid2, s2, r2 = history[k]
        else:
            tid1 = int(rng.randint(n_contigs))
            s1 = int(rng.randint(1000, contig_len - 2000))
            o = rng.rand()
            ins = int(rng.randint(ins_range[0], ins_range[1]))
            tid2 = int(rng.randint(n_contigs)) if rng.rand() < cross_contig_rate else tid1
            if o < 0.85:   r1, r2, s2 = False, True, s1 + ins          # FR
            elif o < 0.90: r1, r2, s2 = True, False, s1 + ins          # RF
            elif o < 0.95: r1, r2, s2 = False, False, s1 + ins         # FF
            elif o < 0.98: r1, r2, s2 = True, True, s1 + ins           # RR
            else:          r1, r2, s2 = True, False, s1                # RF with equal 5' ends -> FR
        history.append((tid1, s1, r1, tid2, s2, r2))
        p1, c1 = clipped(s1, r1)
        p2, c2 = clipped(s2, r2)
        f1 = 1 | 2 | 64 | (16 if r1 else 0) | (32 if r2 else 0)
        f2 = 1 | 2 | 128 | (16 if r2 else 0) | (32 if r1 else 0)
        first_second = rng.rand() < 0.5
        recs = [(f1, tid1, p1, c1, q1), (f2, tid2, p2, c2, q2)]
        if first_second:
            recs.reverse()
        rr.add(qn, *recs[0])
        if rng.rand() < supp_rate:
            sp = int(rng.randint(1000, contig_len - 1000))
            rr.add(qn, recs[0][0] | 2048, int(rng.randint(n_contigs)), sp, f"40S{read_len - 40}M", q1)
        if rng.rand() < supp_rate / 2:
            rr.add(qn, recs[1][0] | 256, int(rng.randint(n_contigs)), int(rng.randint(1000, contig_len - 1000)), f"{read_len}M", q2)
        rr.add(qn, *recs[1])
    return rr.arrays()


def gen_sortdedup_packed(n_records, seed, n_contigs=25, contig_len=124_000_000, read_len=150, dup_rate=0.10,
                         frag_frac=0.03):
    """BASELINE.json configs[3] at scale, generated directly as packed 32-byte records (mgx_rec_t)
    in arrival order: proper pairs (mates adjacent) + fragments with an unmapped mate, 10 % of the
    pairs copying the 5' ends of an earlier pair, 10 % soft-clipped ends, (tile, x, y) = unique
    digits of the template index.  Vectorised; ~6.4 GB for 200 M records."""
    L = n_contigs * contig_len
    n_frag_t = int(n_records * frag_frac / 2)            # fragment templates: 2 records each
    n_pair_t = (n_records - 2 * n_frag_t) // 2
    n_t = n_pair_t + n_frag_t
    g = SplitMix(seed, n_t)
    w = g.next()
    start1 = (w % np.uint64(L - 4000)) + np.uint64(1000)
    w2 = g.next()
    ins = np.uint64(200) + (w2 & np.uint64(0xFFFF)) % np.uint64(400)
    o = ((w2 >> np.uint64(16)) & np.uint64(0xFF)).astype(np.int64)
    r1 = (o >= 230) & ((o < 243) | (o >= 250))          # RF / RR
    r2 = (o < 230) | ((o >= 243) & (o < 250)) & False | (o >= 250)
    r2 = (o < 230) | (o >= 250)                          # FR (90 %) / RR
    start2 = start1 + ins
    isdup = (((w2 >> np.uint64(24)) & np.uint64(0xFFFF)).astype(np.float64) < dup_rate * 65536)
    src = ((w2 >> np.uint64(40)) % np.uint64(max(n_t, 1))).astype(np.int64)
    tidx = np.arange(n_t, dtype=np.int64)
    src = np.minimum(src, np.maximum(tidx - 1, 0))       # copy an EARLIER template
    isdup &= tidx > 0
    start1 = np.where(isdup, start1[src], start1); start2 = np.where(isdup, start2[src], start2)
    r1 = np.where(isdup, r1[src], r1); r2 = np.where(isdup, r2[src], r2)
    w3 = g.next()
    clip1 = np.where((w3 & np.uint64(0xFF)) < 26, (w3 >> np.uint64(8)) % np.uint64(20) + np.uint64(1), np.uint64(0))
    clip2 = np.where(((w3 >> np.uint64(16)) & np.uint64(0xFF)) < 26, (w3 >> np.uint64(24)) % np.uint64(20) + np.uint64(1), np.uint64(0))
    sc1 = ((w3 >> np.uint64(32)) & np.uint64(0xFFFF)) % np.uint64(read_len * 30) + np.uint64(1000)
    sc2 = ((w3 >> np.uint64(48)) & np.uint64(0xFFFF)) % np.uint64(read_len * 30) + np.uint64(1000)
    is_frag = tidx >= n_pair_t
    rec = np.zeros(2 * n_t, dtype=REC_DTYPE)
    a, b = rec[0::2], rec[1::2]
    # leftmost aligned position from the unclipped 5' end (forward: +clip, reverse: -(len-1-clip))
    rl = np.uint64(read_len - 1)
    a["prime5"], b["prime5"] = start1, start2
    a["coord"] = np.where(r1, start1 - rl + clip1, start1 + clip1)
    b["coord"] = np.where(r2, start2 - rl + clip2, start2 + clip2)
    a["flag"] = 1 | 2 | 64 | np.where(r1, 16, 0) | np.where(r2, 32, 0)
    b["flag"] = 1 | 2 | 128 | np.where(r2, 16, 0) | np.where(r1, 32, 0)
    a["score"], b["score"] = sc1.astype(np.uint16), sc2.astype(np.uint16)
    for r in (a, b):
        r["tile"] = (tidx >> 32) & 0xFFFF; r["x"] = (tidx >> 16) & 0xFFFF; r["y"] = tidx & 0xFFFF
    ar = np.arange(n_t, dtype=np.uint32) * 2
    a["mate"], b["mate"] = ar + 1, ar
    # fragments: record a stays mapped & single, record b is its unmapped mate at the same coordinate
    a["mate"][is_frag] = NO_MATE; b["mate"][is_frag] = NO_MATE
    a["flag"][is_frag] = (1 | 8 | 64) | np.where(r1[is_frag], 16, 0)
    b["flag"][is_frag] = 1 | 4 | 128
    b["coord"][is_frag] = a["coord"][is_frag]; b["prime5"][is_frag] = a["coord"][is_frag]
    return rec[:n_records] if 2 * n_t >= n_records else rec, L


def gen_sortdedup_packed_fast(n_records, seed, n_contigs=25, contig_len=124_000_000, read_len=150, dup_rate=0.10,
                              frag_frac=0.03, threads=0):
    """gen_sortdedup_packed, byte for byte, by the threaded C++ generator (200 M records in seconds)."""
    import ctypes as C
    import os
    lib = _synth_lib()
    threads = threads or min(len(os.sched_getaffinity(0)), 32)
    L = n_contigs * contig_len
    n_frag_t = int(n_records * frag_frac / 2)
    n_pair_t = (n_records - 2 * n_frag_t) // 2
    n_t = n_pair_t + n_frag_t
    rec = np.empty(2 * n_t, dtype=REC_DTYPE)
    lib.synth_sortdedup_packed(C.c_uint64(seed), C.c_uint64(n_pair_t), C.c_uint64(n_frag_t), C.c_uint64(L), C.c_int(read_len),
                               C.c_double(dup_rate * 65536), rec.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return (rec[:n_records] if 2 * n_t >= n_records else rec), L


def write_sam_from_packed(path, recs, n_contigs=25, contig_len=124_000_000, read_len=150, seed=1, threads=0, fileobj=None):
    """SAM text (header + one line per packed record, queryname-grouped) for CLI runs at scale, written by the
    threaded C++ generator to ``path``, or to the open binary ``fileobj`` (e.g. the stdin pipe of the tool).
    Returns the bytes written."""
    import ctypes as C
    import os
    lib = _synth_lib()
    lib.synth_sam_text.restype = C.c_longlong
    threads = threads or min(len(os.sched_getaffinity(0)), 32)
    recs = np.ascontiguousarray(recs)
    header = "@HD\tVN:1.6\tSO:queryname\n" + "".join(f"@SQ\tSN:chr{k + 1}\tLN:{contig_len}\n" for k in range(n_contigs)) + "@PG\tID:synth\tPN:synth\n"
    if fileobj is not None:
        fileobj.write(header.encode()); fileobj.flush()
        fd = fileobj.fileno()
    else:
        with open(path, "w") as f:
            f.write(header)
        fd = -1
    n = lib.synth_sam_text(recs.ctypes.data_as(C.c_void_p), C.c_uint64(len(recs)), C.c_uint64(0), C.c_uint64(contig_len), C.c_int(n_contigs),
                           C.c_int(read_len), C.c_uint64(seed), (path or "").encode(), C.c_int(threads), C.c_int(fd))
    if n < 0:
        raise OSError(f"cannot write {path}")
    return len(header) + n


def raw_from_packed(recs, n_contigs=25, contig_len=124_000_000, read_len=150, seed=1):
    """Parsed-record arrays (the layout of mgx_raw_records_t) for the packed records of gen_sortdedup_packed: what a
    SAM/BAM reader would hold for them -- contig + position, a CIGAR with the soft clip that separates the
    coordinate from the 5' end, qualities, an Illumina queryname shared by a template's records.  Vectorised; used
    to time the reference's own classes on a sample of the benchmark workload."""
    n = len(recs)
    L = n_contigs * contig_len
    coord = recs["coord"].astype(np.int64)
    mapped = (recs["flag"] & 4) == 0
    tid = np.where(coord < L, coord // contig_len, -1).astype(np.int32)
    pos = np.where(coord < L, coord % contig_len, -1).astype(np.int64)
    rev = (recs["flag"] & 16) != 0
    clip = np.clip(np.where(rev, 0, coord - recs["prime5"].astype(np.int64)), 0, 20).astype(np.uint32)
    clip = np.where(rev, (np.arange(n) * 7 % 23 < 3) * 5, clip).astype(np.uint32)
    m = (read_len - clip).astype(np.uint32)
    has_clip = clip > 0
    n_ops = np.where(mapped, 1 + has_clip.astype(np.int64), 0)
    cigar_off = np.zeros(n + 1, dtype=np.uint64); cigar_off[1:] = np.cumsum(n_ops)
    cigar = np.zeros(int(cigar_off[-1]), dtype=np.uint32)
    first = cigar_off[:-1].astype(np.int64)
    fwd_clip = mapped & has_clip & ~rev; rev_clip = mapped & has_clip & rev; plain = mapped & ~has_clip
    cigar[first[plain]] = (m[plain] << 4)
    cigar[first[fwd_clip]] = (clip[fwd_clip] << 4) | 4; cigar[first[fwd_clip] + 1] = m[fwd_clip] << 4
    cigar[first[rev_clip]] = m[rev_clip] << 4; cigar[first[rev_clip] + 1] = (clip[rev_clip] << 4) | 4
    rng = np.random.default_rng(seed)
    qual = rng.integers(2, 42, size=n * read_len, dtype=np.uint8)
    qual_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len))
    t = np.arange(n, dtype=np.int64) // 2
    names = [b"SYN:1:FC:1:%d:%d:%d" % ((int(x) >> 32) & 0xFFFF, (int(x) >> 16) & 0xFFFF, int(x) & 0xFFFF) for x in t]
    qname_off = np.zeros(n + 1, dtype=np.uint64); qname_off[1:] = np.cumsum([len(x) for x in names])
    return dict(n_records=n, flag=recs["flag"].astype(np.uint16), tid=tid, pos=pos, cigar_off=cigar_off, cigar=cigar,
                qual_off=qual_off, qual=qual, qname_off=qname_off, qname=np.frombuffer(b"".join(names), dtype=np.uint8).copy(),
                n_targets=n_contigs, target_len=np.full(n_contigs, contig_len, dtype=np.uint64))


def gen_sw_pairs(n_pairs, seed, ref_range=(40, 400), alt_range=(20, 250), strategies=(9, 10, 11, 12)):
    """Smith-Waterman workload: reference windows and alternates that are mutated sub-ranges of them
    (substitutions, insertions, deletions, random flanks), plus a share of unrelated sequences.
    Returns dict(ref_off, ref, alt_off, alt, strategy) with ASCII bases."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    refs, alts = [], []
    for _ in range(n_pairs):
        l1 = int(rng.integers(ref_range[0], ref_range[1] + 1))
        a = rng.integers(0, 4, l1)
        kind = rng.integers(0, 8)
        if kind == 0:
            b = rng.integers(0, 4, int(rng.integers(alt_range[0], alt_range[1] + 1)))
        else:
            want = int(rng.integers(alt_range[0], alt_range[1] + 1))
            s = int(rng.integers(0, max(1, l1 - 1)))
            sub = a[s:s + want]
            rate = (0.0, 0.01, 0.03, 0.08)[int(rng.integers(0, 4))]
            u = rng.random(len(sub))
            out = []
            for ch, x in zip(sub, u):
                if x < rate / 3:
                    continue                                   # deletion
                if x < 2 * rate / 3:
                    out.append(int(rng.integers(0, 4)))        # insertion before
                    out.append(int(ch)); continue
                out.append(int(rng.integers(0, 4)) if x < rate else int(ch))
            if kind >= 6:                                      # unrelated flanks -> soft clips / overhangs
                out = list(rng.integers(0, 4, int(rng.integers(0, 12)))) + out + list(rng.integers(0, 4, int(rng.integers(0, 12))))
            b = np.array(out if out else [0], dtype=np.int64)
        refs.append(acgt[a]); alts.append(acgt[b])
    ref_off = np.zeros(n_pairs + 1, dtype=np.uint64); alt_off = np.zeros(n_pairs + 1, dtype=np.uint64)
    ref_off[1:] = np.cumsum([len(r) for r in refs]); alt_off[1:] = np.cumsum([len(r) for r in alts])
    strat = np.array(strategies, dtype=np.uint8)[rng.integers(0, len(strategies), n_pairs)]
    return dict(ref_off=ref_off, ref=np.concatenate(refs) if refs else np.zeros(0, np.uint8), alt_off=alt_off,
                alt=np.concatenate(alts) if alts else np.zeros(0, np.uint8), strategy=strat)


def gen_bam_record_bytes(n_bytes, seed, read_len=150, qual_bins=None):
    """A coordinate-sorted BAM record stream as the writer sees it (block_size + fixed fields + read name + CIGAR +
    4-bit bases + qualities + tags), the input of the BGZF compressor: names `SYN:1:FC:1:<tile>:<x>:<y>`, 150-base
    reads, random bases, qualities U[2,41] per base as in BASELINE.json configs[3] (or drawn from `qual_bins` in runs,
    the binned qualities of newer instruments), the tags of an aligner.  Vectorised; returns a uint8 array."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    name_len = 28                                         # "SYN:1:FC:1:tttt:xxxxx:yyyyy\0"
    aux = b"NMC\x01MDZ150\0RGZgroup1\0ASC\x96XSC\x00"
    rec_len = 32 + name_len + 4 + (read_len + 1) // 2 + read_len + len(aux)
    n = n_bytes // (rec_len + 4) + 1
    a = np.zeros((n, rec_len + 4), dtype=np.uint8)

    def put(col, values, dtype):
        v = np.ascontiguousarray(values.astype(dtype)).view(np.uint8).reshape(n, -1)
        a[:, col:col + v.shape[1]] = v
    pos = np.cumsum(rng.randint(0, 6, n)).astype(np.int64) + 10000
    put(0, np.full(n, rec_len), "<i4")
    put(4, np.zeros(n), "<i4"); put(8, pos, "<i4")
    a[:, 12] = name_len; a[:, 13] = rng.randint(0, 61, n)
    put(14, 4681 + (pos >> 14), "<u2"); put(16, np.ones(n), "<u2")
    put(18, rng.choice(np.array([99, 147, 83, 163, 1123, 1171]), n), "<u2"); put(20, np.full(n, read_len), "<i4")
    put(24, np.zeros(n), "<i4"); put(28, pos + rng.randint(-400, 400, n), "<i4"); put(32, rng.randint(-600, 600, n), "<i4")
    digits = lambda x, w: ((x[:, None] // 10 ** np.arange(w - 1, -1, -1)[None, :]) % 10 + 48).astype(np.uint8)  # noqa: E731
    a[:, 36:47] = np.frombuffer(b"SYN:1:FC:1:", dtype=np.uint8)
    a[:, 47:51] = digits(rng.randint(1101, 2679, n), 4); a[:, 51] = 58
    a[:, 52:57] = digits(rng.randint(1000, 32000, n), 5); a[:, 57] = 58
    a[:, 58:63] = digits(rng.randint(1000, 99000, n), 5); a[:, 63] = 0
    c = 36 + name_len
    put(c, np.full(n, read_len << 4), "<u4"); c += 4
    nb = (read_len + 1) // 2
    b = 1 << rng.randint(0, 4, (n, 2 * nb)).astype(np.uint8)
    a[:, c:c + nb] = (b[:, 0::2] << 4) | b[:, 1::2]; c += nb
    if qual_bins is None:
        a[:, c:c + read_len] = rng.randint(2, 42, (n, read_len))
    else:
        picks = rng.choice(np.asarray(qual_bins, dtype=np.uint8), (n, read_len // 3 + 1))
        a[:, c:c + read_len] = np.repeat(picks, 3, axis=1)[:, :read_len]
    c += read_len
    a[:, c:c + len(aux)] = np.frombuffer(aux, dtype=np.uint8)
    return a.reshape(-1)[:n_bytes].copy()
