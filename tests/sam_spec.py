"""An independent, spec-based reading of SAM text and BAM bytes (SAMv1 sections 1.4, 4.2) for the tests of row F3: the
expected field tuples of tests/golden/sam_vectors.npz are made by parse_sam_line(), the product's output is decoded by
decode_bam_records() / decode_aux().  Test infrastructure only (nothing here is product code or reference code)."""
import struct

CIGAR_OPS = "MIDNSHP=XB"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def int_tag_type(v):
    """the smallest BAM integer type holding v -- what sam_parse1 chooses for an `i` field (htslib sam.c, sam_parse_i_vals)"""
    if v < 0:
        return "c" if v >= -128 else "s" if v >= -32768 else "i"
    return "C" if v <= 255 else "S" if v <= 65535 else "I"


def parse_aux_text(field):
    tag, typ, val = field[:2], field[3], field[5:]
    if typ == "A":
        return [tag, "A", val[0]]
    if typ == "i":
        return [tag, int_tag_type(int(val)), int(val)]
    if typ == "f":
        return [tag, "f", struct.unpack("<f", struct.pack("<f", float(val)))[0]]
    if typ in "ZH":
        return [tag, typ, val]
    if typ == "B":
        sub, items = val[0], [x for x in val[1:].split(",") if x != ""]
        if sub == "f":
            return [tag, "B" + sub, [struct.unpack("<f", struct.pack("<f", float(x)))[0] for x in items]]
        return [tag, "B" + sub, [int(x) for x in items]]
    raise ValueError(field)


def parse_sam_line(line, ref_names):
    """-> dict of BAM-level fields (0-based positions, cigar as len << 4 | op, qualities as phred or 255s for '*')"""
    f = line.rstrip("\n").split("\t")
    tid = -1 if f[2] == "*" else ref_names.index(f[2])
    cigar, num = [], ""
    if f[5] != "*":
        for ch in f[5]:
            if ch.isdigit():
                num += ch
            else:
                cigar.append(int(num) << 4 | CIGAR_OPS.index(ch)); num = ""
    mtid = tid if f[6] == "=" else -1 if f[6] == "*" else ref_names.index(f[6])
    seq = "" if f[9] == "*" else f[9]
    qual = [255] * len(seq) if f[10] == "*" else [ord(c) - 33 for c in f[10]]
    seq_bam = "".join(SEQ_CODES[SEQ_CODES.index(c.upper())] if c.upper() in SEQ_CODES else "N" for c in seq)
    return dict(qname=f[0], flag=int(f[1]), tid=tid, pos=int(f[3]) - 1, mapq=int(f[4]), cigar=cigar, mtid=mtid, mpos=int(f[7]) - 1,
                tlen=int(f[8]), seq=seq_bam, qual=qual, aux=[parse_aux_text(x) for x in f[11:] if x != ""])


def parse_sam_text(text):
    """-> (header text, [(name, length)], [record dicts]) of a whole SAM file"""
    lines = text.splitlines(keepends=True)
    header = "".join(ln for ln in lines if ln.startswith("@"))
    refs = []
    for ln in lines:
        if ln.startswith("@SQ"):
            kv = dict(x.split(":", 1) for x in ln.rstrip("\n").split("\t")[1:])
            refs.append((kv["SN"], int(kv["LN"])))
    names = [r[0] for r in refs]
    return header, refs, [parse_sam_line(ln, names) for ln in lines if not ln.startswith("@") and ln.strip()]


def ref_span(rec):
    """0-based [beg, end) on the reference (a record without reference-consuming operations spans one base)"""
    n = sum(c >> 4 for c in rec["cigar"] if (c & 15) in (0, 2, 3, 7, 8))
    return rec["pos"], rec["pos"] + (n if n > 0 else 1)


def reg2bin(beg, end):
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


_AUX_FMT = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}


def decode_aux(aux):
    out, p = [], 0
    while p < len(aux):
        tag, typ = aux[p:p + 2].decode(), chr(aux[p + 2]); p += 3
        if typ == "A":
            out.append([tag, "A", chr(aux[p])]); p += 1
        elif typ in _AUX_FMT:
            v, = struct.unpack_from(_AUX_FMT[typ], aux, p); p += struct.calcsize(_AUX_FMT[typ])
            out.append([tag, typ, v])
        elif typ in "ZH":
            e = aux.index(b"\0", p)
            out.append([tag, typ, aux[p:e].decode()]); p = e + 1
        elif typ == "B":
            sub = chr(aux[p]); n, = struct.unpack_from("<I", aux, p + 1); p += 5
            sz = struct.calcsize(_AUX_FMT[sub])
            out.append([tag, "B" + sub, [struct.unpack_from(_AUX_FMT[sub], aux, p + k * sz)[0] for k in range(n)]]); p += n * sz
        else:
            raise ValueError(f"aux type {typ!r}")
    return out


def decode_bam_records(data, p=0):
    """BAM alignment records (block_size-prefixed) in data[p:] -> list of dicts, each with its offset `at`"""
    recs = []
    while p < len(data):
        bs, = struct.unpack_from("<i", data, p)
        tid, pos, l_qn, mapq, bn, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p + 4)
        q = p + 36
        qname = data[q:q + l_qn - 1].decode(); q += l_qn
        cig = list(struct.unpack_from(f"<{n_cig}I", data, q)); q += 4 * n_cig
        seq4 = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join(SEQ_CODES[(seq4[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = list(data[q:q + l_seq]); q += l_seq
        recs.append(dict(at=p, qname=qname, flag=flag, tid=tid, pos=pos, mapq=mapq, bin=bn, cigar=cig, mtid=mtid, mpos=mpos, tlen=tlen,
                         seq=seq, qual=qual, aux=decode_aux(bytes(data[q:p + 4 + bs]))))
        p += 4 + bs
    return recs


def decode_bam_header(data):
    """-> (header text, [(name, length)], offset of the first alignment record)"""
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode().rstrip("\0")
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p); p += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p); p += 4
        name = data[p:p + l - 1].decode(); p += l
        ln, = struct.unpack_from("<i", data, p); p += 4
        refs.append((name, ln))
    return text, refs, p


def bgzf_blocks(raw):
    """[(compressed offset, uncompressed offset, uncompressed size)] of every BGZF block of a file's bytes"""
    out, c, u = [], 0, 0
    while c < len(raw):
        assert raw[c:c + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", raw, c + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, c + bsize - 4)[0]
        out.append((c, u, isize))
        c += bsize; u += isize
    return out


def parse_bai(bai):
    """-> ([{bins: {bin: [(vbeg, vend)]}, linear: [voffset]}] per reference, n_no_coor or None)"""
    assert bai[:4] == b"BAI\x01"
    n_ref, = struct.unpack_from("<i", bai, 4)
    p, refs = 8, []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", bai, p); p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, p); p += 8
            bins[b] = [struct.unpack_from("<QQ", bai, p + 16 * k) for k in range(n_chunk)]; p += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", bai, p); p += 4
        linear = list(struct.unpack_from(f"<{n_intv}Q", bai, p)); p += 8 * n_intv
        refs.append(dict(bins=bins, linear=linear))
    n_no_coor = struct.unpack_from("<Q", bai, p)[0] if p + 8 <= len(bai) else None
    return refs, n_no_coor


def check_index(bam_bytes, bai_bytes):
    """The consistency rules of SAMv1 5.2 that do not depend on where a writer cuts its BGZF blocks: every chunk of every
    bin decodes to whole records of that bin and reference; every mapped record is covered by exactly one chunk of its bin;
    linear-index window w holds the smallest start offset of a record overlapping it (empty windows repeat the previous
    entry, leading empty windows may be 0); the pseudo-bin 37450, where present, counts the mapped / unmapped records;
    n_no_coor counts the records without a reference.  Returns the number of records reached through the bins."""
    import gzip
    data = gzip.decompress(bam_bytes)
    _, refs, p0 = decode_bam_header(data)
    recs = decode_bam_records(data, p0)
    c2u = {c: u for c, u, _ in bgzf_blocks(bam_bytes)}
    to_u = lambda v: c2u[v >> 16] + (v & 0xFFFF)                      # noqa: E731
    by_at = {r["at"]: r for r in recs}
    idx, n_no_coor = parse_bai(bai_bytes)
    assert len(idx) == len(refs)
    seen = set()
    for ref, entry in enumerate(idx):
        mine = [r for r in recs if r["tid"] == ref]
        for b, chunks in entry["bins"].items():
            if b == 37450:
                assert len(chunks) == 2
                n_mapped, n_unmapped = chunks[1]
                assert n_mapped == sum(1 for r in mine if not r["flag"] & 4) and n_unmapped == sum(1 for r in mine if r["flag"] & 4)
                continue
            for vb, ve in chunks:
                q, ue = to_u(vb), (to_u(ve) if (ve >> 16) in c2u else len(data))
                while q < ue:
                    r = by_at[q]
                    assert r["tid"] == ref and r["bin"] == b and reg2bin(*ref_span(r)) == b
                    assert q not in seen
                    seen.add(q)
                    q += 4 + struct.unpack_from("<i", data, q)[0]
                assert q == ue
        want = {}
        for r in mine:                                                 # coordinate-sorted: the first one seen is the smallest
            beg, end = ref_span(r)
            for w in range(max(beg, 0) >> 14, ((max(end, beg + 1) - 1) >> 14) + 1):
                want.setdefault(w, r["at"])
        assert len(entry["linear"]) == (max(want) + 1 if want else 0)
        last = None
        for w, v in enumerate(entry["linear"]):
            if w in want:
                assert to_u(v) == want[w], (ref, w)
                last = v
            else:
                assert v == (last if last is not None else v) and (last is not None or v == 0 or to_u(v) <= min(want.values())), (ref, w)
    assert len(seen) == sum(1 for r in recs if r["tid"] >= 0)
    if n_no_coor is not None:
        assert n_no_coor == sum(1 for r in recs if r["tid"] < 0)
    return len(seen)
