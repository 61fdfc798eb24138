"""End-to-end test of the sortmardup-compatible CLI (SAM text in, BAM + BAI out) on the GPU box:
the decoded BAM must hold the input records unchanged, in the oracle's output order, with 0x400
set exactly where the oracle says; the BAI must index them."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKGDIR = os.path.join(ROOT, "fast-genomic-data-processing_amd")
EXE = os.path.join(PKGDIR, "bin", "sortmardup")
CLI_SRC = [os.path.join(PKGDIR, "csrc", "cli", f) for f in ("sortmardup_main.cpp", "sam_text.cpp", "bam_writer.cpp")]


def build_cli():
    deps = CLI_SRC + [os.path.join(PKGDIR, "csrc", "cli", f) for f in ("sam_text.h", "bam_writer.h")]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(p) for p in deps):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(PKGDIR, "csrc", "cli")] + CLI_SRC +
                          ["-L", PKGDIR, "-lmgx", "-lz", "-Wl,-rpath,$ORIGIN/..", "-o", EXE])
    return EXE


CIG = "MIDNSHP=X"


def query_len(cig):
    return sum(c >> 4 for c in cig if (c & 15) in (0, 1, 4, 7, 8))


def make_sam(raw, path, seed=0):
    """SAM text for the parsed records of synth.gen_sortdedup_raw (qualities cut to the CIGAR's query
    length so the text is valid SAM).  Returns the list of per-record field tuples written."""
    rng = np.random.RandomState(seed)
    names = [f"chr{i + 1}" for i in range(raw["n_targets"])]
    lines = ["@HD\tVN:1.6\tSO:queryname"] + [f"@SQ\tSN:{n}\tLN:{int(l)}" for n, l in zip(names, raw["target_len"])] + \
            ["@RG\tID:grp1\tSM:sample", "@PG\tID:synth\tPN:synth"]
    header = "\n".join(lines) + "\n"
    recs = []
    out = [header]
    co, qo, no = raw["cigar_off"].astype(np.int64), raw["qual_off"].astype(np.int64), raw["qname_off"].astype(np.int64)
    for i in range(raw["n_records"]):
        cig = [int(c) for c in raw["cigar"][co[i]:co[i + 1]]]
        qual = raw["qual"][qo[i]:qo[i + 1]]
        if cig:
            qual = qual[:query_len(cig)] if query_len(cig) <= len(qual) else np.resize(qual, query_len(cig))
        qname = raw["qname"][no[i]:no[i + 1]].tobytes().decode()
        flag, tid, pos = int(raw["flag"][i]), int(raw["tid"][i]), int(raw["pos"][i])
        seq = "".join("ACGT"[b] for b in rng.randint(0, 4, len(qual))) if len(qual) else "*"
        qs = "".join(chr(int(q) + 33) for q in qual) if len(qual) else "*"
        cs = "".join(f"{c >> 4}{CIG[c & 15]}" for c in cig) if cig else "*"
        mtid = tid if (flag & 1) else -1
        mpos = int(rng.randint(0, 1000)) if mtid >= 0 else -1
        tlen = int(rng.randint(-500, 500))
        mapq = int(rng.randint(0, 61))
        nm = int(rng.randint(-3, 70000))
        tags = [f"NM:i:{nm}", "MD:Z:50A49", "RG:Z:grp1", "XA:A:c", "XB:B:s,-1,2,300"]
        rn = names[tid] if tid >= 0 else "*"
        rnext = "*" if mtid < 0 else ("=" if mtid == tid else names[mtid])
        out.append("\t".join([qname, str(flag), rn, str(pos + 1), str(mapq), cs, rnext, str(mpos + 1), str(tlen), seq, qs] + tags) + "\n")
        recs.append(dict(qname=qname, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cig, mtid=mtid, mpos=mpos, tlen=tlen,
                         seq=seq, qual=np.asarray(qual, dtype=np.uint8), nm=nm))
    with open(path, "w") as f:
        f.write("".join(out))
    return header, names, recs


def raw_from_recs(recs, target_len):
    off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)  # noqa: E731
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if sum(len(x) for x in xs) else np.zeros(0, dt)  # noqa: E731
    return dict(n_records=len(recs), flag=np.array([r["flag"] for r in recs], dtype=np.uint16),
                tid=np.array([r["tid"] for r in recs], dtype=np.int32), pos=np.array([r["pos"] for r in recs], dtype=np.int64),
                cigar_off=off([r["cigar"] for r in recs]), cigar=cat([np.asarray(r["cigar"], dtype=np.uint32) for r in recs], np.uint32),
                qual_off=off([r["qual"] for r in recs]), qual=cat([r["qual"] for r in recs], np.uint8),
                qname_off=off([r["qname"] for r in recs]),
                qname=np.frombuffer("".join(r["qname"] for r in recs).encode(), dtype=np.uint8).copy(),
                n_targets=len(target_len), target_len=np.asarray(target_len, dtype=np.uint64))


def decode_bam(path):
    data = gzip.decompress(open(path, "rb").read())
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p); p += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p); p += 4
        name = data[p:p + l - 1].decode(); p += l
        ln, = struct.unpack_from("<i", data, p); p += 4
        refs.append((name, ln))
    recs = []
    while p < len(data):
        bs, = struct.unpack_from("<i", data, p); p += 4
        tid, pos, l_qn, mapq, bn, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p)
        q = p + 32
        qname = data[q:q + l_qn - 1].decode(); q += l_qn
        cig = list(struct.unpack_from(f"<{n_cig}I", data, q)); q += 4 * n_cig
        seq4 = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(seq4[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = np.frombuffer(data[q:q + l_seq], dtype=np.uint8); q += l_seq
        aux = data[q:p + bs]
        recs.append(dict(qname=qname, flag=flag, tid=tid, pos=pos, mapq=mapq, bin=bn, cigar=cig, mtid=mtid, mpos=mpos,
                         tlen=tlen, seq=seq if l_seq else "*", qual=qual, aux=aux))
        p += bs
    return text, refs, recs


def bgzf_blocks(path):
    """[(compressed offset, uncompressed offset, uncompressed size)] of every BGZF block."""
    raw = open(path, "rb").read()
    out, c, u = [], 0, 0
    while c < len(raw):
        assert raw[c:c + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", raw, c + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, c + bsize - 4)[0]
        out.append((c, u, isize))
        c += bsize; u += isize
    return out


def reg2bin(beg, end):
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_templates,threads,slice_bytes,stdin,deflate", [(1200, 4, 0, False, "device"), (150, 1, 0, False, "device"), (1200, 3, 4096, False, "zlib"),
                                                                          (600, 2, 1500, True, "device"), (9000, 5, 0, False, "device"), (9000, 5, 0, False, "pinned"),
                                                                          (1200, 3, 4096, False, "pinned")])
def test_cli_end_to_end(tmp_path, synth, sd_oracle, n_templates, threads, slice_bytes, stdin, deflate):
    """slice_bytes > 0 forces the streaming ingest to cut the text into many slices (each ending on a queryname-group
    boundary) that are parsed out of order and committed in order; the result must not depend on it."""
    raw = synth.gen_sortdedup_raw(n_templates, 41 + n_templates, n_contigs=3, contig_len=120000, dup_rate=0.3)
    sam, bam = str(tmp_path / "in.sam"), str(tmp_path / "out.bam")
    header, names, recs = make_sam(raw, sam)
    open(bam, "w").write("stale")                            # the tool must replace an existing file
    cmd = [build_cli(), "-O", bam, "-t", str(threads), "-z", deflate] + (["-s", str(slice_bytes)] if slice_bytes else [])
    if stdin:
        res = subprocess.run(cmd, stdin=open(sam, "rb"), capture_output=True, text=True)
    else:
        res = subprocess.run(cmd + ["-I", sam], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "sort + duplicate search done" in res.stdout
    if slice_bytes:
        n_slices = int(res.stdout.split(" slices")[0].split()[-1])
        assert n_slices > 20
    text, refs, got = decode_bam(bam)
    assert text == header
    assert refs == [(n, int(l)) for n, l in zip(names, raw["target_len"])]
    # expected order and duplicate flags from the oracle on the same parsed records
    raw2 = raw_from_recs(recs, raw["target_len"])
    orecs, idx, L = sd_oracle.pack(raw2)
    order, dup, _ = sd_oracle.run(L, orecs)
    assert len(got) == len(recs)
    for k, g in enumerate(got):
        src = recs[idx[order[k]]]
        want_flag = src["flag"] | (0x400 if dup[order[k]] else 0)
        assert (g["qname"], g["flag"], g["tid"], g["pos"]) == (src["qname"], want_flag, src["tid"], src["pos"]), k
        assert (g["mapq"], g["cigar"], g["mtid"], g["mpos"], g["tlen"], g["seq"]) == \
               (src["mapq"], src["cigar"], src["mtid"], src["mpos"], src["tlen"], src["seq"])
        assert np.array_equal(g["qual"], src["qual"])
        assert b"MDZ50A49\x00" in g["aux"] and b"RGZgrp1\x00" in g["aux"] and b"XAAc" in g["aux"]
        assert b"XBBs\x03\x00\x00\x00" + struct.pack("<hhh", -1, 2, 300) in g["aux"]
        nm = src["nm"]
        enc = (b"c" + struct.pack("<b", nm) if -128 <= nm < 0 else b"C" + struct.pack("<B", nm) if 0 <= nm <= 255 else
               b"S" + struct.pack("<H", nm) if 0 <= nm <= 65535 else b"I" + struct.pack("<I", nm))
        assert b"NM" + enc in g["aux"]
    # ---- the index: every chunk of every bin decodes to records of that bin
    bai = open(bam + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01"
    n_ref, = struct.unpack_from("<i", bai, 4)
    assert n_ref == len(names)
    blocks = bgzf_blocks(bam)
    c2u = {c: u for c, u, _ in blocks}
    data = gzip.decompress(open(bam, "rb").read())
    # where every record starts in the uncompressed stream, with its reference span (from the decoded records)
    rec_starts = {}
    q0 = data.index(b"BAM\x01")
    l_text, = struct.unpack_from("<i", data, 4)
    q0 = 8 + l_text
    n_r, = struct.unpack_from("<i", data, q0); q0 += 4
    for _ in range(n_r):
        l_nm, = struct.unpack_from("<i", data, q0); q0 += 4 + l_nm + 4
    while q0 < len(data):
        bs, tid, pos, l_qn, mapq, bn, n_cig = struct.unpack_from("<iiiBBHH", data, q0)
        cig = struct.unpack_from(f"<{n_cig}I", data, q0 + 36 + l_qn)
        ref_len = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        if tid >= 0:
            rec_starts.setdefault(tid, []).append((q0, (pos, pos + (ref_len if ref_len > 0 else 1))))
        q0 += 4 + bs
    p = 8
    indexed = 0
    for ref in range(n_ref):
        n_bin, = struct.unpack_from("<i", bai, p); p += 4
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, p); p += 8
            for _ in range(n_chunk):
                vb, ve = struct.unpack_from("<QQ", bai, p); p += 16
                if b == 37450:
                    continue
                ub, ue = c2u[vb >> 16] + (vb & 0xffff), c2u.get(ve >> 16, len(data)) + (ve & 0xffff)
                q = ub
                while q < ue:
                    bs, tid, pos, l_qn, mapq, bn = struct.unpack_from("<iiiBBH", data, q)
                    assert tid == ref and bn == b
                    indexed += 1
                    q += 4 + bs
                assert q == ue
        n_intv, = struct.unpack_from("<i", bai, p); p += 4
        linear = struct.unpack_from(f"<{n_intv}Q", bai, p); p += 8 * n_intv
        # linear index: window w (16 kbp) holds the smallest virtual offset of a record of this reference that overlaps it
        # (an empty window repeats the previous one), and that offset points at such a record's start
        want = {}
        for off_u, r in rec_starts.get(ref, []):
            for w in range(max(r[0], 0) >> 14, ((max(r[1], r[0] + 1) - 1) >> 14) + 1):
                want.setdefault(w, off_u)                      # the stream is coordinate-sorted: the first one seen is the smallest
        assert n_intv == (max(want) + 1 if want else 0)
        last = 0
        for w in range(n_intv):
            v = linear[w]
            if w in want:
                assert c2u[v >> 16] + (v & 0xffff) == want[w], (ref, w)
                last = v
            else:
                assert v == last, (ref, w)
    n_no_coor, = struct.unpack_from("<Q", bai, p)
    assert n_no_coor == sum(1 for r in recs if r["tid"] < 0)
    assert indexed == sum(1 for r in recs if r["tid"] >= 0)


@pytest.mark.gpu
def test_cli_device_and_zlib_deflate_hold_the_same_stream(tmp_path, synth):
    """-z device (BGZF blocks compressed on the MI355X) and -z zlib (the reference's way) differ in the compressed
    bytes only: the uncompressed BAM stream is the same, byte for byte, and so is the index once virtual offsets are
    mapped to uncompressed positions."""
    raw = synth.gen_sortdedup_raw(6000, 77, n_contigs=4, contig_len=300000, dup_rate=0.2)
    sam = str(tmp_path / "in.sam")
    make_sam(raw, sam)
    streams, sizes = [], []
    for z in ("device", "zlib", "pinned"):
        bam = str(tmp_path / f"{z}.bam")
        res = subprocess.run([build_cli(), "-I", sam, "-O", bam, "-t", "4", "-z", z], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        streams.append(gzip.decompress(open(bam, "rb").read()))
        sizes.append(os.path.getsize(bam))
        assert open(bam, "rb").read()[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert streams[0] == streams[1] == streams[2]
    assert sizes[0] < 1.25 * sizes[1] and sizes[2] < 1.25 * sizes[1]


@pytest.mark.gpu
def test_cli_index_built_in_parts_and_file_written_by_helpers(tmp_path, synth):
    """round 3: the BAI of a large output is built in contiguous parts on several threads and merged (the serial index byte for
    byte: a bin's chunks joined across the seams, linear windows by minimum), and the BAM is written by the thread that
    receives the blocks; MGX_CLI_BAI_PARTS forces the parts on a small input, MGX_CLI_WRITERS the helper threads of rounds 1-2:
    same .bam, same .bai"""
    raw = synth.gen_sortdedup_raw(9000, 78, n_contigs=5, contig_len=200000, dup_rate=0.2, frag_rate=0.05, supp_rate=0.03)
    sam = str(tmp_path / "in.sam")
    make_sam(raw, sam)
    out = {}
    for tag, env in (("serial", {}), ("parts7", {"MGX_CLI_BAI_PARTS": "7"}), ("parts64", {"MGX_CLI_BAI_PARTS": "64"}), ("helpers", {"MGX_CLI_WRITERS": "3"})):
        bam = str(tmp_path / f"{tag}.bam")
        res = subprocess.run([build_cli(), "-I", sam, "-O", bam, "-t", "8"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert res.returncode == 0, res.stderr
        out[tag] = (open(bam, "rb").read(), open(bam + ".bai", "rb").read())
    assert len(out["serial"][1]) > 500
    for tag in ("parts7", "parts64", "helpers"):
        assert out[tag] == out["serial"], tag


@pytest.mark.gpu
@pytest.mark.parametrize("deflate", ["device", "pinned", "zlib"])
def test_cli_header_only_input(tmp_path, deflate):
    """no alignment records at all: a BAM with the header, the end-of-file block and an index without entries"""
    sam, bam = str(tmp_path / "in.sam"), str(tmp_path / "out.bam")
    header = "@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:1000\n@SQ\tSN:chr2\tLN:500\n"
    open(sam, "w").write(header)
    res = subprocess.run([build_cli(), "-I", sam, "-O", bam, "-t", "2", "-z", deflate], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    text, refs, got = decode_bam(bam)
    assert text == header and refs == [("chr1", 1000), ("chr2", 500)] and got == []
    bai = open(bam + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 2


def test_cli_usage_and_build(pkg):
    exe = build_cli()
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 2 and "usage" in res.stderr


@pytest.mark.gpu
def test_cli_on_the_sam_files_the_reference_holds(tmp_path, sd_oracle):
    """the SAM files of the reference's vendored htslib test suite (tests/golden/sam_vectors.npz) through
    the whole tool: every field of every record as an independent spec-based parser reads it from the text, in the oracle's
    order with the oracle's duplicate flags; the BAM + BAI pass the checker that is pinned on htslib-written files
    (tests/test_sam_vectors.py); for index.sam the index has the structure of the reference-held index.bam.bai."""
    import json
    import sam_spec
    z = np.load(os.path.join(ROOT, "tests", "golden", "sam_vectors.npz"))
    exp = json.loads(bytes(z["expected_json"]).decode())
    for key in z.files:
        if not key.startswith("sam:"):
            continue
        name = key[4:]
        want = exp[name]
        if not want["refs"]:
            continue                                    # a SAM file without @SQ lines cannot become a BAM
        sam, bam = str(tmp_path / "in.sam"), str(tmp_path / "out.bam")
        open(sam, "wb").write(bytes(z[key]))
        res = subprocess.run([build_cli(), "-I", sam, "-O", bam, "-t", "3"], capture_output=True, text=True)
        assert res.returncode == 0, f"{name}: {res.stderr}"
        raw_bam = open(bam, "rb").read()
        data = gzip.decompress(raw_bam)
        text, refs, p0 = sam_spec.decode_bam_header(data)
        assert text == want["header"] and refs == [tuple(r) for r in want["refs"]], name
        got = sam_spec.decode_bam_records(data, p0)
        recs = [dict(r, qual=np.asarray(r["qual"], dtype=np.uint8)) for r in want["records"]]
        orecs, idx, L = sd_oracle.pack(raw_from_recs(recs, [r[1] for r in want["refs"]]))
        order, dup, _ = sd_oracle.run(L, orecs)
        assert len(got) == len(recs), name
        for k, g in enumerate(got):
            src = want["records"][idx[order[k]]]
            assert g["flag"] == src["flag"] | (0x400 if dup[order[k]] else 0), (name, k)
            for f in ("qname", "tid", "pos", "mapq", "cigar", "mtid", "mpos", "tlen", "seq", "qual"):
                assert g[f] == src[f], (name, k, f)
            assert len(g["aux"]) == len(src["aux"]) and all(a[:2] == b[:2] for a, b in zip(g["aux"], src["aux"])), (name, k)
        bai = open(bam + ".bai", "rb").read()
        n_indexed = sam_spec.check_index(raw_bam, bai)
        assert n_indexed == sum(1 for r in recs if r["tid"] >= 0), name
        if name == "index.sam":
            mine, my_nocoor = sam_spec.parse_bai(bai)
            theirs, their_nocoor = sam_spec.parse_bai(bytes(z["bin:index.bam.bai"]))
            assert len(mine) == len(theirs) and my_nocoor == their_nocoor
            for ref, (a, b) in enumerate(zip(mine, theirs)):
                assert set(a["bins"]) - {37450} == set(b["bins"]) - {37450}, ref       # the same bins are populated
                assert len(a["linear"]) == len(b["linear"]), ref                        # the same 16 kbp windows are covered
                if 37450 in a["bins"] and 37450 in b["bins"]:
                    assert a["bins"][37450][1] == b["bins"][37450][1], ref              # mapped / unmapped record counts


@pytest.mark.gpu
def test_cli_falls_back_to_host_memory_when_the_device_cannot_hold_the_records(tmp_path, synth):
    """-z device needs every BAM byte in HBM; when the estimate exceeds the free device memory the tool says so up
    front and keeps the bytes in host memory (-z pinned) instead of failing in the middle of the ingest."""
    raw = synth.gen_sortdedup_raw(3000, 91, n_contigs=3, contig_len=200000, dup_rate=0.2)
    sam = str(tmp_path / "in.sam")
    make_sam(raw, sam)
    outs = []
    for env_free in (None, "1000000"):
        bam = str(tmp_path / f"o{len(outs)}.bam")
        env = dict(os.environ)
        if env_free:
            env["MGX_CLI_DEVICE_FREE"] = env_free
        res = subprocess.run([build_cli(), "-I", sam, "-O", bam, "-t", "4"], capture_output=True, text=True, env=env)
        assert res.returncode == 0, res.stderr
        assert ("keeping the BAM bytes in host memory" in res.stderr) == bool(env_free)
        outs.append(gzip.decompress(open(bam, "rb").read()))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_record_store_takes_an_empty_put(pkg):
    """put(b"") as the first call used to touch an empty chunk list"""
    import ctypes as C
    comp = pkg.BgzfCompressor(0)
    st = C.c_void_p()
    assert comp.lib.mgx_bgzf_store_create(comp.h, C.byref(st)) == 0
    addr = C.c_uint64(123)
    assert comp.lib.mgx_bgzf_store_put(st, None, C.c_uint64(0), C.byref(addr)) == 0 and addr.value == 0
    assert comp.lib.mgx_bgzf_store_reserve(st, C.c_uint64(1 << 20)) == 0
    data = np.arange(1000, dtype=np.uint8)
    assert comp.lib.mgx_bgzf_store_put(st, data.ctypes.data_as(C.c_void_p), C.c_uint64(1000), C.byref(addr)) == 0 and addr.value != 0
    fr, tot = C.c_uint64(), C.c_uint64()
    assert comp.lib.mgx_bgzf_device_memory(0, C.byref(fr), C.byref(tot)) == 0 and 0 < fr.value <= tot.value
    comp.lib.mgx_bgzf_store_destroy(st)
    comp.close()
