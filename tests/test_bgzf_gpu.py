"""Row F3 (output end): the device BGZF compressor.  The checker is zlib's inflate: every block must be a valid gzip
member with the BGZF extra field whose payload inflates to the input bytes, with the right CRC-32 and ISIZE -- the
round trip the format defines; compressed bytes themselves are implementation-specific (as zlib's are by level)."""
import gzip
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comp(pkg):
    c = pkg.BgzfCompressor(0)
    yield c
    c.close()


def check_blocks(data, offsets, out, out_off):
    data = bytes(data)
    out = bytes(out)
    assert len(out_off) == len(offsets)
    assert int(out_off[-1]) == len(out)
    for i in range(len(offsets) - 1):
        blk = out[int(out_off[i]):int(out_off[i + 1])]
        want = data[int(offsets[i]):int(offsets[i + 1])]
        assert blk[:4] == b"\x1f\x8b\x08\x04" and blk[10:16] == b"\x06\x00BC\x02\x00", i      # gzip + BGZF extra field
        bsize = struct.unpack("<H", blk[16:18])[0]
        assert bsize + 1 == len(blk), i
        d = zlib.decompressobj(-15)
        got = d.decompress(blk[18:-8]) + d.flush()
        assert d.eof and d.unused_data == b"", i
        assert got == want, i
        crc, isize = struct.unpack("<II", blk[-8:])
        assert crc == zlib.crc32(want) and isize == len(want), i
    return True


def bam_like(rng, n_bytes):
    """Bytes with the statistics of BAM records: small binary header fields, read names with a shared prefix,
    4-bit packed bases, qualities from a few bins with runs, text tags."""
    parts = []
    total = 0
    k = 0
    while total < n_bytes:
        l_seq = 151
        name = f"SYN:1:FC:1:{1100 + k % 40}:{(k * 37) % 20000}:{(k * 91) % 20000}".encode() + b"\0"
        core = struct.pack("<iiBBHHHiiii", k % 25, 1000 + 3 * k, len(name), int(rng.integers(0, 61)), 4681, 1, 99, l_seq, k % 25, 1300 + 3 * k, 450)
        cigar = struct.pack("<I", l_seq << 4)
        seq = rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes()
        q = np.repeat(rng.choice([2, 11, 25, 37], size=40, p=[0.05, 0.1, 0.25, 0.6]), rng.integers(1, 9, 40))[:l_seq]
        q = np.pad(q, (0, l_seq - len(q)), constant_values=37).astype(np.uint8).tobytes()
        aux = b"NMC\x01MDZ151\0RGZgroup1\0ASC\x97XSC\x00"
        rec = core + name + cigar + seq + q + aux
        parts.append(struct.pack("<i", len(rec)) + rec)
        total += len(rec) + 4
        k += 1
    return b"".join(parts)[:n_bytes]


def test_round_trip_kinds_of_data(comp):
    rng = np.random.default_rng(1)
    pieces = [
        b"",                                                    # an empty block is an empty member
        b"A",
        b"ABC" * 5,
        bytes(65280),                                           # zeros: distance-1 matches of 258
        rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),   # incompressible: stored
        rng.integers(0, 4, 65280, dtype=np.uint8).tobytes(),     # 2 bits of entropy per byte: Huffman only
        (b"the quick brown fox jumps over the lazy dog. " * 2000)[:65280],
        bam_like(rng, 65280),
        bam_like(rng, 12345),
        bytes(range(256)) * 255,
        rng.integers(0, 256, 3, dtype=np.uint8).tobytes(),
        b"\xff" * 4,
        (b"ab" * 40000)[:65279],
    ]
    data = b"".join(pieces)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.uint64)
    out, out_off = comp.compress(data, offsets)
    assert check_blocks(data, offsets, out, out_off)
    sizes = np.diff(out_off.astype(np.int64))
    assert sizes[0] <= 31 and sizes[3] < 800 and sizes[5] < 0.34 * 65280 and sizes[6] < 2500
    assert sizes[4] == 65280 + 5 + 26                           # stored
    st = comp.stats()
    assert st["n_stored"] >= 1


def test_whole_stream_is_a_valid_multi_member_gzip_file(comp, pkg):
    rng = np.random.default_rng(2)
    data = bam_like(rng, 3_000_000)
    out, out_off = comp.compress(data)                          # cut every 0xff00 bytes
    assert len(out_off) - 1 == (len(data) + 0xff00 - 1) // 0xff00
    assert gzip.decompress(bytes(out) + pkg.bgzf.EOF_BLOCK) == data
    # compression in the neighbourhood of zlib's on the same blocks
    z1 = sum(len(zlib.compress(data[i:i + 0xff00], 1)) for i in range(0, len(data), 0xff00))
    z6 = sum(len(zlib.compress(data[i:i + 0xff00], 6)) for i in range(0, len(data), 0xff00))
    print(f"bgzf bytes: device {len(out)}, zlib -1 {z1}, zlib -6 {z6}, input {len(data)}")
    assert len(out) < 1.15 * z1


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_random_block_sizes_and_many_batches(comp, seed):
    """ragged pieces (0..65280 bytes) of mixed content, enough of them for several internal batches"""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(0, 65281, 2200 if seed == 3 else 300)
    sizes[rng.integers(0, len(sizes), 20)] = rng.integers(0, 8, 20)
    total = int(sizes.sum())
    base = np.frombuffer(bam_like(rng, 2_000_000), dtype=np.uint8)
    data = np.tile(base, total // len(base) + 1)[:total].copy()
    noise = rng.integers(0, len(sizes), len(sizes) // 10)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    for i in noise:                                              # every tenth piece is random bytes
        data[int(offsets[i]):int(offsets[i + 1])] = rng.integers(0, 256, int(sizes[i]), dtype=np.uint8)
    out, out_off = comp.compress(data, offsets)
    assert check_blocks(data.tobytes(), offsets, out, out_off)


def test_batches_in_flight_and_determinism(comp, pkg):
    rng = np.random.default_rng(6)
    datas = [bam_like(rng, 40 * 0xff00 - 17 * k) for k in range(3)]
    batches = [pkg.BgzfBatch(comp, 40 * 0xff00, 40) for _ in range(3)]
    for b, d in zip(batches, datas):
        b.input[:len(d)] = np.frombuffer(d, dtype=np.uint8)
        off = list(range(0, len(d), 0xff00)) + [len(d)]
        b.offsets[:len(off)] = off
        b.submit(len(off) - 1)
    outs = [b.wait() for b in batches]
    for (o, oo), d in zip(outs, datas):
        assert gzip.decompress(bytes(o) + pkg.bgzf.EOF_BLOCK) == d
    # the same input gives the same bytes again (no dependence on scheduling)
    o2, oo2 = comp.compress(datas[0])
    assert bytes(o2) == bytes(outs[0][0]) and np.array_equal(oo2, outs[0][1])
    for b in batches:
        b.close()


def test_bad_arguments_are_errors(comp, pkg):
    data = np.zeros(70000, dtype=np.uint8)
    with pytest.raises(pkg.MgxError):
        comp.compress(data, np.array([0, 70000], dtype=np.uint64))          # a piece over 65280 bytes
    with pytest.raises(pkg.MgxError):
        comp.compress(data, np.array([0, 500, 100], dtype=np.uint64))       # decreasing offsets


@pytest.mark.parametrize("n,seed,len_range", [(5000, 11, (40, 500)), (1, 12, (100, 101)), (300, 13, (60000, 140000)), (200000, 14, (180, 420))])
def test_record_store_emits_the_sorted_marked_stream(comp, pkg, n, seed, len_range):
    """Records put into HBM in arrival order by several calls, emitted in a random order with random duplicate marks:
    the inflated stream is block_size + record for every record in that order, FLAG |= 0x400 where marked, cut every
    65 280 bytes (records longer than a block span several); uoff and the block offsets locate every record."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(len_range[0], len_range[1], n).astype(np.uint32)
    recs = [rng.integers(0, 256, int(l), dtype=np.uint8) for l in lens]
    for r in recs:
        r[15] &= 0xfb                                        # the duplicate bit starts clear, so the expectation is a plain OR
    store = pkg.BgzfStore(comp)
    addr = np.zeros(n, dtype=np.uint64)
    k = 0
    while k < n:                                             # several put() calls of a few records each
        m = int(min(n - k, rng.integers(1, max(2, n // 7 + 2))))
        base = store.put(np.concatenate(recs[k:k + m]))
        addr[k:k + m] = base + np.concatenate([[0], np.cumsum(lens[k:k + m - 1].astype(np.uint64))]).astype(np.uint64)
        k += m
    order = rng.permutation(n).astype(np.uint32)
    dup = (rng.random(n) < 0.2).astype(np.uint8)
    blocks, block_at, uoff = store.emit(order, dup, addr, lens)
    store.close()
    want = bytearray()
    for q in range(n):
        r = recs[order[q]].copy()
        if dup[order[q]]:
            r[15] |= 4
        assert int(uoff[q]) == len(want)
        want += struct.pack("<I", len(r)) + r.tobytes()
    assert int(uoff[n]) == len(want)
    assert gzip.decompress(blocks + pkg.bgzf.EOF_BLOCK) == bytes(want)
    assert len(block_at) - 1 == (len(want) + 0xff00 - 1) // 0xff00
    for b in range(0, len(block_at) - 1, max(1, (len(block_at) - 1) // 50)):       # every block is where block_at says and holds its 65 280 bytes
        blk = blocks[int(block_at[b]):int(block_at[b + 1])]
        d = zlib.decompressobj(-15)
        assert d.decompress(blk[18:-8]) + d.flush() == bytes(want[b * 0xff00:(b + 1) * 0xff00])


def test_record_store_over_several_windows(comp, pkg):
    """~260 MB of stream: four windows of 1024 blocks, three in flight -- the rotation of the batches and the records
    that straddle two windows"""
    rng = np.random.default_rng(21)
    n = 700_000
    lens = rng.integers(200, 520, n).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    data = np.tile(np.frombuffer(bam_like(rng, 4_000_000), dtype=np.uint8), int(off[-1]) // 4_000_000 + 1)[:int(off[-1])].copy()
    data[off[:-1] + 15] &= 0xfb
    store = pkg.BgzfStore(comp)
    addr = np.zeros(n, dtype=np.uint64)
    for a in range(0, n, 90_000):
        b = min(n, a + 90_000)
        addr[a:b] = store.put(data[off[a]:off[b]]) + (off[a:b] - off[a]).astype(np.uint64)
    order = rng.permutation(n).astype(np.uint32)
    dup = (rng.random(n) < 0.1).astype(np.uint8)
    blocks, block_at, uoff = store.emit(order, dup, addr, lens)
    store.close()
    out_len = 4 + lens[order].astype(np.int64)
    want_off = np.concatenate([[0], np.cumsum(out_len)])
    assert np.array_equal(uoff.astype(np.int64), want_off)
    got = np.frombuffer(gzip.decompress(blocks + pkg.bgzf.EOF_BLOCK), dtype=np.uint8)
    assert len(got) == want_off[-1] and len(block_at) - 1 == (len(got) + 0xff00 - 1) // 0xff00 > 3 * 1024
    # every record: length field, bytes, duplicate bit (checked in vectorised form on a sample of 60 000 and at the window seams)
    win = 1024 * 0xff00
    seams = np.unique(np.clip(np.searchsorted(want_off, np.arange(1, 4) * win)[:, None] + np.arange(-2, 3)[None, :], 0, n - 1).ravel())
    for q in np.concatenate([rng.integers(0, n, 60_000), seams]):
        r = int(order[q]); a = int(want_off[q]); L = int(lens[r])
        assert int.from_bytes(got[a:a + 4].tobytes(), "little") == L
        rec = data[off[r]:off[r] + L].copy()
        if dup[r]:
            rec[15] |= 4
        assert np.array_equal(got[a + 4:a + 4 + L], rec), q


@pytest.mark.parametrize("env", [{"MGX_BGZF_GRID": "3"}, {"MGX_BGZF_LAZY": "0"}, {"MGX_BGZF_COST_BASE": "0"}, {"MGX_BGZF_COST_BASE": "30", "MGX_BGZF_COST_RLE": "30"}])
def test_knobs_change_bytes_not_content(pkg, monkeypatch, env):
    """three workgroups walking all blocks (LDS reused from block to block), no lazy evaluation, every match taken, almost
    none taken: different compressed bytes, the same content"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = pkg.BgzfCompressor(0)
    rng = np.random.default_rng(31)
    data = bam_like(rng, 1_500_000) + rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes() + bytes(100_000)
    out, out_off = c.compress(data)
    c.close()
    assert gzip.decompress(bytes(out) + pkg.bgzf.EOF_BLOCK) == data
