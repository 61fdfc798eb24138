"""CPU-side checks of the sort / mark-duplicate path (run with -m "not gpu")."""
import os

import numpy as np
import pytest

from conftest import RAW_KEYS, ROOT


def load_golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "sortdedup_small.npz"))
    raw = {k: z[k] for k in RAW_KEYS}
    raw["n_records"] = int(z["n_records"])
    raw["target_len"] = z["target_len"]
    raw["n_targets"] = len(z["target_len"])
    return raw, z["expected_order"], z["expected_dup"], z["expected_arrival"]


def in_input_terms(order, dup, input_index):
    """(order, dup) over arrival indices -> over input indices."""
    d = np.zeros(len(dup), dtype=np.uint8)
    d[input_index] = dup
    return input_index[order], d


def test_oracle_matches_reference_golden(sd_oracle):
    raw, want_order, want_dup, want_arrival = load_golden()
    recs, idx, L = sd_oracle.pack(raw)
    assert np.array_equal(idx, want_arrival)              # arrival order (mates pulled adjacent)
    order, dup, counts = sd_oracle.run(L, recs)
    got_order, got_dup = in_input_terms(order, dup, idx)
    # a record that already carried 0x400 is not re-reported by the reference harness
    got_dup = got_dup & ((raw["flag"] & 0x400) == 0)
    assert np.array_equal(got_order, want_order)
    assert np.array_equal(got_dup, want_dup)
    assert counts[2] >= want_dup.sum()


@pytest.mark.parametrize("seed,style", [(1, "illumina7"), (2, "illumina6"), (3, "plain")])
def test_oracle_matches_reference_build_live(sd_oracle, sd_ref, synth, seed, style):
    raw = synth.gen_sortdedup_raw(1500, seed, qname_style=style, dup_rate=0.3 if style != "plain" else 0.0)
    recs, idx, L = sd_oracle.pack(raw)
    order, dup, _ = sd_oracle.run(L, recs)
    got_order, got_dup = in_input_terms(order, dup, idx)
    want_order, want_dup, want_arrival = sd_ref.run(raw)
    assert np.array_equal(idx, want_arrival)
    assert np.array_equal(got_order, want_order)
    if style != "plain":        # non-Illumina qnames tie on tile/x/y: the reference's winner is undefined
        assert np.array_equal(got_dup, want_dup)


def test_product_pack_equals_oracle_pack(pkg, sd_oracle, synth):
    """mgx_sortdedup_pack (host C++, product) against the oracle's restatement, byte for byte."""
    for raw in (load_golden()[0], synth.gen_sortdedup_raw(2500, 11), synth.gen_sortdedup_raw(800, 12, qname_style="illumina6")):
        recs, idx, L = pkg.sortdedup.pack(raw)
        orecs, oidx, oL = sd_oracle.pack(raw)
        assert L == oL and np.array_equal(idx, oidx)
        assert recs.tobytes() == orecs.tobytes()


def test_packed_generator_is_consistent(synth, sd_oracle):
    """The vectorised 32-byte-record generator used by the bench: mates point at each other and the
    oracle finds the duplicate families it plants."""
    recs, L = synth.gen_sortdedup_packed(200000, 0x5EED0004, n_contigs=4, contig_len=2_000_000)
    n = len(recs)
    m = recs["mate"]
    has = m != synth.NO_MATE
    assert (m[m[has]] == np.nonzero(has)[0]).all()
    order, dup, counts = sd_oracle.run(L, recs)
    assert np.all(np.diff(recs["coord"][order].astype(np.int64)) >= 0)
    assert 0.05 * n < dup.sum() < 0.25 * n
    assert counts[0] + counts[1] > 0.45 * n


def test_empty_and_tiny(sd_oracle, synth):
    rr = synth.RawRecords([1000])
    raw = rr.arrays()
    recs, idx, L = sd_oracle.pack(raw)
    assert len(recs) == 0 and L == 1000
    rr.add("a", 0, 0, 10, "5M", [30] * 5)
    recs, idx, L = sd_oracle.pack(rr.arrays())
    order, dup, _ = sd_oracle.run(L, recs)
    assert order.tolist() == [0] and dup.tolist() == [0]


def test_threaded_pack_equals_serial(pkg, sd_oracle, synth, monkeypatch):
    """Above 200 000 records mgx_sortdedup_pack cuts the input at qname changes and packs the pieces
    on several threads; the result must be byte-identical to the serial oracle."""
    raw = synth.gen_sortdedup_raw(101_000, 77, read_len=50, supp_rate=0.1, frag_rate=0.1)
    assert raw["n_records"] > 200_000
    orecs, oidx, oL = sd_oracle.pack(raw)
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("MGX_PACK_THREADS", threads)
        recs, idx, L = pkg.sortdedup.pack(raw)
        assert L == oL and np.array_equal(idx, oidx) and recs.tobytes() == orecs.tobytes()


def test_pack_rejects_decreasing_offsets(pkg, synth):
    raw = synth.gen_sortdedup_raw(50, 3)
    bad = dict(raw); bad["cigar_off"] = raw["cigar_off"].copy(); bad["cigar_off"][10] = bad["cigar_off"][12] + 5
    import pytest
    with pytest.raises(pkg.MgxError, match="monotonic"):
        pkg.sortdedup.pack(bad)


def test_pack_with_the_callers_scores(pkg, synth):
    """mgx_sortdedup_pack_scored: a SAM reader that passes over every quality character anyway hands in
    BAMRecord::score per record (qualities of at least 15 summed in 16 bits, wrapping); the records equal those of
    mgx_sortdedup_pack, which scans the qualities itself -- also for sums beyond 65 535"""
    for seed, n in ((3, 1200), (4, 1)):
        raw = synth.gen_sortdedup_raw(n, seed, dup_rate=0.3, frag_rate=0.1, supp_rate=0.05)
        q = np.asarray(raw["qual"], dtype=np.uint8).copy()
        if seed == 3:
            q[: len(q) // 3] = 93                   # long runs of the highest quality: the 16-bit sum wraps
            raw["qual"] = q
        off = np.asarray(raw["qual_off"], dtype=np.int64)
        vals = np.where(q >= 15, q, 0).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(vals)])
        score = ((cs[off[1:]] - cs[off[:-1]]) & 0xFFFF).astype(np.uint16)
        want, widx, wl = pkg.sortdedup.pack(raw)
        got, gidx, gl = pkg.sortdedup.pack(raw, score=score)
        assert np.array_equal(got, want) and np.array_equal(gidx, widx) and gl == wl
