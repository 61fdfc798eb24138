"""GPU parity tests of the Smith-Waterman path (row F4) through the C ABI, against the oracle and the
golden vectors of the reference's own aligner.  Bar: CIGAR text and offset identical."""
import os

import numpy as np
import pytest

from test_smithwaterman_oracle import cig_bytes, load_gold

pytestmark = pytest.mark.gpu


def test_golden_vectors(sw_engine):
    for k in range(3):
        w, params, cig, off = load_gold(k)
        got_c, got_o = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params)
        assert np.array_equal(got_o, off)
        assert got_c == [cig_bytes(r) for r in cig]


@pytest.mark.parametrize("ref_range,alt_range,n", [((1, 64), (1, 64), 1500), ((60, 130), (20, 150), 800),
                                                   ((250, 520), (100, 300), 300), ((900, 1100), (50, 400), 60),
                                                   ((1500, 2048), (100, 300), 20),
                                                   ((1900, 2048), (4000, 8000), 3)])      # long alternates
def test_random_vs_oracle(sw_engine, sw_oracle, synth, ref_range, alt_range, n):
    """every row class of the fill kernel (1, 2, 4, 8, 16, 32 rows per lane), all strategies"""
    for seed, params in ((11, (25, -50, -110, -6)), (12, (3, -1, -4, -3))):
        w = synth.gen_sw_pairs(n, seed + ref_range[0], ref_range=ref_range, alt_range=alt_range)
        want_c, want_o, want_s = sw_oracle.batch(w, params)
        got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params, want_score=True)
        assert np.array_equal(got_o, want_o)
        assert np.array_equal(got_s, want_s)
        assert got_c == want_c


@pytest.mark.parametrize("paired", ["0", "1"])
def test_both_shape_families(sw_engine, sw_oracle, synth, monkeypatch, paired):
    """large batches put two pairs on a wavefront (32 lanes each), small ones keep 64 lanes per pair:
    force each family over every row class"""
    monkeypatch.setenv("MGX_SW_PAIRED", paired)
    for k, (rr, n) in enumerate((((1, 70), 300), ((100, 520), 300), ((600, 2048), 24))):
        w = synth.gen_sw_pairs(n, 200 + k, ref_range=rr, alt_range=(5, 260))
        want_c, want_o, _ = sw_oracle.batch(w, (25, -50, -110, -6))
        got_c, got_o = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"])
        assert np.array_equal(got_o, want_o) and got_c == want_c


def test_single_pair_entry_point(sw_engine, sw_oracle):
    """mgx_sw_align == SmithWaterman_align argument for argument, including short text buffers"""
    ref = np.frombuffer(b"ACGTACGTAAACCCGGGTTTACGATCGATCGGCTA", dtype=np.uint8)
    alt = np.frombuffer(b"TTACGTTTACGTAAACGGGTTACGATGATCGGC", dtype=np.uint8)
    for st in (9, 10, 11, 12):
        for cap in (None, 3, 6):
            want_c, want_o, _ = sw_oracle.align(ref, alt, (25, -50, -110, -6), st, cap=cap)
            got_c, got_o = sw_engine.align(ref.tobytes(), alt.tobytes(), (25, -50, -110, -6), st, cigar_length=cap)
            assert got_c == want_c and got_o == want_o


def test_mixed_classes_keep_input_order(sw_engine, sw_oracle, synth):
    """pairs of different row classes interleaved: results come back in input order"""
    parts = [synth.gen_sw_pairs(40, 70 + k, ref_range=r, alt_range=(10, 90)) for k, r in enumerate(((5, 60), (300, 500), (70, 120), (1000, 1300)))]
    order = np.random.default_rng(5).permutation(160)
    refs, alts, strat = [], [], []
    for q in order:
        w = parts[q // 40]; p = q % 40
        refs.append(w["ref"][int(w["ref_off"][p]):int(w["ref_off"][p + 1])]); alts.append(w["alt"][int(w["alt_off"][p]):int(w["alt_off"][p + 1])])
        strat.append(w["strategy"][p])
    ro = np.zeros(161, dtype=np.uint64); ao = np.zeros(161, dtype=np.uint64)
    ro[1:] = np.cumsum([len(r) for r in refs]); ao[1:] = np.cumsum([len(r) for r in alts])
    w = dict(ref_off=ro, ref=np.concatenate(refs), alt_off=ao, alt=np.concatenate(alts), strategy=np.array(strat, dtype=np.uint8))
    want_c, want_o, _ = sw_oracle.batch(w, (25, -50, -110, -6))
    got_c, got_o = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"])
    assert np.array_equal(got_o, want_o) and got_c == want_c


def test_batches_larger_than_the_arena_are_chunked(sw_engine, sw_oracle, synth, monkeypatch):
    """a batch whose back-trace matrices exceed the arena is processed in several chunks"""
    w = synth.gen_sw_pairs(300, 91, ref_range=(40, 300), alt_range=(20, 150))
    want_c, want_o, _ = sw_oracle.batch(w, (25, -50, -110, -6))
    monkeypatch.setenv("MGX_SW_ARENA_LIMIT", str(1 << 20))          # about a dozen pairs per chunk
    got_c, got_o = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"])
    assert np.array_equal(got_o, want_o) and got_c == want_c
    assert sw_engine.stats()["n_launches"] >= 8              # one launch per chunk when every pair takes the 16-bit kernel


def test_limits_are_errors_not_faults(pkg, sw_engine):
    z = np.zeros(3000, dtype=np.uint8) + 65
    with pytest.raises(pkg.MgxError):       # reference longer than 2048
        sw_engine.align_batch([0, 2049], z[:2049], [0, 10], z[:10], [9])
    with pytest.raises(pkg.MgxError):       # empty alternate
        sw_engine.align_batch([0, 10], z[:10], [0, 0], z[:0], [9])
    with pytest.raises(pkg.MgxError):       # unknown strategy
        sw_engine.align_batch([0, 10], z[:10], [0, 10], z[:10], [3])
    c, o = sw_engine.align_batch(np.zeros(1, np.uint64), z[:0], np.zeros(1, np.uint64), z[:0], np.zeros(0, np.uint8))
    assert c == [] and len(o) == 0


def _concat(pairs, strategies):
    ro = np.zeros(len(pairs) + 1, dtype=np.uint64); ao = np.zeros(len(pairs) + 1, dtype=np.uint64)
    ro[1:] = np.cumsum([len(r) for r, _ in pairs]); ao[1:] = np.cumsum([len(a) for _, a in pairs])
    return dict(ref_off=ro, ref=np.concatenate([r for r, _ in pairs]), alt_off=ao, alt=np.concatenate([a for _, a in pairs]),
                strategy=np.array(strategies, dtype=np.uint8))


@pytest.mark.parametrize("i16", ["0", "1"])
@pytest.mark.parametrize("paired", ["0", "1"])
def test_packed_16_bit_fill_and_32_bit_fill_agree_with_the_oracle(sw_engine, sw_oracle, synth, monkeypatch, i16, paired):
    """round 3: pairs whose scores provably fit 16 bits are filled two to a lane group with packed arithmetic and
    nibble back-trace (k_sw_fill16); MGX_SW_I16=0 sends every pair through the 32-bit kernel.  Every row class of
    both, odd class sizes (filler jobs), both parameter sets, all strategies."""
    monkeypatch.setenv("MGX_SW_I16", i16)
    monkeypatch.setenv("MGX_SW_PAIRED", paired)
    for k, (rr, ar, n) in enumerate((((1, 40), (1, 60), 501), ((30, 470), (5, 260), 777), ((400, 1100), (20, 200), 101))):
        for params in ((25, -50, -110, -6), (3, -1, -4, -3)):
            w = synth.gen_sw_pairs(n, 900 + k, ref_range=rr, alt_range=ar)
            want_c, want_o, want_s = sw_oracle.batch(w, params)
            got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params, want_score=True)
            assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c
            st = sw_engine.stats()
            assert (st["n_pairs_i16"] > 0) == (i16 == "1")


def test_16_bit_admission(sw_engine, sw_oracle, synth):
    """the 16-bit kernel takes a pair only when the bound on its scores holds: large parameters, long references and
    long alternates fall back pair by pair, inside one batch"""
    w = synth.gen_sw_pairs(200, 77, ref_range=(50, 300), alt_range=(20, 150))
    for params, expect in (((25, -50, -110, -6), 200), ((2500, -5000, -11000, -600), 0), ((25, -50, 110, 6), None)):
        want_c, want_o, want_s = sw_oracle.batch(w, params)
        got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params, want_score=True)
        assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c
        if expect is not None:
            assert sw_engine.stats()["n_pairs_i16"] == expect
    parts = [synth.gen_sw_pairs(40, 300 + k, ref_range=r, alt_range=a) for k, (r, a) in enumerate((((100, 400), (50, 150)), ((1800, 2048), (500, 700)),
                                                                                                   ((200, 300), (3000, 5000))))]
    pairs, strat = [], []
    for q in np.random.default_rng(9).permutation(120):
        v = parts[q // 40]; p = q % 40
        pairs.append((v["ref"][int(v["ref_off"][p]):int(v["ref_off"][p + 1])], v["alt"][int(v["alt_off"][p]):int(v["alt_off"][p + 1])]))
        strat.append(v["strategy"][p])
    w = _concat(pairs, strat)
    want_c, want_o, want_s = sw_oracle.batch(w, (25, -50, -110, -6))
    got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], want_score=True)
    assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c
    assert 40 <= sw_engine.stats()["n_pairs_i16"] < 120      # the first group whole; long references with long alternates and 3000-base alternates not


def test_16_bit_fill_at_the_ends_of_its_range(sw_engine, sw_oracle):
    """sequences that drive the scores to the bounds the admission rule computes: nothing but mismatches (lowest H),
    nothing but matches (highest), one long gap, at the largest lengths the rule admits for these parameters"""
    A, Cc = np.full(1000, 65, np.uint8), np.full(1000, 67, np.uint8)
    rng = np.random.default_rng(3)
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1000)]
    pairs, strat = [], []
    for st in (9, 10, 11, 12):
        for ref, alt in ((A[:1000], Cc[:300]), (A[:1000], A[:300]), (rnd[:1000], rnd[350:650]), (rnd[:1000], np.concatenate([rnd[:150], rnd[850:1000]])),
                         (A[:300], Cc[:300]), (rnd[:300], rnd[:300]), (Cc[:7], A[:300]), (A[:1000], Cc[:1])):
            pairs.append((ref, alt)); strat.append(st)
    w = _concat(pairs, strat)
    for params in ((25, -50, -110, -6), (3, -1, -4, -3), (100, -100, -30, -30)):
        want_c, want_o, want_s = sw_oracle.batch(w, params)
        got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params, want_score=True)
        assert np.array_equal(got_s, want_s) and np.array_equal(got_o, want_o) and got_c == want_c
        assert sw_engine.stats()["n_pairs_i16"] > 0


def test_transposed_fill(sw_engine, sw_oracle, synth, monkeypatch):
    """MGX_SW_TRANSPOSE=1 (opt-in: measured slower on the realignment shape): the 32-bit kernel with the lanes over the
    alternate sequence whenever it is the shorter one"""
    monkeypatch.setenv("MGX_SW_TRANSPOSE", "1")
    monkeypatch.setenv("MGX_SW_I16", "0")
    for paired in ("0", "1"):
        monkeypatch.setenv("MGX_SW_PAIRED", paired)
        w = synth.gen_sw_pairs(400, 55, ref_range=(30, 600), alt_range=(5, 300))
        want_c, want_o, want_s = sw_oracle.batch(w, (25, -50, -110, -6))
        got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], want_score=True)
        assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c


def test_largest_sizes_and_many_chunks(sw_engine, sw_oracle, synth, monkeypatch):
    """the largest lengths the ABI takes (reference 2048, alternate 32 767: the 32-bit kernel, one pair per chunk-sized
    arena) and a batch of pairs at the 16-bit kernel's admission edge cut into many chunks by a small arena"""
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, 2048)]
    alt_long = np.concatenate([acgt[rng.integers(0, 4, 15000)], ref[100:1900], acgt[rng.integers(0, 4, 32767 - 15000 - 1800)]])
    w = _concat([(ref, alt_long), (ref[:2047], alt_long[:30000]), (ref, ref[5:2040])], [9, 12, 10])
    want_c, want_o, want_s = sw_oracle.batch(w, (25, -50, -110, -6))
    got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], want_score=True)
    assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c
    assert sw_engine.stats()["n_pairs_i16"] == 0
    monkeypatch.setenv("MGX_SW_ARENA_LIMIT", str(16 << 20))
    w = synth.gen_sw_pairs(600, 4242, ref_range=(700, 1000), alt_range=(100, 300))
    want_c, want_o, want_s = sw_oracle.batch(w, (25, -50, -110, -6))
    got_c, got_o, got_s = sw_engine.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], want_score=True)
    assert np.array_equal(got_o, want_o) and np.array_equal(got_s, want_s) and got_c == want_c
    st = sw_engine.stats()
    assert st["n_pairs_i16"] == 600 and st["n_launches"] >= 4
