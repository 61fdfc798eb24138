"""CPU tests of the Smith-Waterman oracle (oracle/smithwaterman_oracle.c): against the golden vectors
produced by the reference's own aligner and, when oracle/_ref is present, against that aligner live."""
import os

import numpy as np

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "smithwaterman.npz")


def load_gold(k):
    g = np.load(GOLD)
    w = {key: g[f"p{k}_{key}"] for key in ("ref_off", "ref", "alt_off", "alt", "strategy")}
    return w, tuple(int(x) for x in g[f"p{k}_params"]), g[f"p{k}_cigar"], g[f"p{k}_offset"]


def cig_bytes(row):
    z = np.nonzero(row == 0)[0]
    return bytes(row[:z[0]]) if len(z) else bytes(row)


def test_oracle_matches_golden_vectors(sw_oracle):
    for k in range(3):
        w, params, cig, off = load_gold(k)
        got_c, got_o, _ = sw_oracle.batch(w, params)
        assert np.array_equal(got_o, off)
        assert got_c == [cig_bytes(r) for r in cig]


def test_oracle_matches_reference_build(sw_oracle, sw_ref, synth):
    for seed, params in ((1, (25, -50, -110, -6)), (2, (3, -1, -4, -3)), (3, (1, -2, -3, -1))):
        w = synth.gen_sw_pairs(300, seed, ref_range=(1, 150), alt_range=(1, 120))
        stride = 2 * 150 + 40
        cig, off = sw_ref.batch(w, params, stride)
        got_c, got_o, _ = sw_oracle.batch(w, params)
        assert np.array_equal(got_o, off)
        assert got_c == [cig_bytes(r) for r in cig]


def test_text_capacity_rule(sw_oracle, sw_ref):
    """An element whose text does not fit the buffer is skipped (PairWiseSW.h:431-436)."""
    ref = np.frombuffer(b"ACGTACGTAAACCCGGGTTT", dtype=np.uint8); alt = np.frombuffer(b"ACGTTTACGTAAACGGGTT", dtype=np.uint8)
    for cap in (2, 3, 5, 8, 40):
        for st in (9, 10, 11, 12):
            c, o, _ = sw_oracle.align(ref, alt, (3, -1, -4, -3), st, cap=cap)
            rc, ro = sw_ref.align(ref, alt, (3, -1, -4, -3), st, cap=cap)
            assert c == rc and o == ro
