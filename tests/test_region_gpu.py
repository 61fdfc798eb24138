"""Row F2: mgx_pairhmm_region = modifyReadQualities + PairHMM + normalizeLikelihoods +
filterPoorlyModeledEvidence on the device, against the oracle's restatement of the same steps."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT
from test_pairhmm_oracle import assert_log10_close

pytestmark = pytest.mark.gpu


def repeat_rich_region(synth, n_reads, n_haps, seed, r_range=(30, 151), h_range=(120, 300)):
    """Reads full of homopolymers and short tandem repeats (what the PCR error model keys on)."""
    d = synth.gen_pairhmm_region(n_reads, n_haps, seed, r_range=r_range, h_range=h_range)
    rng = np.random.RandomState(seed)
    bases = d["bases"].copy()
    ro = d["read_off"].astype(np.int64)
    units = [b"A", b"T", b"AC", b"GT", b"CAG", b"TTA", b"ACGT", b"AAAAC", b"ACACGT", b"ACGTACG", b"ACGTACGT", b"ACGTACGTA"]
    for r in range(n_reads):
        for _ in range(rng.randint(0, 4)):
            u = units[rng.randint(len(units))]
            reps = rng.randint(2, 26)
            s = (u * reps)[: max(1, min(len(u) * reps, ro[r + 1] - ro[r] - 1))]
            at = rng.randint(ro[r], ro[r + 1] - len(s) + 1)
            bases[at:at + len(s)] = np.frombuffer(s, dtype=np.uint8)
    d["bases"] = bases
    d["qual"] = rng.randint(2, 42, len(bases)).astype(np.uint8)
    d["ins"] = rng.randint(3, 60, len(bases)).astype(np.uint8)
    d["dele"] = rng.randint(3, 60, len(bases)).astype(np.uint8)
    mapq = rng.randint(0, 61, n_reads).astype(np.uint8)
    return d, mapq


def oracle_model(d, mapq, rate=3, thr=18, gcp=10):
    olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libpairhmm_oracle.so"))
    m = {k: d[k].copy() for k in ("qual", "ins", "dele", "gcp")}
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    ro = np.ascontiguousarray(d["read_off"], dtype=np.uint64)
    olib.ph_oracle_read_model(ctypes.c_int64(len(ro) - 1), P(ro), P(np.ascontiguousarray(d["bases"])), P(m["qual"]), P(m["ins"]),
                              P(m["dele"]), P(m["gcp"]), P(np.ascontiguousarray(mapq)), rate, thr, gcp)
    return m, olib


@pytest.mark.parametrize("n_reads,n_haps,seed", [(60, 17, 1), (200, 40, 2), (5, 1, 3)])
def test_region_matches_oracle_pipeline(engine, oracle, synth, n_reads, n_haps, seed):
    d, mapq = repeat_rich_region(synth, n_reads, n_haps, seed)
    mod, olib = oracle_model(d, mapq)
    assert (mod["ins"] != d["ins"]).any() and (mod["qual"] != d["qual"]).any()      # the model did something
    dm = dict(d); dm.update(mod)
    # (1) the device's read model produces the oracle's bytes: with normalisation off, the region call
    #     must equal the plain PairHMM on the oracle-modified arrays bit for bit
    raw, _ = engine.region(d, mapq, log10_mismapping_rate=float("-inf"))
    dm_cross = dict(dm); dm_cross["pair_read"] = None; dm_cross["pair_hap"] = None
    assert np.array_equal(raw.ravel(), engine.compute(dm_cross))
    # (2) the whole pipeline against the oracle's
    want, _ = oracle.batch(dm)
    want = want.reshape(n_reads, n_haps).copy()
    keep_want = np.zeros(n_reads, dtype=np.uint8)
    ro = np.ascontiguousarray(d["read_off"], dtype=np.uint64)
    olib.ph_oracle_normalize_filter(ctypes.c_int64(n_reads), ctypes.c_int64(n_haps), ro.ctypes.data_as(ctypes.c_void_p),
                                    want.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(-4.5), ctypes.c_double(0.02),
                                    keep_want.ctypes.data_as(ctypes.c_void_p))
    got, keep = engine.region(d, mapq)
    assert_log10_close(got.ravel(), want.ravel())
    best = got.max(axis=1)
    R = np.diff(d["read_off"].astype(np.int64))
    thr = np.minimum(2.0, np.ceil(R * 0.02)) * -4.0
    sure = np.abs(best - thr) > 1e-4                   # away from the threshold the decision must agree
    assert np.array_equal(keep[sure], keep_want[sure])
    if n_haps > 1:
        assert np.all(got >= best[:, None] - 4.5 - 1e-9)                          # capped
        assert (keep == 0).any() or n_reads < 20                                   # some random reads are dropped


def test_region_long_reads(engine, oracle, synth):
    """Reads past 1024 bases (the strip-mined class): the read model looks for tandem repeats in global
    memory instead of the LDS copy; same checks as above."""
    n_reads, n_haps = 24, 5
    d, mapq = repeat_rich_region(synth, n_reads, n_haps, 21, r_range=(700, 2600), h_range=(900, 3000))
    assert np.diff(d["read_off"].astype(np.int64)).max() > 1024
    mod, olib = oracle_model(d, mapq)
    dm = dict(d); dm.update(mod)
    dm_cross = dict(dm); dm_cross["pair_read"] = None; dm_cross["pair_hap"] = None
    raw, _ = engine.region(d, mapq, log10_mismapping_rate=float("-inf"))
    assert np.array_equal(raw.ravel(), engine.compute(dm_cross))
    want, _ = oracle.batch(dm)
    assert_log10_close(raw.ravel(), want)
    # and in one batch with short-read regions
    d2, mapq2 = repeat_rich_region(synth, 40, 9, 22)
    got = engine.regions([d2, d, d2], [mapq2, mapq, mapq2])
    for (o, k), (dd, mq) in zip(got, [(d2, mapq2), (d, mapq), (d2, mapq2)]):
        wo, wk = engine.region(dd, mq)
        assert np.array_equal(o, wo) and np.array_equal(k, wk)


def test_region_model_off(engine, synth):
    d, mapq = repeat_rich_region(synth, 30, 8, 9)
    plain = dict(d); plain["pair_read"] = None; plain["pair_hap"] = None
    # no PCR model, thresholds that change nothing, caller's gcp kept, no normalisation = plain PairHMM
    d2 = dict(d); d2["qual"] = np.maximum(d["qual"], 18); d2["ins"] = np.maximum(d["ins"], 6); d2["dele"] = np.maximum(d["dele"], 6)
    p2 = dict(d2); p2["pair_read"] = None; p2["pair_hap"] = None
    got, keep = engine.region(d2, np.full(30, 255, dtype=np.uint8), pcr_rate_factor=0, constant_gcp=-1,
                              log10_mismapping_rate=float("-inf"))
    assert np.array_equal(got.ravel(), engine.compute(p2))


def test_several_regions_in_one_batch(engine, synth):
    """rows F1 + F2: mgx_pairhmm_regions returns, per region, exactly what mgx_pairhmm_region returns"""
    shapes = [(60, 17, 11), (5, 1, 12), (200, 40, 13), (33, 7, 14)]
    regs, mqs = [], []
    for n_reads, n_haps, seed in shapes:
        d, mapq = repeat_rich_region(synth, n_reads, n_haps, seed)
        regs.append(d); mqs.append(mapq)
    got = engine.regions(regs, mqs)
    for d, mq, (o, k) in zip(regs, mqs, got):
        wo, wk = engine.region(d, mq)
        assert np.array_equal(o, wo) and np.array_equal(k, wk)
    # a different model applies to all regions alike
    got2 = engine.regions(regs[:2], mqs[:2], pcr_rate_factor=1, log10_mismapping_rate=-3.0)
    for d, mq, (o, k) in zip(regs[:2], mqs[:2], got2):
        wo, wk = engine.region(d, mq, pcr_rate_factor=1, log10_mismapping_rate=-3.0)
        assert np.array_equal(o, wo) and np.array_equal(k, wk)
