"""GPU parity tests of the PairHMM path (run with -m gpu): the HIP kernels, called through the
C ABI, against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest

from test_pairhmm_oracle import KEYS, TOL, assert_log10_close, load_golden

pytestmark = pytest.mark.gpu


def run(engine, d, flags=False):
    b = engine.batch(d)
    b.run()
    res = b.results(with_flags=True)
    st = b.stats()
    b.close()
    return res[0], res[1], st


@pytest.mark.parametrize("name", ["pairhmm_cfg1.npz", "pairhmm_edge.npz"])
def test_golden_vectors(engine, name):
    g = load_golden(name)
    out, used, _ = run(engine, g)
    assert_log10_close(out, g["expected"])
    # same float-first / double-fallback decision as the reference, except within an ulp of 1e-28f
    assert (used != g["used_f64"]).sum() <= 1


def test_one_shot_compute_matches_staged(engine):
    g = load_golden("pairhmm_cfg1.npz")
    a = engine.compute(g)
    b, _, _ = run(engine, g)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("r_range,h_range,n", [((1, 128), (1, 256), 20000), ((100, 128), (200, 256), 8000),
                                                ((129, 512), (10, 600), 1500), ((1, 16), (1, 40), 5000),
                                                ((129, 192), (100, 400), 6000), ((513, 1024), (300, 1200), 300)])
def test_random_ragged_vs_oracle(engine, oracle, synth, r_range, h_range, n):
    d = synth.gen_pairhmm_pairs(n, 0x5EED0002 ^ n, r_range=r_range, h_range=h_range, hap_n_rate=0.01)
    want, wused = oracle.batch(d)
    out, used, st = run(engine, d)
    assert_log10_close(out, want)
    assert (used != wused).sum() <= max(2, n // 2000)
    assert st["cells"] == d["cells"] and st["alg_bytes"] == d["alg_bytes"]
    assert st["n_rerun_f64"] == int(used.sum())


def test_region_all_pairs_with_duplicate_reads(engine, oracle, synth):
    d = synth.gen_pairhmm_region(96, 128, 77, r_range=(60, 128), h_range=(180, 256), dup_reads=8)
    want, _ = oracle.batch(d)
    out, _, _ = run(engine, d)
    assert_log10_close(out, want)


def test_force_double_matches_fp64_oracle(pkg, oracle, synth):
    """PairHMMNativeArgumentCollection.useDoublePrecision: every test case through the fp64 kernel."""
    import ctypes
    import os
    from conftest import ROOT
    d = synth.gen_pairhmm_pairs(4000, 99, r_range=(1, 128), h_range=(1, 256))
    eng = pkg.PairHMMEngine(0, flags=pkg.pairhmm.FORCE_DOUBLE)
    out, used, _ = run(eng, d)
    eng.close()
    assert used.all()
    olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libpairhmm_oracle.so"))
    olib.ph_oracle_init()
    olib.ph_oracle_prob_f64.restype = ctypes.c_double
    ro, ho = d["read_off"].astype(np.int64), d["hap_off"].astype(np.int64)
    P = lambda a, o: ctypes.c_void_p(a.ctypes.data + int(o))  # noqa: E731
    log10_init = np.log10(np.ldexp(1.0, 1020))
    for i in range(0, 4000, 7):
        R, H = int(ro[i + 1] - ro[i]), int(ho[i + 1] - ho[i])
        v = olib.ph_oracle_prob_f64(R, P(d["bases"], ro[i]), P(d["qual"], ro[i]), P(d["ins"], ro[i]),
                                    P(d["dele"], ro[i]), P(d["gcp"], ro[i]), H, P(d["hap_bases"], ho[i]))
        assert abs(out[i] - (np.log10(v) - log10_init)) < 1e-9


def test_full_size_properties(engine, oracle, synth):
    """BASELINE.json configs[1] at full size (1M test cases, R=128, H=256): a sampled oracle check
    plus size-independent properties -- order invariance and determinism."""
    n = 1 << 20
    d = synth.gen_pairhmm_pairs(n, 0x5EED0002)
    out, used, st = run(engine, d)
    assert st["cells"] == n * 128 * 256
    assert not np.isnan(out).any() and (out < 0).all()
    # every 4096th test case against the oracle
    idx = np.arange(0, n, 4096)
    sub = dict(d)
    sub["pair_read"], sub["pair_hap"] = d["pair_read"][idx], d["pair_hap"][idx]
    want, _ = oracle.batch(sub)
    assert_log10_close(out[idx], want)
    # permuting the test-case list permutes the results bit for bit
    perm = np.random.RandomState(5).permutation(n)
    dp = dict(d)
    dp["pair_read"], dp["pair_hap"] = d["pair_read"][perm], d["pair_hap"][perm]
    outp, usedp, _ = run(engine, dp)
    assert np.array_equal(outp, out[perm]) and np.array_equal(usedp, used[perm])


def test_read_longer_than_supported_is_an_error_not_a_fallback(pkg, engine, synth):
    """Reads up to 2^20 bases are computed (the strip-mined class, test_reads_longer_than_one_strip); beyond that the
    call fails with -E2BIG instead of falling back to anything."""
    d = synth.gen_pairhmm_pairs(1, 3, r_range=((1 << 20) + 1, (1 << 20) + 1), h_range=(10, 10))
    with pytest.raises(pkg.MgxError, match="row limit"):
        engine.compute(d)


def test_empty_batch(engine):
    g = load_golden("pairhmm_cfg1.npz")
    e = dict(g)
    e["pair_read"] = np.zeros(0, dtype=np.uint32)
    e["pair_hap"] = np.zeros(0, dtype=np.uint32)
    assert engine.compute(e).shape == (0,)


def test_cross_product_form_equals_explicit_pairs(engine, oracle, synth):
    """pair_read == pair_hap == NULL: every read x every haplotype, jobs generated on the device."""
    for (nr, nh, seed, rr) in [(40, 25, 1, (20, 128)), (7, 3, 2, (1, 16)), (200, 64, 3, (60, 128)), (33, 10, 4, (100, 300))]:
        d = synth.gen_pairhmm_region(nr, nh, seed, r_range=rr, h_range=(50, 256))
        explicit = engine.compute(d)
        cross = dict(d)
        cross["pair_read"] = None
        cross["pair_hap"] = None
        got = engine.compute(cross)
        assert got.shape == (nr * nh,)
        assert np.array_equal(got, explicit)
        want, _ = oracle.batch(d)
        assert_log10_close(got, want)


def test_result_does_not_depend_on_batch_composition(engine, synth):
    """A test case must give the same bits whatever else is in the batch (different wavefront
    neighbours change how many columns run in the unguarded loop versus the guarded tail)."""
    d = synth.gen_pairhmm_pairs(6000, 77, r_range=(30, 128), h_range=(20, 256))
    full = engine.compute(d)
    rng = np.random.RandomState(0)
    for trial in range(3):
        idx = np.sort(rng.choice(6000, 1500, replace=False))
        sub = dict(d)
        sub["pair_read"], sub["pair_hap"] = d["pair_read"][idx], d["pair_hap"][idx]
        assert np.array_equal(engine.compute(sub), full[idx])


def test_non_monotonic_offsets_are_rejected(pkg, engine, synth):
    d = synth.gen_pairhmm_pairs(16, 3, r_range=(10, 20), h_range=(20, 30))
    d = dict(d)
    ro = d["read_off"].copy(); ro[5], ro[6] = ro[6], ro[5]
    d["read_off"] = ro
    with pytest.raises(pkg.MgxError, match="monotonic"):
        engine.compute(d)


def test_long_haplotypes(pkg, engine, oracle, synth):
    """Haplotypes far beyond the usual few hundred bases: the per-group LDS staging buffer grows and
    the workgroup shrinks; beyond what LDS can hold the call fails loudly."""
    d = synth.gen_pairhmm_pairs(48, 11, r_range=(60, 150), h_range=(9000, 20000))
    want, _ = oracle.batch(d)
    out = engine.compute(d)
    assert_log10_close(out, want)
    d = synth.gen_pairhmm_pairs(4, 12, r_range=(100, 100), h_range=(70000, 70000))
    with pytest.raises(pkg.MgxError, match="LDS staging"):
        engine.compute(d)


def test_one_context_per_worker_thread(pkg, synth):
    """The reference runs one VectorLoglessPairHMM per worker thread (Mutect2Engine.cpp:27-29,
    main.cpp:570-574).  Four host threads, each with its own context, stream regions concurrently;
    every result must equal the single-threaded one bit for bit, and recycled slabs must not leak
    one region's data into the next."""
    import threading
    regions = [synth.gen_pairhmm_region(20 + 3 * k, 5 + k % 7, 100 + k, r_range=(30, 140), h_range=(60, 300)) for k in range(24)]
    for r in regions:
        r["pair_read"] = None; r["pair_hap"] = None
    ref_eng = pkg.PairHMMEngine(0)
    want = [ref_eng.compute(r) for r in regions]
    ref_eng.close()
    got = [None] * len(regions)
    errors = []

    def worker(tid):
        try:
            eng = pkg.PairHMMEngine(0)
            for rep in range(3):
                for k in range(tid, len(regions), 4):
                    got[k] = eng.compute(regions[k])
                    assert np.array_equal(got[k], want[k]), (tid, rep, k)
            eng.close()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    assert all(np.array_equal(g, w) for g, w in zip(got, want))


def test_batches_in_flight(engine, oracle, synth):
    """Several batches created and run before any result is fetched (the staged ABI)."""
    ds = [synth.gen_pairhmm_pairs(3000 + 500 * k, 900 + k, r_range=(10, 128), h_range=(20, 256)) for k in range(5)]
    bs = [engine.batch(d) for d in ds]
    for b in bs: b.run()
    for b in reversed(bs): b.run()          # a batch may be run again
    for d, b in zip(ds, bs):
        want, _ = oracle.batch(d)
        assert_log10_close(b.results(), want)
        b.close()


def test_many_regions_in_one_batch(engine, synth):
    """Row F1: coalescing active regions into one device batch returns, bit for bit, what one call per
    region returns (regions of different sizes and read-length classes, one of them a single pair)."""
    shapes = [(24, 16, (20, 128), (64, 256)), (1, 1, (100, 100), (200, 200)), (40, 25, (90, 151), (150, 400)),
              (7, 3, (200, 400), (300, 600)), (60, 9, (30, 60), (40, 90)), (3, 40, (151, 151), (200, 300))]
    regions = [synth.gen_pairhmm_region(nr, nh, 100 + k, r_range=rr, h_range=hr) for k, (nr, nh, rr, hr) in enumerate(shapes)]
    got = engine.compute_regions(regions)
    for d, o in zip(regions, got):
        dd = dict(d); dd["pair_read"] = None; dd["pair_hap"] = None
        want = engine.compute(dd).reshape(o.shape)
        assert np.array_equal(o, want)
    # sub-views: a region whose offset tables do not start at zero
    d = regions[2]
    ro, ho = np.asarray(d["read_off"]), np.asarray(d["hap_off"])
    sub = dict(d); sub["read_off"] = ro[5:21]; sub["hap_off"] = ho[3:12]
    full = got[2]
    part = engine.compute_regions([sub])[0]
    assert np.array_equal(part, full[5:20, 3:11])


def test_region_batch_with_empty_and_many_regions(engine, synth):
    """an empty region inside the batch contributes nothing; a thousand small regions stay exact"""
    regs = [synth.gen_pairhmm_region(6, 4, 500 + k, r_range=(20, 90), h_range=(40, 120)) for k in range(1000)]
    empty = dict(regs[0]); empty["read_off"] = np.zeros(1, dtype=np.uint64)
    regs.insert(500, empty)
    got = engine.compute_regions(regs)
    assert got[500].shape == (0, 4)
    for k in (0, 1, 499, 501, 777, 1000):
        d = dict(regs[k]); d["pair_read"] = None; d["pair_hap"] = None
        assert np.array_equal(got[k], engine.compute(d).reshape(got[k].shape))


@pytest.mark.parametrize("r_range,h_range,n,gcp", [((1025, 1100), (900, 1300), 40, 10), ((2049, 3300), (50, 700), 40, 10),
                                                    ((1025, 2500), (1000, 2600), 24, 0), ((1024, 1026), (1, 40), 30, 10)])
def test_reads_longer_than_one_strip(pkg, engine, oracle, synth, r_range, h_range, n, gcp):
    """Reads of more than 1024 bases take the strip-mined class (64 x 16 rows per strip, boundary rows through
    global memory), in fp32 and in the fp64 re-run; gap-continuation byte 0 forces the plain 8-operation form.
    The reference handles any read length."""
    d = synth.gen_pairhmm_pairs(n, 77 + n, r_range=r_range, h_range=h_range, gcp=gcp, random_read_rate=0.1)
    want, wused = oracle.batch(d)
    out, used, st = run(engine, d)
    assert_log10_close(out, want)
    assert (used != wused).sum() <= 2
    assert st["n_rerun_f64"] == int(used.sum()) and st["cells"] == d["cells"]
    # forced double precision and the queue take the same class
    eng64 = pkg.PairHMMEngine(0, flags=pkg.pairhmm.FORCE_DOUBLE)
    out64, used64, _ = run(eng64, d)
    eng64.close()
    assert used64.all()
    assert_log10_close(out64, want)
    q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=2, batch_pairs=7)
    assert np.array_equal(q.run(d), out)
    q.close()


def test_long_and_short_reads_in_one_region(engine, oracle, synth):
    """A cross-product batch whose reads span every class, including the strip-mined one."""
    d = synth.gen_pairhmm_region(24, 7, 5, r_range=(20, 2600), h_range=(300, 900))
    want, _ = oracle.batch(d)
    got = engine.compute(dict(d, pair_read=None, pair_hap=None))
    assert_log10_close(got, want)
    assert np.array_equal(engine.compute_regions([d])[0].ravel(), got)


def flush_regime_pairs(seed, n, r_lens=(151, 200, 300, 400, 700, 1100)):
    """Reads cut from their haplotype with so many high-quality mismatches that the fp64 likelihood lands around
    2^-1022 * 2^1020 (log10 between about -560 and -660): below, at and above the flush-to-zero threshold."""
    rng = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads, haps, quals = [], [], []
    for i in range(n):
        R = int(r_lens[i % len(r_lens)])
        H = R + int(rng.randint(0, 120))
        hap = acgt[rng.randint(0, 4, H)]
        at = int(rng.randint(0, H - R + 1))
        read = hap[at:at + R].copy()
        q = int(rng.choice([q_ for q_ in (60, 70, 80, 93) if R * q_ >= 7000] or [93]))
        per = q / 10.0 + 0.477
        k = min(R, int(round((560 + 100 * rng.rand()) / per)))          # mismatches wanted
        pos = rng.choice(R, k, replace=False)
        read[pos] = acgt[(np.searchsorted(acgt, read[pos]) + 1 + rng.randint(0, 3, k)) % 4]
        reads.append(read); haps.append(hap); quals.append(np.full(R, q, dtype=np.uint8))
    ro = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    ho = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.uint64)
    nb = int(ro[-1])
    return {"read_off": ro, "hap_off": ho, "bases": np.concatenate(reads), "qual": np.concatenate(quals),
            "ins": rng.randint(85, 94, nb).astype(np.uint8), "dele": rng.randint(85, 94, nb).astype(np.uint8),
            "gcp": np.concatenate(quals),          # gaps as dear as mismatches: no cheap way round them
 "hap_bases": np.concatenate(haps),
            "pair_read": np.arange(n, dtype=np.uint32), "pair_hap": np.arange(n, dtype=np.uint32),
            "cells": int(sum(len(r) * len(h) for r, h in zip(reads, haps)))}


def test_results_near_the_flush_to_zero_threshold(pkg, engine, oracle):
    """Flush-to-zero is on in the reference (IntelPairHmm.cc:230): a likelihood within reach of 2^-1022 depends on which
    intermediate products were flushed, so fp64 results below 1e-280 are computed again in the reference's exact
    operation order (unfused, plain form).  The outcome, including which test cases end as -inf, equals the oracle's."""
    d = flush_regime_pairs(5, 240)
    want, _ = oracle.batch(d)
    fin = np.isfinite(want)
    assert fin.any() and (~fin).any() and (want[fin] < -600).any()         # the sample straddles the threshold
    out, used, st = run(engine, d)
    assert used.all()
    assert st["n_exact"] >= int((~fin).sum()) and st["n_exact"] >= int((want[fin] < -590).sum())
    assert_log10_close(out, want)
    # forced double precision takes the same third tier
    eng64 = pkg.PairHMMEngine(0, flags=pkg.pairhmm.FORCE_DOUBLE)
    out64, _, st64 = run(eng64, d)
    eng64.close()
    assert np.array_equal(out64, out) and st64["n_exact"] == st["n_exact"]


def test_packed_kernel_is_bit_identical_to_the_scalar_kernel(pkg, oracle, synth):
    """Round 3: with MGX_PAIRHMM_PACKED_FP32 the classes with an even number of rows per lane run
    pairhmm_fwd_pk (two rows per v_pk_* instruction, the high half of a lane one column behind the low half; opt-in because
    it measured slower).  Every cell is computed by the same operations in the same order as
    in the default kernel, so the two must agree bit for bit -- over every lane-group width and row count, with 'N' in
    either sequence, reads shorter than a half-lane, haplotypes shorter than the fill/drain."""
    scalar = pkg.PairHMMEngine(0)
    packed = pkg.PairHMMEngine(0, flags=pkg.pairhmm.PACKED_FP32)
    for seed, rr, hr, n in ((1, (1, 128), (1, 256), 30000), (2, (97, 128), (200, 256), 6000), (3, (1, 40), (1, 30), 8000),
                            (4, (120, 128), (1, 12), 3000)):
        d = synth.gen_pairhmm_pairs_fast(n, 0xBEEF00 + seed, r_range=rr, h_range=hr)
        a = packed.compute(d)
        b = scalar.compute(d)
        assert np.array_equal(a, b), f"seed {seed}: {(a != b).sum()} of {n} differ"
        want, _ = oracle.batch(d)
        assert_log10_close(a, want)
    # shared reads / haplotypes (a region) and the cross-product form
    d = synth.gen_pairhmm_region(64, 40, 99, r_range=(20, 128), h_range=(30, 256))
    assert np.array_equal(packed.compute(d), scalar.compute(d))
    # a gap-continuation byte of 0 (pGAPM == 0) sends the test case to the double-precision list of the packed kernel;
    # the scalar kernel keeps it in fp32 (plain form): both within tolerance of the oracle, the other test cases identical
    d = synth.gen_pairhmm_pairs_fast(4000, 0xBEEF10, r_range=(30, 128), h_range=(40, 256))
    gcp = d["gcp"].copy()
    ro = d["read_off"].astype(np.int64)
    hit = np.zeros(4000, dtype=bool)
    for i in range(0, 4000, 37):
        gcp[ro[i] + (i % int(ro[i + 1] - ro[i]))] = 0 if i % 2 else 128
        hit[i] = True
    d2 = dict(d, gcp=gcp)
    a, ua = packed.batch(d2), None
    a.run(); av, au = a.results(with_flags=True); a.close()
    bv = scalar.compute(d2)
    want, _ = oracle.batch(d2)
    assert_log10_close(av, want); assert_log10_close(bv, want)
    pr = d2["pair_read"]
    assert au[hit[pr]].any()                                    # (classes with an odd number of rows per lane keep the scalar kernel and its plain form)
    assert np.array_equal(av[~hit[pr]], bv[~hit[pr]])
    scalar.close(); packed.close()
