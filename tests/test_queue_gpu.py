"""GPU tests of the PairHMM host work queue (BASELINE.json configs[2]): a long stream cut into batches
that lanes pull off an atomic index must give exactly the values of one call on the whole stream."""
import numpy as np
import pytest

from test_pairhmm_oracle import assert_log10_close
from test_queue_cpu import _streams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["region", "cross", "independent", "shuffled"])
def test_queue_equals_single_call(pkg, engine, synth, name):
    d = _streams(synth)[name]
    want = engine.compute(d)
    q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=3, depth=2, batch_pairs=50)
    got, used = q.run(d, with_flags=True)
    st = q.stats()
    q.close()
    assert np.array_equal(got, want)
    assert st["n_batches"] == (len(want) + 49) // 50 and st["n_pairs"] == len(want) and sum(st["batches_per_device"]) == st["n_batches"]


def test_queue_full_batches_vs_oracle_and_ranges(pkg, engine, oracle, synth):
    d = synth.gen_pairhmm_pairs_fast(40000, 0x5EED0003, r_range=(32, 128), h_range=(64, 256))
    want, wused = oracle.batch(d)
    # two "devices" (the same ordinal twice: the multi-device code path on a one-GPU box), default batch size cut down
    q = pkg.PairHMMQueue(devices=(0, 0), lanes_per_device=2, depth=3, batch_pairs=4096)
    got, used = q.run(d, with_flags=True)
    assert_log10_close(got, want)
    assert (used != wused).sum() <= 20
    assert np.array_equal(got, engine.compute(d))                 # a test case's value does not depend on its batch
    st = q.stats()
    assert st["n_batches"] == 10 and st["n_lanes"] == 4 and st["batches_per_device"][0] + st["batches_per_device"][1] == 10
    assert st["bytes_h2d"] >= d["alg_bytes"] - 4 * 40000
    # the shard of one rank: a sub-range of the same stream
    part = q.run(d, lo=12345, hi=31000)
    assert np.array_equal(part, got[12345:31000])
    assert len(q.run(d, lo=7, hi=7)) == 0
    with pytest.raises(pkg.MgxError):
        q.run(d, lo=5, hi=40001)
    q.close()


def test_queue_reports_bad_input(pkg, synth):
    d = synth.gen_pairhmm_pairs(500, 3, r_range=(10, 40), h_range=(20, 60))
    bad = dict(d); bad["pair_hap"] = d["pair_hap"].copy(); bad["pair_hap"][321] = 500
    q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=2, batch_pairs=100)
    with pytest.raises(pkg.MgxError, match="321"):
        q.run(bad)
    assert np.array_equal(q.run(d), pkg.PairHMMEngine(0).compute(d))      # the queue is usable after an error
    q.close()


def test_region_calls_on_a_timing_context(pkg, oracle, synth):
    """region()/regions() on a context created with MGX_PAIRHMM_TIMING used to index an empty event
    vector; events are now allocated by the first timed run of any batch."""
    eng = pkg.PairHMMEngine(0, flags=pkg.pairhmm.TIMING)
    plain = pkg.PairHMMEngine(0)
    d = synth.gen_pairhmm_region(20, 6, 11, r_range=(20, 80), h_range=(40, 100))
    mq = np.full(20, 60, dtype=np.uint8)
    a, ka = eng.region(d, mq)
    b, kb = plain.region(d, mq)
    assert np.array_equal(a, b) and np.array_equal(ka, kb)
    ra = eng.regions([d, d], [mq, mq]); rb = plain.regions([d, d], [mq, mq])
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(ra, rb))
    assert np.array_equal(eng.compute_regions([d])[0], plain.compute_regions([d])[0])
    # every run of a timed batch is measured; stats average them
    bt = eng.batch(d)
    for _ in range(5):
        bt.run()
    st = bt.stats()
    assert st["n_runs_timed"] == 5 and st["ms_f32_dominant"] > 0
    bt.run()
    assert bt.stats()["n_runs_timed"] == 1
    bt.close(); eng.close(); plain.close()


def test_pair_count_without_pair_arrays_is_rejected(pkg, synth):
    """n_pairs > 0 with NULL pair arrays and no reads must be -EINVAL, not a NULL dereference."""
    import ctypes as C
    d = synth.gen_pairhmm_pairs(4, 1, r_range=(5, 9), h_range=(5, 9))
    inp, keep = pkg.pairhmm.make_input(d)
    inp.pair_read = None; inp.pair_hap = None; inp.n_reads = 0; inp.n_pairs = 4
    eng = pkg.PairHMMEngine(0)
    out = np.zeros(4)
    rc = eng.lib.mgx_pairhmm_compute(eng.ctx, C.byref(inp), out.ctypes.data_as(C.c_void_p))
    assert rc == -22
    eng.close()


def test_regions_through_the_queue(pkg, engine, synth):
    """Row F1 through the queue: runs of whole regions pulled by the lanes; values identical to one call per region."""
    regions = [synth.gen_pairhmm_region(5 + (g * 7) % 40, 1 + (g * 3) % 25, 100 + g, r_range=(20, 128), h_range=(64, 256)) for g in range(60)]
    want = [engine.compute(dict(r, pair_read=None, pair_hap=None)).reshape(len(r["read_off"]) - 1, len(r["hap_off"]) - 1) for r in regions]
    q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=3, depth=2, batch_pairs=3000)
    got = q.run_regions(regions)
    st = q.stats()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert st["n_batches"] > 3 and st["n_pairs"] == sum(w.size for w in want)
    one = engine.compute_regions(regions)
    assert all(np.array_equal(a, b) for a, b in zip(one, want))
    # a region larger than batch_pairs is a batch of its own; empty lists are fine
    big = [synth.gen_pairhmm_region(90, 40, 7, r_range=(20, 60), h_range=(40, 90))]
    assert np.array_equal(q.run_regions(big)[0], engine.compute_regions(big)[0])
    assert q.run_regions([]) == []
    q.close()
