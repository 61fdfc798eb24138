"""BASELINE.json configs[4] under test (run with -m gpu): the PairHMM host work queue and the sort / mark-duplicate
pipeline CO-RESIDENT on one GPU, each driven by its own host thread on its own context and streams, running at the
same time.  Each path's results must be bit-identical to its solo run on the same inputs, and both are checked
against their CPU oracles.  Reference: the worker threads of Mutect2Cpp (M2/main.cpp:254, 302-315) and the stages of
sortmardup (sortmardup/main.cpp:249-357) are separate programs there; on a node they share the GPUs."""
import threading

import numpy as np
import pytest

from test_pairhmm_oracle import assert_log10_close

pytestmark = pytest.mark.gpu


def test_pairhmm_queue_and_sort_pipeline_share_one_gpu(pkg, oracle, sd_oracle, synth):
    # PairHMM side: ragged test cases (several read-length classes, some fp64 re-runs), streamed through the queue
    d = synth.gen_pairhmm_pairs_fast(60000, 0x5EED0003, r_range=(32, 128), h_range=(64, 256))
    want_hmm, want_used = oracle.batch(d)
    # sort side: the configs[3] distribution at 2 M records, plus a small raw set with every corner case flavour
    recs, L = synth.gen_sortdedup_packed_fast(2_000_000, 0x5EED0004)
    want_order, want_dup, _ = sd_oracle.run(L, recs)

    q = pkg.PairHMMQueue(devices=(0,), lanes_per_device=3, depth=2, batch_pairs=4096)
    sd = pkg.SortDedupEngine(0)
    sd.upload(L, recs)

    # solo runs
    solo_hmm = q.run(d)
    sd.run(); solo_order, solo_dup = sd.results()
    assert_log10_close(solo_hmm, want_hmm)
    assert np.array_equal(solo_order, want_order) and np.array_equal(solo_dup, want_dup)

    # together: both threads start from one barrier and keep going until each has done `rounds` passes
    rounds = 4
    start = threading.Barrier(2)
    got_hmm, got_sort, errors = [], [], []
    busy = {"hmm": [], "sort": []}

    def hmm_thread():
        try:
            start.wait()
            import time
            for _ in range(rounds):
                t0 = time.perf_counter(); got_hmm.append(q.run(d)); busy["hmm"].append((t0, time.perf_counter()))
        except Exception as e:        # noqa: BLE001
            errors.append(e)

    def sort_thread():
        try:
            start.wait()
            import time
            for _ in range(rounds):
                t0 = time.perf_counter(); sd.run(); got_sort.append(sd.results()); busy["sort"].append((t0, time.perf_counter()))
        except Exception as e:        # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=hmm_thread), threading.Thread(target=sort_thread)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(got_hmm) == rounds and len(got_sort) == rounds
    for g in got_hmm:
        assert np.array_equal(g, solo_hmm)                      # a test case's value does not depend on what else runs
    for o, dflag in got_sort:
        assert np.array_equal(o, solo_order) and np.array_equal(dflag, solo_dup)
    # the two paths really overlapped in time (the first pass of each starts at the barrier)
    assert busy["hmm"][0][0] < busy["sort"][0][1] and busy["sort"][0][0] < busy["hmm"][0][1]
    q.close(); sd.close()


def test_corner_case_records_beside_a_resident_pairhmm_batch(pkg, engine, oracle, sd_engine, sd_oracle, synth):
    """The golden-file flavours of records (clips, cross-contig pairs, total ties) sorted while a resident PairHMM batch
    re-runs on the same device from another thread."""
    d = synth.gen_pairhmm_pairs_fast(30000, 7, r_range=(20, 128), h_range=(30, 256))
    want, _ = oracle.batch(d)
    batch = engine.batch(d)
    raw = synth.gen_sortdedup_raw(20000, 33, n_contigs=3, contig_len=40_000, dup_rate=0.4, cross_contig_rate=0.2)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    stop = threading.Event()
    outs = []

    def spin():
        while not stop.is_set():
            batch.run()
            outs.append(batch.results())
    t = threading.Thread(target=spin)
    t.start()
    try:
        for _ in range(5):
            order, dup = sd_engine.sort_mark(L, recs)
            assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
    finally:
        stop.set(); t.join()
    assert len(outs) >= 1
    for o in outs[:3] + outs[-1:]:
        assert_log10_close(o, want)
        assert np.array_equal(o, outs[0])
    batch.close()
