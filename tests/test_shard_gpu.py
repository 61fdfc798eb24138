"""One record set over several shards on the device (run with -m gpu): every shard goes through
mgx_sortdedup_upload_shard + run + results on one card, one after the other (what N processes do on N
cards), and the merged order and duplicate flags must be bit-identical to the single-shard run."""
import numpy as np
import pytest

from test_shard_cpu import boundary_case, sharded
from test_sortdedup_oracle import load_golden

pytestmark = pytest.mark.gpu


def device_shard(engine):
    def run(routed, k, sh):
        engine.upload_shard(routed, k)
        engine.run()
        return engine.results()
    return run


@pytest.mark.parametrize("k_shards", [1, 2, 3, 4])
def test_small_cases_vs_oracle(pkg, sd_engine, sd_oracle, synth, k_shards):
    cases = [load_golden()[0], boundary_case(synth), synth.gen_sortdedup_raw(3000, 21, n_contigs=3, contig_len=30_000, dup_rate=0.3, cross_contig_rate=0.2),
             synth.gen_sortdedup_raw(1500, 22, qname_style="plain", contig_len=5_000)]
    for raw in cases:
        recs, idx, L = pkg.sortdedup.pack(raw)
        want_order, want_dup, _ = sd_oracle.run(L, recs)
        order, dup, _ = sharded(pkg, device_shard(sd_engine), L, recs, k_shards)
        assert np.array_equal(order, want_order)
        assert np.array_equal(dup, want_dup)


def test_the_routed_mark_reaches_the_device_bitmap(pkg, sd_engine, sd_oracle, synth):
    raw = boundary_case(synth)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order, dup, info = sharded(pkg, device_shard(sd_engine), L, recs, 2)
    assert len(info[1]["marks"]) == 2
    assert np.array_equal(dup, want_dup) and np.array_equal(order, want_order) and dup[4] == 1 and dup[6] == 0


@pytest.mark.parametrize("k_shards", [2, 4])
def test_eight_million_records_equal_the_single_shard_run(pkg, sd_engine, synth, k_shards):
    recs, L = synth.gen_sortdedup_packed(8_000_000, 0x5EED0004)
    # some pairs far apart (second end in another shard) and fragments sitting on pair ends
    rng = np.random.RandomState(1)
    far = rng.choice(len(recs) // 2 - 200_000, 20000, replace=False) * 2
    recs["prime5"][far + 1] = (recs["prime5"][far + 1] + np.uint64(L // 2)) % np.uint64(L - 1000)
    recs["coord"][far + 1] = recs["prime5"][far + 1]
    frag = np.nonzero(recs["mate"] == 0xFFFFFFFF)[0][::2][:50000]
    src = rng.choice(far, len(frag)) + 1
    recs["prime5"][frag] = recs["prime5"][src]; recs["coord"][frag] = recs["coord"][src]
    recs["flag"][frag] = (recs["flag"][frag] & ~np.uint16(16)) | (recs["flag"][src] & np.uint16(16))
    want_order, want_dup = sd_engine.sort_mark(L, recs)
    order, dup, info = sharded(pkg, device_shard(sd_engine), L, recs, k_shards)
    assert sum(len(s["marks"]) for s in info) > 5000
    assert np.array_equal(order, want_order)
    assert np.array_equal(dup, want_dup)
    # the single-shard call still works on the same context afterwards
    o2, d2 = sd_engine.sort_mark(L, recs)
    assert np.array_equal(o2, want_order) and np.array_equal(d2, want_dup)


def test_empty_halves(pkg, sd_engine, sd_oracle, synth):
    """Everything in one corner of the genome: the other shards order and mark nothing."""
    raw = synth.gen_sortdedup_raw(300, 4, n_contigs=1, contig_len=9_000)
    raw["target_len"] = np.array([9_000, 500_000], dtype=np.uint64); raw["n_targets"] = 2
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order, dup, info = sharded(pkg, device_shard(sd_engine), L, recs, 4)
    assert len(info[1]["mark_recs"]) == 0 and len(info[2]["order_coord"]) == 0
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
