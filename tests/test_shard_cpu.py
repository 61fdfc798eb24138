"""One record set over several shards, without a GPU: the product's host router (mgx_sortdedup_route) and merge
around the CPU oracle standing in for the per-shard device pipeline must reproduce the single-shard result bit
for bit -- order and duplicate flags -- for every shard count, including pairs whose second end (and bitmap
mark) lands in another shard.  Reference: sortmardup/tbb/range_partitioner.h:98-100, main.cpp:160-192."""
import numpy as np
import pytest

from test_sortdedup_oracle import load_golden


def sharded(pkg, run_shard, L, recs, k_shards, only=None):
    routed = pkg.Routed(L, recs, k_shards) if only is None else pkg.Routed(L, recs, k_shards, only_shard=only)
    order = np.zeros(len(recs), dtype=np.uint32)
    dup = np.zeros(len(recs), dtype=np.uint8)
    info = []
    for k in (range(k_shards) if only is None else [only]):
        sh = routed.shard_arrays(k)
        o, d = run_shard(routed, k, sh)
        routed.merge(k, o, d, order, dup)
        info.append(sh)
    routed.close()
    return order, dup, info


def boundary_case(synth):
    """Hand-made records around the middle of a 2-shard genome: a pair that straddles the boundary, a fragment on
    the far side that collides with the straddling pair's second end (duplicate only because of the routed mark),
    a fragment that collides with nothing, and duplicate pairs on either side."""
    rr = synth.RawRecords([1000, 1000])          # L = 2000 -> 2 shards of width 1000 (positions 0..999 | 1000..)
    q = np.full(50, 30, dtype=np.uint8)
    lo = np.full(50, 20, dtype=np.uint8)

    def pair(name, tid1, pos1, tid2, pos2, qual=q, rev2=True):
        rr.add(name, 1 | 2 | 64 | (32 if rev2 else 0), tid1, pos1, "50M", qual)
        rr.add(name, 1 | 2 | 128 | (16 if rev2 else 0), tid2, pos2, "50M", qual)
    pair("SYN:1:FC:1:1:1:1", 0, 960, 1, 10)               # record 1 at 960 (shard 0), record 2 reverse: 5' end 1000+10+49 (shard 1)
    pair("SYN:1:FC:1:1:1:2", 0, 960, 1, 10, qual=lo)      # its duplicate (lower score)
    rr.add("SYN:1:FC:1:1:1:3", 1 | 8 | 64 | 16, 1, 10, "50M", q)     # fragment, reverse, 5' end 1059: collides with the mark
    rr.add("SYN:1:FC:1:1:1:3", 1 | 4 | 128, 1, 10, "", q)
    rr.add("SYN:1:FC:1:1:1:4", 1 | 8 | 64, 1, 10, "50M", q)          # fragment, forward at 1010: no pair end there
    rr.add("SYN:1:FC:1:1:1:4", 1 | 4 | 128, 1, 10, "", q)
    pair("SYN:1:FC:1:1:1:5", 1, 300, 1, 500)              # both ends in shard 1
    pair("SYN:1:FC:1:1:1:6", 1, 300, 1, 500, qual=lo)
    rr.add("SYN:1:FC:1:1:1:7", 1 | 8 | 64, 0, 960, "50M", q)         # fragment at the straddling pair's first end (same shard)
    rr.add("SYN:1:FC:1:1:1:7", 1 | 4 | 128, 0, 960, "", q)
    rr.add("SYN:1:FC:1:1:1:8", 77, -1, -1, "", q); rr.add("SYN:1:FC:1:1:1:8", 141, -1, -1, "", q)   # unmapped pair: coordinate L
    return rr.arrays()


@pytest.mark.parametrize("k_shards", [1, 2, 3, 4, 7])
def test_router_and_merge_reproduce_single_shard(pkg, sd_oracle, synth, k_shards):
    cases = [load_golden()[0], boundary_case(synth), synth.gen_sortdedup_raw(3000, 21, n_contigs=3, contig_len=30_000, dup_rate=0.3, cross_contig_rate=0.2),
             synth.gen_sortdedup_raw(1500, 22, qname_style="plain", contig_len=5_000)]
    for raw in cases:
        recs, idx, L = pkg.sortdedup.pack(raw)
        want_order, want_dup, _ = sd_oracle.run(L, recs)
        order, dup, info = sharded(pkg, lambda r, k, sh: sd_oracle.run_shard(L, sh), L, recs, k_shards)
        assert np.array_equal(order, want_order)
        assert np.array_equal(dup, want_dup)
        # the halves partition the input: every record ordered once, every non-ignorable record marked once
        assert sum(len(s["order_coord"]) for s in info) == len(recs)
        ign = (recs["flag"] & (0x4 | 0x100 | 0x800)) != 0
        assert sum(len(s["mark_recs"]) for s in info) == int((~ign).sum())
        for s in info:
            assert ((s["order_coord"] >= s["coord_lo"]) & (s["order_coord"] < np.uint64(min(s["coord_hi"], 2**63)))).all() or s["coord_hi"] > 2**63
            assert (np.diff(s["order_arrival"].astype(np.int64)) > 0).all()         # arrival order kept


def test_boundary_marks_are_what_makes_the_fragment_a_duplicate(pkg, sd_oracle, synth):
    raw = boundary_case(synth)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    assert want_dup[4] == 1 and want_dup[6] == 0 and want_dup[2] == 1 and want_dup[3] == 1      # fragment :3 collides, :4 does not
    order, dup, info = sharded(pkg, lambda r, k, sh: sd_oracle.run_shard(L, sh), L, recs, 2)
    assert np.array_equal(dup, want_dup) and np.array_equal(order, want_order)
    assert len(info[1]["marks"]) == 2 and len(info[0]["marks"]) == 0           # the two straddling pairs' second ends
    assert set((info[1]["marks"] >> np.uint64(1)).tolist()) == {1059} and (info[1]["marks"] & np.uint64(1)).all()
    # without the routed marks shard 1 would miss the duplicate
    routed = pkg.Routed(L, recs, 2)
    sh = routed.shard_arrays(1); sh["marks"] = sh["marks"][:0]
    o, d = sd_oracle.run_shard(L, sh)
    got = np.zeros(len(recs), dtype=np.uint8)
    routed.merge(1, o, d, np.zeros(len(recs), dtype=np.uint32), got)
    assert got[4] == 0
    routed.close()


def test_a_rank_materialises_only_its_shard(pkg, sd_oracle, synth):
    raw = synth.gen_sortdedup_raw(2000, 5, n_contigs=4, contig_len=20_000)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    order = np.zeros(len(recs), dtype=np.uint32); dup = np.zeros(len(recs), dtype=np.uint8)
    for rank in range(4):                     # what 4 processes do, one after the other
        o, d, info = sharded(pkg, lambda r, k, sh: sd_oracle.run_shard(L, sh), L, recs, 4, only=rank)
        sl = slice(info[0]["order_base"], info[0]["order_base"] + len(info[0]["order_coord"]))
        order[sl] = o[sl]; dup |= d
    assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)
    r = pkg.Routed(L, recs, 4, only_shard=2)
    with pytest.raises(pkg.MgxError):
        r.merge(1, np.zeros(1, np.uint32), np.zeros(1, np.uint8), order, dup)
    r.close()


def test_route_rejects_bad_input(pkg, synth):
    recs, L = synth.gen_sortdedup_packed(1000, 3, n_contigs=2, contig_len=100_000)
    with pytest.raises(pkg.MgxError):
        pkg.Routed(L, recs, 0)
    bad = recs.copy(); bad["mate"][10] = 5000
    with pytest.raises(pkg.MgxError):
        pkg.Routed(L, bad, 2)


def test_route_degenerate_inputs(pkg, sd_oracle, synth):
    """No records at all, a single fragment, more shards than records."""
    empty = np.zeros(0, dtype=synth.REC_DTYPE)
    r = pkg.Routed(1000, empty, 3)
    for k in range(3):
        sh = r.shard_arrays(k)
        assert len(sh["order_coord"]) == len(sh["mark_recs"]) == len(sh["marks"]) == 0
    r.close()
    raw = synth.gen_sortdedup_raw(3, 1, n_contigs=1, contig_len=5000)
    recs, idx, L = pkg.sortdedup.pack(raw)
    want_order, want_dup, _ = sd_oracle.run(L, recs)
    for k_shards in (1, 5, 64):
        order, dup, info = sharded(pkg, lambda rr, k, sh: sd_oracle.run_shard(L, sh), L, recs, k_shards)
        assert np.array_equal(order, want_order) and np.array_equal(dup, want_dup)


def _rank_cases(pkg, synth):
    """The record sets of the multi-process test, the same on every rank (seeded / hand-made)."""
    return [load_golden()[0], boundary_case(synth),
            synth.gen_sortdedup_raw(2500, 31, n_contigs=3, contig_len=30_000, dup_rate=0.3, cross_contig_rate=0.2)]


def _route_worker(rank, world, port, tmp):
    """One process per shard, as bench.py --gpus N runs them: the rank routes the WHOLE record set with
    only_shard = rank (sortmardup/tbb/range_partitioner.h:98-100: partition = key / range), runs the per-shard
    pipeline on its own shard (the CPU oracle standing in for the device) and merges its piece into full-size
    arrays; rank 0 collects the pieces -- a sum of disjoint order slices and an OR of the flags, which is all
    mgx_sortdedup_merge leaves to do across processes."""
    import importlib
    import os
    import sys
    from conftest import PKG, ROOT
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import SortDedupOracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG)
    orc = SortDedupOracle()
    for c, raw in enumerate(_rank_cases(pkg, pkg.synth)):
        recs, idx, L = pkg.sortdedup.pack(raw)
        routed = pkg.Routed(L, recs, world, only_shard=rank)
        sh = routed.shard_arrays(rank)
        o, d = orc.run_shard(L, sh)
        order = np.zeros(len(recs), dtype=np.uint32); dup = np.zeros(len(recs), dtype=np.uint8)
        routed.merge(rank, o, d, order, dup)
        # a rank holds nothing of the other shards
        other = routed.shard((rank + 1) % world)
        assert not other.order_coord and not other.mark_recs and not other.marks
        routed.close()
        n_order = torch.tensor([len(sh["order_coord"]), len(sh["marks"])], dtype=torch.int64)
        t_order = torch.from_numpy(order.astype(np.int64)); t_dup = torch.from_numpy(dup.astype(np.int64))
        dist.reduce(t_order, dst=0, op=dist.ReduceOp.SUM)
        dist.reduce(t_dup, dst=0, op=dist.ReduceOp.MAX)
        dist.reduce(n_order, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            np.savez(os.path.join(tmp, f"case{c}.npz"), order=t_order.numpy().astype(np.uint32), dup=t_dup.numpy().astype(np.uint8),
                     n_order=n_order.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_route_their_own_shard_gloo(tmp_path, pkg, sd_oracle, synth):
    """the N > 1 path of the sortmardup leg with one PROCESS per shard (world size 2, gloo)."""
    import os
    import torch.multiprocessing as mp
    world = 2
    port = 31500 + os.getpid() % 2000
    mp.spawn(_route_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n_marks = []
    for c, raw in enumerate(_rank_cases(pkg, synth)):
        recs, idx, L = pkg.sortdedup.pack(raw)
        want_order, want_dup, _ = sd_oracle.run(L, recs)
        got = np.load(tmp_path / f"case{c}.npz")
        assert int(got["n_order"][0]) == len(recs)                    # every record ordered by exactly one rank
        assert np.array_equal(got["order"], want_order), f"case {c}: order"
        assert np.array_equal(got["dup"], want_dup), f"case {c}: duplicate flags"
        n_marks.append(int(got["n_order"][1]))
    assert n_marks[1] == 2                                            # the boundary case's two routed marks crossed ranks
