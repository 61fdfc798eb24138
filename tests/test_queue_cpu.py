"""Host half of the PairHMM work queue without a GPU: the packer (mgx_pairhmm_pack_batch) cuts a
stream into self-contained batches; computing every batch with the CPU oracle and concatenating must
equal the oracle on the whole stream.  The world_size-2 gloo test drives the N>1 path of bench.py:
every rank owns shard_bounds() of ONE stream and pulls its batches through the packer."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _streams(synth):
    region = synth.gen_pairhmm_region(37, 11, 5, r_range=(20, 90), h_range=(40, 120))          # shared reads and haplotypes
    cross = dict(region); cross["pair_read"] = None; cross["pair_hap"] = None                  # cross-product form
    indep = synth.gen_pairhmm_pairs(300, 9, r_range=(10, 70), h_range=(20, 90))                 # pair i = read i x hap i
    rng = np.random.RandomState(3)
    shuffled = dict(region)
    perm = rng.permutation(len(region["pair_read"]))
    shuffled["pair_read"] = region["pair_read"][perm]; shuffled["pair_hap"] = region["pair_hap"][perm]
    return {"region": region, "cross": cross, "independent": indep, "shuffled": shuffled}


@pytest.mark.parametrize("name", ["region", "cross", "independent", "shuffled"])
@pytest.mark.parametrize("batch", [1, 64, 1000])
def test_packed_batches_equal_whole_stream(pkg, synth, oracle, name, batch):
    d = _streams(synth)[name]
    full = dict(d)
    if full.get("pair_read") is None:
        nr, nh = len(d["read_off"]) - 1, len(d["hap_off"]) - 1
        full["pair_read"] = np.repeat(np.arange(nr, dtype=np.uint32), nh); full["pair_hap"] = np.tile(np.arange(nh, dtype=np.uint32), nr)
    whole, wused = oracle.batch(full, threads=1)
    n = len(whole)
    if batch == 1:
        cuts = list(range(0, 40))           # a few single-test-case batches are enough
    else:
        cuts = list(range(0, n, batch))
    for lo in cuts:
        hi = min(n, lo + batch)
        sub = pkg.pairhmm.pack_batch(d, lo, hi)
        # every referenced sequence exactly once, in first-use order
        assert sub["n_reads"] == len(np.unique(full["pair_read"][lo:hi])) and sub["n_haps"] == len(np.unique(full["pair_hap"][lo:hi]))
        first_use = full["pair_read"][lo:hi][np.sort(np.unique(sub["pair_read"], return_index=True)[1])]
        assert len(first_use) == sub["n_reads"]
        got, gused = oracle.batch(sub, threads=1)
        assert np.array_equal(got, whole[lo:hi]) and np.array_equal(gused, wused[lo:hi])


def test_pack_rejects_bad_ranges_and_indices(pkg, synth):
    d = synth.gen_pairhmm_pairs(10, 1, r_range=(5, 9), h_range=(5, 9))
    with pytest.raises(pkg.MgxError):
        pkg.pairhmm.pack_batch(d, 5, 11)
    bad = dict(d); bad["pair_read"] = d["pair_read"].copy(); bad["pair_read"][3] = 10
    with pytest.raises(pkg.MgxError):
        pkg.pairhmm.pack_batch(bad, 0, 10)


def test_fast_generator_equals_numpy_generator(synth):
    for kw in (dict(), dict(r_range=(32, 128), h_range=(64, 256)), dict(r_range=(20, 60), h_range=(30, 70), hap_n_rate=0.02)):
        a = synth.gen_pairhmm_pairs(700, 0x5EED0002, **kw)
        b = synth.gen_pairhmm_pairs_fast(700, 0x5EED0002, threads=3, **kw)
        for k in ("read_off", "hap_off", "bases", "qual", "ins", "dele", "gcp", "hap_bases", "pair_read", "pair_hap"):
            assert np.array_equal(a[k], b[k]), k
        assert a["cells"] == b["cells"] and a["alg_bytes"] == b["alg_bytes"]
    # a rank generates only its shard of the stream
    a = synth.gen_pairhmm_pairs(900, 77, r_range=(32, 128), h_range=(64, 256))
    b = synth.gen_pairhmm_pairs_fast(300, 77, r_range=(32, 128), h_range=(64, 256), first_pair=600)
    assert np.array_equal(a["bases"][int(a["read_off"][600]):], b["bases"]) and np.array_equal(a["hap_bases"][int(a["hap_off"][600]):], b["hap_bases"])


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import PairHMMOracle, _ensure_oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG)
    shard = importlib.import_module(PKG + ".shard")
    assert shard.env_rank() == (rank, rank, world)
    d = pkg.synth.gen_pairhmm_pairs(1000, 0x5EED0003, r_range=(20, 60), h_range=(40, 90))     # ONE stream, the same on every rank
    lo, hi = shard.shard_bounds(len(d["pair_read"]), rank, world)
    orc = PairHMMOracle(_ensure_oracle())
    outs = []
    for b0 in range(lo, hi, 128):               # the queue's batches of this rank's shard
        sub = pkg.pairhmm.pack_batch(d, b0, min(hi, b0 + 128))
        outs.append(orc.batch(sub, threads=1)[0])
    np.save(os.path.join(tmp, f"out{rank}.npy"), np.concatenate(outs))
    np.save(os.path.join(tmp, f"bounds{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    assert shard.max_over_ranks(1.0 + rank, dist) == float(world)
    dist.destroy_process_group()


def test_two_ranks_share_one_stream_gloo(tmp_path, oracle, synth):
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    d = synth.gen_pairhmm_pairs(1000, 0x5EED0003, r_range=(20, 60), h_range=(40, 90))
    whole, _ = oracle.batch(d, threads=1)
    parts = [np.load(tmp_path / f"out{r}.npy") for r in range(world)]
    bounds = [np.load(tmp_path / f"bounds{r}.npy") for r in range(world)]
    assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == len(whole)
    assert np.array_equal(np.concatenate(parts), whole)


def test_shard_bounds_cover(pkg):
    shard = importlib.import_module(PKG + ".shard")
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
