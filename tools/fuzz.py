"""Randomised soak: loops over seeded random inputs for the paths and compares the device with the oracles
(bit-exact for sort/dedup and Smith-Waterman, 1e-5 for PairHMM, zlib's inflate for the BGZF blocks).  usage: fuzz.py [seconds] [seed0] [kind 0..6: that path only; 3 = Smith-Waterman]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("fast-genomic-data-processing_amd")
synth = pkg.synth
from conftest import PairHMMOracle, SortDedupOracle, SmithWatermanOracle, _ensure_oracle
from test_pairhmm_gpu import flush_regime_pairs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
ph, sd, sw = pkg.PairHMMEngine(0), pkg.SortDedupEngine(0), pkg.SmithWatermanEngine(0)
oph, osd, osw = PairHMMOracle(_ensure_oracle()), SortDedupOracle(), SmithWatermanOracle()
rng = np.random.default_rng(seed0)
t0 = time.time(); it = 0; fails = 0; counts = {"sort": 0, "pairhmm": 0, "regions": 0, "sw": 0, "shards": 0, "queue": 0, "bgzf": 0}
import zlib
bz = pkg.BgzfCompressor(0)
bam_bytes = synth.gen_bam_record_bytes(6_000_000, 99, qual_bins=(2, 11, 25, 37))
queue = pkg.PairHMMQueue(devices=(0, 0), lanes_per_device=2, depth=2, batch_pairs=700)
while time.time() - t0 < budget:
    seed = int(rng.integers(1, 2**31 - 1)); it += 1
    kind = only if only >= 0 else it % 7
    try:
        if kind == 0:
            kw = dict(n_contigs=int(rng.integers(1, 6)), contig_len=int(rng.choice([5000, 60000, 400000, 3000000])),
                      dup_rate=float(rng.choice([0.0, 0.1, 0.5])), frag_rate=float(rng.choice([0.0, 0.05, 0.3])),
                      supp_rate=float(rng.choice([0.0, 0.05])), cross_contig_rate=float(rng.choice([0.0, 0.05, 0.3])),
                      ins_range=tuple(int(x) for x in rng.choice([[200, 600], [8000, 40000], [100, 70000]])),
                      qname_style=str(rng.choice(["illumina7", "illumina6", "plain"])))
            kw["ins_range"] = (kw["ins_range"][0], min(kw["ins_range"][1], max(kw["contig_len"] // 3, kw["ins_range"][0] + 1)))
            if kw["contig_len"] <= 5000: kw["ins_range"] = (200, 600)
            os.environ["MGX_SORTDEDUP_STREAMS"] = str(rng.choice(["1", "3"]))
            n_t = int(rng.integers(1, 30000)) if it % 40 else int(rng.integers(100000, 300000))      # now and then a larger one
            raw = synth.gen_sortdedup_raw(n_t, seed, **kw)
            recs, idx, L = pkg.sortdedup.pack(raw)
            wo, wd, _ = osd.run(L, recs)
            o, d = sd.sort_mark(L, recs)
            ok = np.array_equal(o, wo) and np.array_equal(d, wd)
            counts["sort"] += 1
        elif kind == 1:
            n = int(rng.integers(1, 20000)); rmax = int(rng.choice([40, 128, 151, 300])); hmax = int(rng.choice([60, 256, 500]))
            if it % 30 == 1:      # likelihoods around the flush-to-zero threshold of fp64 (the exact tier), reads up to 1100 bases
                d = flush_regime_pairs(seed % 100000, int(rng.integers(6, 120)))
            elif it % 30 == 7:    # reads past one strip
                d = synth.gen_pairhmm_pairs(int(rng.integers(1, 40)), seed, r_range=(900, 2600), h_range=(200, 2800), random_read_rate=0.1)
            else:
                d = synth.gen_pairhmm_pairs(n, seed, r_range=(1, rmax), h_range=(1, hmax))
            want, _ = oph.batch(d)
            got = ph.compute(d)
            fin = np.isfinite(want)
            ok = np.array_equal(np.isfinite(got), fin) and (not fin.any() or float(np.abs(got[fin] - want[fin]).max()) <= 1e-5)
            counts["pairhmm"] += 1
        elif kind == 2:
            regs = [synth.gen_pairhmm_region(int(rng.integers(1, 40)), int(rng.integers(1, 20)), seed + k, r_range=(10, int(rng.choice([60, 151, 260]))),
                                             h_range=(20, 300)) for k in range(int(rng.integers(1, 12)))]
            got = ph.compute_regions(regs)
            ok = True
            for dd, g in zip(regs, got):
                d2 = dict(dd); d2["pair_read"] = None; d2["pair_hap"] = None
                ok &= bool(np.array_equal(g, ph.compute(d2).reshape(g.shape)))
            counts["regions"] += 1
        elif kind == 4:
            # one record set over 1..5 shards (router + shard upload + merge), streamed upload in random pieces
            raw = synth.gen_sortdedup_raw(int(rng.integers(1, 6000)), seed, n_contigs=int(rng.integers(1, 5)), contig_len=int(rng.choice([4000, 50000, 900000])),
                                          dup_rate=float(rng.choice([0.1, 0.4])), cross_contig_rate=float(rng.choice([0.0, 0.3])), frag_rate=float(rng.choice([0.05, 0.3])))
            recs, idx, L = pkg.sortdedup.pack(raw)
            wo, wd, _ = osd.run(L, recs)
            k_shards = int(rng.integers(1, 6))
            routed = pkg.Routed(L, recs, k_shards)
            o = np.zeros(len(recs), dtype=np.uint32); d = np.zeros(len(recs), dtype=np.uint8)
            for k in range(k_shards):
                sd.upload_shard(routed, k); sd.run()
                so, sdup = sd.results()
                routed.merge(k, so, sdup, o, d)
            routed.close()
            ok = np.array_equal(o, wo) and np.array_equal(d, wd)
            cuts = sorted(set([0, len(recs)] + [int(x) for x in rng.integers(0, len(recs) + 1, 4)]))
            sd.upload_chunks(L, (recs[a:b] for a, b in zip(cuts, cuts[1:])), n_expected=int(rng.integers(0, len(recs) + 1)))
            sd.run()
            o2, d2 = sd.results()
            ok &= bool(np.array_equal(o2, wo) and np.array_equal(d2, wd))
            counts["shards"] += 1
        elif kind == 6:
            # BGZF: ragged pieces of mixed content; every block must inflate to its piece with the right CRC and size
            nb = int(rng.integers(1, 300))
            sizes = rng.integers(0, 65281, nb); sizes[rng.integers(0, nb, max(1, nb // 8))] = rng.integers(0, 40, max(1, nb // 8))
            total = int(sizes.sum()); offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
            start = int(rng.integers(0, len(bam_bytes) - 1))
            data = np.resize(np.roll(bam_bytes, -start), total).copy() if total else np.zeros(0, dtype=np.uint8)
            for i in rng.integers(0, nb, nb // 6 + 1):      # some pieces of noise, of few symbols, of one byte value
                a_, b_ = int(offs[i]), int(offs[i + 1]); mode = int(rng.integers(0, 3))
                data[a_:b_] = rng.integers(0, 256 if mode == 0 else 4, b_ - a_, dtype=np.uint8) if mode < 2 else int(rng.integers(0, 256))
            out, oo = bz.compress(data, offs)
            raw, ob = data.tobytes(), bytes(out)
            ok = len(oo) == nb + 1 and int(oo[-1]) == len(ob)
            for i in range(nb):
                blk = ob[int(oo[i]):int(oo[i + 1])]; want = raw[int(offs[i]):int(offs[i + 1])]
                dz = zlib.decompressobj(-15)
                ok &= blk[:4] == b"\x1f\x8b\x08\x04" and int.from_bytes(blk[16:18], "little") + 1 == len(blk)
                ok &= dz.decompress(blk[18:-8]) + dz.flush() == want and dz.eof
                ok &= int.from_bytes(blk[-8:-4], "little") == zlib.crc32(want) and int.from_bytes(blk[-4:], "little") == len(want)
            counts["bgzf"] += 1
        elif kind == 5:
            # the host work queue: a stream cut into small batches over two lanes x two "devices" equals one call
            if it % 12 == 5:
                d = synth.gen_pairhmm_region(int(rng.integers(1, 60)), int(rng.integers(1, 30)), seed, r_range=(1, int(rng.choice([60, 151]))), h_range=(1, 300))
            else:
                d = synth.gen_pairhmm_pairs(int(rng.integers(1, 6000)), seed, r_range=(1, int(rng.choice([40, 128, 200]))), h_range=(1, int(rng.choice([60, 256]))))
            ok = bool(np.array_equal(queue.run(d), ph.compute(d)))
            regs = [synth.gen_pairhmm_region(int(rng.integers(1, 30)), int(rng.integers(1, 12)), seed + k, r_range=(10, 128), h_range=(20, 260)) for k in range(int(rng.integers(1, 9)))]
            got = queue.run_regions(regs)
            one = ph.compute_regions(regs)
            ok &= all(np.array_equal(a, b) for a, b in zip(got, one))
            counts["queue"] += 1
        else:
            os.environ["MGX_SW_PAIRED"] = str(rng.choice(["0", "1"]))
            os.environ["MGX_SW_I16"] = str(rng.choice(["0", "1", "1", "1"]))          # the packed 16-bit fill (round 3) or the 32-bit one for every pair
            os.environ["MGX_SW_TRANSPOSE"] = str(rng.choice(["0", "0", "1"]))
            w = synth.gen_sw_pairs(int(rng.integers(1, 400)), seed, ref_range=(1, int(rng.choice([60, 300, 700, 2048]))), alt_range=(1, int(rng.choice([40, 200, 600]))))
            params = tuple(int(x) for x in rng.choice([[25, -50, -110, -6], [3, -1, -4, -3], [1, -2, -3, -1], [10, -15, -30, -5]]))
            if rng.integers(0, 3) == 0:         # anything: the 16-bit admission rule decides pair by pair
                params = (int(rng.integers(1, 80)), -int(rng.integers(0, 160)), -int(rng.integers(0, 300)), -int(rng.integers(0, 50)))
            wc, wo, ws = osw.batch(w, params)
            gc, go, gs = sw.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"], params, want_score=True)
            ok = gc == wc and np.array_equal(go, wo) and np.array_equal(gs, ws)
            counts["sw"] += 1
    except Exception as e:            # noqa: BLE001
        ok = False; print("EXCEPTION", kind, seed, repr(e), flush=True)
    if not ok:
        fails += 1; print("MISMATCH kind", kind, "seed", seed, flush=True)
    if it % 50 == 0:
        print(f"{it} iterations, {time.time() - t0:.0f} s, failures {fails}, {counts}", flush=True)
print(f"done: {it} iterations in {time.time() - t0:.0f} s, failures {fails}, {counts}")
sys.exit(1 if fails else 0)
