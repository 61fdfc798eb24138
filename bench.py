"""Headline benchmark (driver contract: one process per GPU, rank 0 prints ONE JSON line).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

PairHMM (`value`, GCUPS):
  N = 1  BASELINE.json configs[1]: 1M synthetic test cases, read 128 x haplotype 256 (sub-run 2a), resident in HBM.
  N > 1  BASELINE.json configs[2]: ONE stream of 64M test cases split over the N ranks (64M / N per GPU, each rank
         generates only its shard), every shard resident in HBM; no collective on the data path -- torch.distributed
         only provides the barrier and the max-over-ranks of the timed region.
  A "step" is one pass of the hot path over the resident shard: the fp32 recurrence kernel over all test cases plus
  the fp64 re-run of those that underflowed, results left in HBM.  Inputs are uploaded before the timed region.
  Next to it: "queue" = the same shard streamed from host memory through the host work queue (65536-test-case
  batches, packing/H2D of batch k+1 under the kernels of batch k; PCIe-inclusive, never `value`), "ragged" = sub-run 2b.
sortmardup ("sortmardup", Mrecords/s): BASELINE.json configs[3], ONE 200M-record data set; at N > 1 it is routed into N
  coordinate shards (records by coordinate, templates by record-1 5' end, indicator marks to the owning shard).
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: see native.load()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "fast-genomic-data-processing_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
VALU_INT32_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # one int32 lane-operation per lane per clock: 78.6 T lane-op/s
SW_INSTR_PER_CELL = 12.7       # k_sw_fill16 as compiled: 279 VALU instructions per step of 11 rows x 2 pairs


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def host_threads(world):
    """Host threads one rank may use: the ranks of a node share its cores."""
    return max(2, min(32, host_cores() // max(world, 1)))


def measured_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction, plus WRITE_SIZE)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return t.get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(synth, n_sample, seed):
    """Times the CPU checker on a bounded sample of the same workload, on this box's host cores.
    kind = "reference": the reference's own AVX kernels (oracle/_ref, prebuilt in the build
    container); kind = "port": this repo's scalar restatement (oracle/pairhmm_oracle.c)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import PairHMMOracle, _ensure_oracle
    d = synth.gen_pairhmm_pairs(n_sample, seed)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_pairhmm.so")
    if os.path.exists(ref_so):
        orc, kind, sample_mul = PairHMMOracle(ref_so, "ref_pairhmm_batch"), "reference", 1
    else:
        orc, kind, sample_mul = PairHMMOracle(_ensure_oracle()), "port", 8
        n_sample //= sample_mul
        d = synth.gen_pairhmm_pairs(n_sample, seed)
    cores = host_cores()
    orc.batch(synth.gen_pairhmm_pairs(256, seed), threads=cores)   # warm the tables / threads
    t0 = time.perf_counter()
    orc.batch(d, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": d["cells"] / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": kind,
            "sample": f"first {n_sample} test cases of the workload (R=128, H=256), "
                      f"{d['cells'] / 1e9:.2f} Gcells in {dt:.2f} s wall on {cores} threads"}


def sort_cpu_baseline(pkg, L, recs):
    """The CPU checker (oracle/sortdedup_oracle.c, a qsort-based restatement; the reference's TBB main.cpp is unlinkable,
    DESIGN.md 2) on ALL host cores, the way the reference uses them: the sample is range-partitioned by key
    (sortmardup/tbb/range_partitioner.h:98-100; here the product's host router, one coordinate range per core) and the
    partitions are sorted and searched for duplicates in parallel (sortmardup/main.cpp:249-357, tbb::parallel_for over
    partitions).  Timed: the parallel per-partition work; the partitioning itself is reported beside it."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import SortDedupOracle
    cores = host_cores()
    ns = min(len(recs), 32_000_000)
    ns -= ns % 2
    orc = SortDedupOracle()
    t0 = time.perf_counter()
    routed = pkg.Routed(L, recs[:ns], cores)
    shards = [routed.shard_arrays(k) for k in range(cores)]
    route_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    bitmap = np.zeros((4 * L >> 6) + 2, dtype=np.uint64)      # the reference's ONE 4L-bit indicator (main.cpp:115), pages touched on demand
    with ThreadPoolExecutor(cores) as ex:                     # ctypes releases the GIL around the C call
        res = list(ex.map(lambda sh: orc.run_shard(L, sh, bitmap), shards))
    cdt = time.perf_counter() - t0
    routed.close()
    return {"value": ns / cdt / 1e6, "unit": "Mrecords/s", "cores": cores, "kind": "port",
            "sample": f"first {ns} records of the set, range-partitioned into {cores} coordinate shards ({route_s:.2f} s, not timed) and "
                      f"sorted + searched for duplicates on {cores} threads by the qsort-based restatement, {cdt:.2f} s wall",
            "records_ordered": int(sum(len(o) for o, _ in res))}


def sortmardup_leg(pkg, synth, args, rank, local_rank, world, dist, torch, backend="nccl"):
    """GPU radix sorts + duplicate marking of BASELINE.json configs[3], records resident in HBM.  N = 1: the whole
    200 M-record set on one GPU.  N > 1: the SAME set (every rank generates it from the same seed) routed into N
    coordinate shards by the host router -- records by coordinate, templates by their smaller 5' end, bitmap marks
    to the shard that owns the position -- and rank r runs shard r; no collective on the data path."""
    recs, L = synth.gen_sortdedup_packed_fast(args.sort_records, 0x5EED0004, threads=host_threads(world))
    eng = pkg.SortDedupEngine(local_rank)
    route_s, shard_info = 0.0, None
    t0 = time.perf_counter()
    if world == 1:
        eng.upload(L, recs)                # first use: allocates the device buffers and the pinned staging
        t0 = time.perf_counter()
        eng.upload(L, recs)                # the upload that is timed: packed host records -> HBM, steady state
    else:
        routed = pkg.Routed(L, recs, world, only_shard=rank)
        route_s = time.perf_counter() - t0
        sh = routed.shard(rank)
        shard_info = {"n_order": int(sh.n_order), "n_mark": int(sh.n_mark), "n_marks_routed_in": int(sh.n_marks),
                      "coord_lo": int(sh.coord_lo), "coord_hi": min(int(sh.coord_hi), L + 1)}
        t0 = time.perf_counter()
        eng.upload_shard(routed, rank)
    upload_s = time.perf_counter() - t0
    for _ in range(2):
        eng.run()
    eng.stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(local_rank)
    t0 = time.perf_counter()
    for _ in range(args.sort_steps):
        eng.run()
    st = eng.stats()                      # waits for the stream
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(local_rank)
    dt = time.perf_counter() - t0
    tmax = dt
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
    if world > 1:
        # outside the timed region: this rank's slice of the global order is sorted and lies in its coordinate range
        order, dup = eng.results()
        cs = recs["coord"][order]
        assert (np.diff(cs.astype(np.int64)) >= 0).all() and (len(cs) == 0 or (cs[0] >= shard_info["coord_lo"] and cs[-1] < max(shard_info["coord_hi"], L + 1)))
        shard_info["dup_marked"] = int(dup.sum())
        routed.close()
    out = None
    if rank == 0:
        n = len(recs)
        # dominant kernel of this leg: the full-width scatter launches of the record (coordinate) sort.
        # In the timed runs the three sorts overlap on three streams, so a launch shares the device with
        # other kernels; its duration ALONE comes from two extra, untimed runs with the sorts serialised.
        ms_overlapped = st["ms_scatter_records"] / max(st["n_scatter_records"], 1)
        os.environ["MGX_SORTDEDUP_STREAMS"] = "1"
        eng.run(); eng.run()
        st1 = eng.stats()
        os.environ.pop("MGX_SORTDEDUP_STREAMS")
        n_dom = max(st1["n_scatter_records"], 1)
        ms_scatter_avg = st1["ms_scatter_records"] / n_dom
        bytes_per_scatter = st1["scatter_records_bytes"] / n_dom
        achieved = st1["scatter_records_bytes"] / max(st1["ms_scatter_records"], 1e-9) / 1e6
        out = {"metric": "sortmardup Mrecords/s", "value": n * args.sort_steps / tmax / 1e6,
               "unit": "Mrecords/s", "n_gpus": world, "steps": args.sort_steps, "ms_per_step": tmax / args.sort_steps * 1e3,
               "dtype": "u64", "scaling": "strong",
               "config": {"workload": f"BASELINE.json configs[3]: ONE set of {n} synthetic packed BAM records, 97% in proper "
                                      "pairs, 10% duplicate pairs, L=3.1e9; radix sorts + duplicate search, records resident in HBM"
                                      + ("" if world == 1 else f"; routed into {world} coordinate shards by the host router, one per GPU "
                                                                "(rank 0's shard described below)"),
                          "records_total": n, "n_double": st["n_double"], "n_single": st["n_single"],
                          "dup_records": st["n_dup_records"], "radix_passes": st["n_radix_passes"], "shard": shard_info},
               "device_ms": st["ms_total"], "upload_s": upload_s, "route_s": route_s,
               "pcie_inclusive_mrecords_s": (n if world == 1 else shard_info["n_order"]) / (upload_s + st["ms_total"] * 1e-3) / 1e6,
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS,
                            "traffic": measured_traffic("k_radix_scatter<false, false, 4, false>") if n == 200_000_000 and world == 1 else None,
                            "traffic_source": "offline rocprofv3 --pmc passes (profiles/pmc_traffic.json), not measured in this run",
                            "kernel": "k_radix_scatter<false, false, 4, false> (record sort on packed coord<<32|arrival words, "
                                      f"{st1['n_scatter_records']} full-width launches per run)", "kernel_ms": ms_scatter_avg,
                            "kernel_ms_overlapped": ms_overlapped,
                            "all_scatter_launches": {"n": st1["n_radix_passes"], "ms": st1["ms_radix_scatter"],
                                                     "GBps": st1["radix_scatter_bytes"] / max(st1["ms_radix_scatter"], 1e-9) / 1e6,
                                                     "device_ms_serialised": st1["ms_total"]},
                            "alg_bytes_per_launch": bytes_per_scatter,
                            "note": "algorithmic bytes = keys+payload read once and written once per pass; kernel_ms is the "
                                    "launch alone (sorts serialised, MGX_SORTDEDUP_STREAMS=1), kernel_ms_overlapped its "
                                    "duration inside the timed three-stream runs"},
               "model_roofline": {"alg_bytes": st["alg_bytes"], "achieved": st["alg_bytes"] / (st["ms_total"] * 1e-3) / 1e9,
                                  "unit": "GB/s", "frac": st["alg_bytes"] / (st["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "A MODEL, not a measured bandwidth: the bytes the reference-shaped LSD-8 pipeline of SURVEY.md 8d would move "
                                          "(307.5 B/record at this config) divided by this pipeline's time -- it runs fewer and narrower passes than the "
                                          "model (DESIGN.md 4.2), so the figure can exceed what a copy kernel reaches; the measured roof is `roofline`"}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = sort_cpu_baseline(pkg, L, recs)
    eng.close()
    return out



def smithwaterman_leg(pkg, synth, args, rank, local_rank):
    """Row F4 (widening, not part of BASELINE.json's metric): reads against their best haplotype, the
    batch Mutect2Cpp realigns after PairHMM.  Rank 0 only; reported next to the headline numbers."""
    if rank != 0:
        return None
    n = args.sw_pairs
    w = synth.gen_sw_pairs(n, 0x5EED0020, ref_range=(250, 400), alt_range=(100, 151), strategies=(9,))
    eng = pkg.SmithWatermanEngine(local_rank)
    eng.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"])
    fills, traces = [], []
    for _ in range(3):
        cig, off = eng.align_batch(w["ref_off"], w["ref"], w["alt_off"], w["alt"], w["strategy"])
        st = eng.stats(); fills.append(st["ms_fill"]); traces.append(st["ms_trace"])
    ms_fill, ms_trace = float(np.median(fills)), float(np.median(traces))
    out = {"metric": "Smith-Waterman GCUPS (matrix fill + back-trace on the device, inputs resident)",
           "value": st["cells"] / (ms_fill + ms_trace) / 1e6, "unit": "GCUPS", "dtype": "i16 (packed; i32 for pairs whose scores may not fit)",
           "pairs_on_the_16_bit_kernel": st["n_pairs_i16"],
           "config": {"workload": "synthetic reads (100-151 bases) against haplotype windows (250-400 bases), STANDARD_NGS "
                                  "parameters, SOFTCLIP", "pairs": n, "cells": st["cells"]},
           "ms_fill": ms_fill, "ms_trace": ms_trace, "backtrace_bytes": st["backtrace_bytes"],
           "roofline": {"bound": "valu", "achieved": st["cells"] * SW_INSTR_PER_CELL / (ms_fill * 1e-3) / 1e12, "peak": VALU_INT32_PEAK_TOPS,
                        "unit": "T lane-op/s", "frac": st["cells"] * SW_INSTR_PER_CELL / (ms_fill * 1e-3) / 1e12 / VALU_INT32_PEAK_TOPS, "traffic": None,
                        "kernel": "k_sw_fill16<32,false>", "kernel_ms": ms_fill,
                        "hbm": {"achieved": 0.5 * st["cells"] / (ms_fill * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": 0.5 * st["cells"] / (ms_fill * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "half an algorithmic byte per cell (the back-trace nibble)",
                                "written": st["backtrace_bytes"]},
                        "note": "the fill is integer-VALU bound: 12.7 instructions per cell as compiled (25 packed 16-bit instructions per two cells, "
                                "DESIGN.md 4b) against 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-op/s (the packed instructions issue at half that rate: "
                                "SQ_INSTS_VALU 1.97e8 per launch = 3.2 cycles per instruction, profiles/pmc_traffic.json); padding rows, fill/drain steps and "
                                "the longer partner of a lane group keep ~60 % of the cell slots busy at this shape; the trace is a latency chain"}}
    if not args.no_cpu_baseline:
        so = os.path.join(ROOT, "oracle", "_ref", "libref_smithwaterman.so")
        if os.path.exists(so):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import SmithWatermanRef
            ref = SmithWatermanRef(so)
            m = n
            sub = dict(ref_off=w["ref_off"][:m + 1], ref=w["ref"], alt_off=w["alt_off"][:m + 1], alt=w["alt"], strategy=w["strategy"][:m])
            dts = []
            for _ in range(5):                      # the first call pays the page faults of its per-call 4 MB matrices
                t0 = time.perf_counter()
                rc, ro = ref.batch(sub, (25, -50, -110, -6), 2 * 400 + 1)
                dts.append(time.perf_counter() - t0)
            dt = float(np.median(dts))
            cells = float((np.diff(sub["ref_off"].astype(np.int64)) * np.diff(sub["alt_off"].astype(np.int64))).sum())
            lens = (rc != 0).sum(axis=1)
            same = all(rc[p, :lens[p]].tobytes() == cig[p] for p in range(m)) and bool(np.array_equal(ro, off[:m]))
            out["cpu_baseline"] = {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": host_cores(), "kind": "reference",
                                   "sample": f"the same {m} pairs, reference AVX2 aligner (one call per pair, OpenMP over pairs), median of 5 runs, {dt:.3f} s",
                                   "identical_to_device": same}
    eng.close()
    return out


def bgzf_leg(pkg, synth, args, rank, local_rank):
    """Row F3, output end (widening): BGZF compression of a coordinate-sorted BAM record stream on the device, the job of
    htslib's bgzf_compress / zlib in the reference's writer threads.  Rank 0 only.  Input batches resident in pinned host
    memory; `value` is kernel-only (blocks resident in HBM), the pinned-to-pinned rate is reported beside it."""
    if rank != 0:
        return None
    import zlib
    B, per = 0xff00, 2048
    n_batches = max(3, (args.bgzf_mb << 20) // (per * B))
    comp = pkg.BgzfCompressor(local_rank)
    src = synth.gen_bam_record_bytes(3 * per * B, 0x5EED0030)
    batches = [pkg.BgzfBatch(comp, per * B, per) for _ in range(3)]
    for k, b in enumerate(batches):
        b.input[:] = src[k * per * B:(k + 1) * per * B]
        b.offsets[:] = np.arange(per + 1, dtype=np.uint64) * B
    for b in batches:
        b.submit(per)
    first = [b.wait() for b in batches]
    ok = all(zlib.crc32(b"".join(zlib.decompress(bytes(o[int(oo[i]) + 18:int(oo[i + 1]) - 8]), -15) for i in range(0, per, 97))) ==
             zlib.crc32(b"".join(bytes(src[(k * per + i) * B:(k * per + i + 1) * B]) for i in range(0, per, 97)))
             for k, (o, oo) in enumerate(first))
    out_bytes = sum(int(oo[-1]) for _, oo in first)
    # pinned -> pinned: three batches in flight, nothing but submit / wait in the timed loop (mgx_bgzf_stats synchronises the
    # context's stream: called per batch it would drain the pipeline it is meant to measure)
    t0 = time.perf_counter()
    for i in range(n_batches):
        b = batches[i % 3]
        if i >= 3:
            lib_wait(comp, b)
        b.submit(per)
    for i in range(n_batches, n_batches + 3):
        lib_wait(comp, batches[i % 3])
    dt = time.perf_counter() - t0
    # kernel-only: one batch at a time, the HIP events around its deflate + offsets + pack kernels
    ks, ps = [], []
    for i in range(6):
        b = batches[i % 3]
        b.submit(per); lib_wait(comp, b); ks.append(comp.stats()["ms_kernels"]); ps.append(comp.stats()["ms_pack"])
    st = comp.stats()
    kernel_ms = float(np.median(ks))
    ratio = out_bytes / (3 * per * B)
    alg = per * B * (1 + ratio)                       # read the input once, write the compressed blocks once
    out = {"metric": "BGZF compression GB/s of BAM bytes (LZ77 + dynamic Huffman + CRC-32 per 65280-byte block, on the device)",
           "value": per * B / kernel_ms / 1e6, "unit": "GB/s", "dtype": "u8",
           "config": {"workload": "coordinate-sorted BAM records of BASELINE.json configs[3] (150-base reads, qualities U[2,41]), blocks of 65280 bytes",
                      "blocks_per_batch": per, "batches": n_batches, "bytes": n_batches * per * B},
           "compressed_over_input": ratio, "pinned_to_pinned_GBps": n_batches * per * B / dt / 1e9, "blocks_stored": int(st["n_stored"]),
           "pack_ms": float(np.median(ps)),
           "inflates_to_input": bool(ok),
           "roofline": {"bound": "lds", "achieved": alg / kernel_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / kernel_ms / 1e6 / HBM_PEAK_GBS, "traffic": measured_traffic("k_bgzf_deflate"),
                        "traffic_source": "offline rocprofv3 --pmc passes (profiles/pmc_traffic.json), not measured in this run",
                        "kernel": "k_bgzf_deflate", "kernel_ms": kernel_ms,
                        "kernel_ms_source": "HIP events around the deflate kernel of a batch (input resident in HBM), median over six batches; the pack kernel "
                                            "behind it stores the finished blocks straight into pinned host memory (pack_ms: that transfer)",
                        "alg_bytes_per_launch": alg,
                        "note": "what binds it is LDS latency on serial chains, not a byte rate: one wavefront per workgroup walks the hash table in "
                                "order (409 k of the 880 k cycles of a block; DESIGN.md 4.6), beside it the Huffman construction and the bit offsets; one "
                                "64 KB block per workgroup, one workgroup per CU.  No LDS roof is published for such chains, so `achieved` / `frac` are "
                                "the algorithmic bytes (input once, compressed output once) against the HBM peak, as BASELINE.json's metric words it"}}
    if not args.no_cpu_baseline:
        sample = bytes(src[:256 * B])
        t0 = time.perf_counter()
        z = sum(len(zlib.compress(sample[i:i + B], 6)) for i in range(0, len(sample), B))
        dtz = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample) / dtz / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
                               "sample": f"the first 256 blocks through zlib level 6 (what bgzf_compress calls, bgzf.c:610) on one core, {dtz:.2f} s",
                               "compressed_over_input": z / len(sample)}
    for b in batches:
        b.close()
    comp.close()
    return out


def lib_wait(comp, b):
    """mgx_bgzf_batch_wait without copying the result out of the pinned buffer"""
    import ctypes
    o, oo = ctypes.c_void_p(), ctypes.c_void_p()
    if comp.lib.mgx_bgzf_batch_wait(comp.h, b.b, ctypes.byref(o), ctypes.byref(oo)):
        raise RuntimeError(comp.lib.mgx_last_error().decode())


def cli_leg(pkg, synth, args, rank):
    """Rows B1-B2, B14 / F3 end to end (widening, not BASELINE.json's `value`): the sortmardup-compatible tool on SAM text of
    configs[3]'s distribution -- text in, BAM + BAI out -- with the BAM bytes resident in HBM and compressed on the device
    (`-z device`), and with zlib on the writer threads as the reference does (`-z zlib`).  Rank 0 only."""
    if rank != 0:
        return None
    import shutil, subprocess, tempfile, gzip, hashlib
    exe = os.path.join(ROOT, "fast-genomic-data-processing_amd", "bin", "sortmardup")
    if not os.path.exists(exe):
        return {"skipped": "the CLI is not built (python __graft_entry__.py build)"}
    d = tempfile.mkdtemp(prefix="mgx_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        n = args.cli_records
        recs, L = synth.gen_sortdedup_packed_fast(n, 0x5EED0004, threads=host_cores())
        sam = os.path.join(d, "in.sam")
        text_bytes = synth.write_sam_from_packed(sam, recs)
        del recs
        out = {"metric": "sortmardup CLI end to end, Mrecords/s (SAM text in, coordinate-sorted duplicate-marked BAM + BAI out)",
               "config": {"workload": "BASELINE.json configs[3] distribution as SAM text", "records": n, "text_bytes": text_bytes, "threads": host_cores()}}
        digests = {}
        for mode in ("device", "zlib"):
            bam = os.path.join(d, mode + ".bam")
            t0 = time.perf_counter()
            res = subprocess.run([exe, "-I", sam, "-O", bam, "-t", str(host_cores()), "-z", mode], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if res.returncode:
                out[mode] = {"error": res.stderr[-300:]}
                continue
            stages = {ln.split(":")[0].strip(): float(ln.split(":")[1].split("s")[0]) for ln in res.stdout.splitlines() if " done: " in ln}
            h = hashlib.md5()
            with gzip.open(bam, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            digests[mode] = h.hexdigest()
            tool_s = sum(stages.values())
            out[mode] = {"value": n / dt / 1e6, "unit": "Mrecords/s", "seconds": dt, "stages_s": stages, "bam_bytes": os.path.getsize(bam),
                         "tool_clock": {"seconds": tool_s, "mrecords_s": n / tool_s / 1e6 if tool_s > 0 else None,
                                        "note": "the tool's own stage clock (what the reference prints: main.cpp:597-607); the rest of `seconds` is the process "
                                                "itself: a HIP program that allocates one buffer and exits takes 0.25-0.3 s of wall clock on this box"}}
        out["same_uncompressed_stream"] = len(digests) == 2 and digests["device"] == digests["zlib"]
        out["note"] = ("wall time of the process, text in the page cache; -z device: BAM bytes resident in HBM from ingest on, gathered in sorted "
                       "order, duplicate-flagged and BGZF-compressed on the device; -z zlib: zlib level 6 on the writer threads")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def timed_resident(eng, batch, steps, warmup, barrier):
    """W untimed + K timed runs of a resident batch, bracketed by barrier + synchronize.  HIP events are
    recorded on the kernel's own stream around every launch of every run; batch.stats() after the
    final sync averages the timed runs (the warmup runs are dropped by a stats() call before them)."""
    for _ in range(warmup):
        batch.run()
    eng.sync()
    batch.stats()                      # forget the warmup runs' events
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    return dt, batch.stats()


def pairhmm_rooflines(st, traffic):
    """the binding roof of the PairHMM recurrence is fp32 VECTOR issue (12 flop per cell against the
    157.3 TFLOP/s vector peak), so that is what `roofline` reports; the HBM fraction BASELINE.json's
    metric asks for -- 5R+H+4 algorithmic bytes per test case against 8 TB/s -- rides beside it as `roofline.hbm`."""
    ms_dom = st["ms_f32_dominant"]
    alg_bytes = st["dominant_alg_bytes"]
    achieved = alg_bytes / (ms_dom * 1e-3) / 1e9
    valu = 12.0 * st["dominant_cells"] / (ms_dom * 1e-3) / 1e12
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "alg_bytes_per_launch": alg_bytes,
           "note": "5R+H+4 algorithmic bytes per test case (SURVEY.md 8d); at 36 cells per byte no correct PairHMM kernel comes near this roof"}
    roof = {"bound": "valu", "achieved": valu, "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu / VALU_FP32_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": "offline rocprofv3 --pmc passes (profiles/pmc_traffic.json), not measured in this run" if traffic else None,
            "kernel": st["dominant_kernel"], "kernel_ms": ms_dom,
            "kernel_ms_source": f"HIP events on the compute stream around every launch inside the timed loop, mean of {st['n_runs_timed']} runs",
            "step_kernels_ms": st["ms_f32"] + st["ms_f64"], "alg_flop_per_launch": 12.0 * st["dominant_cells"],
            "kernel_gcups": st["dominant_cells"] / (ms_dom * 1e-3) / 1e9, "hbm": hbm,
            "note": "12 flop per cell (SURVEY.md 8d: M 4 mul + 2 add, X and Y 2 mul + 1 add) x the cells one launch processes, against the "
                    "fp32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s = 2 cycles per wave64 v_fma_f32 per SIMD; packed fp32 has the "
                    "same peak); `traffic` is HBM bytes per launch"}
    return roof


def regions_leg(pkg, synth, local_rank, lanes):
    """Row F1: 1000 active regions of 40 reads x 25 haplotypes, host buffers in -> results in host
    memory out, timed around the C call only (the ctypes marshalling is done beforehand): one device batch
    (mgx_pairhmm_compute_regions) and the same through the queue (lanes flatten / upload the next regions while the
    previous ones compute)."""
    distinct = [synth.gen_pairhmm_region(40, 25, 1000 + g, r_range=(20, 128), h_range=(64, 256)) for g in range(50)]
    regions = [distinct[g % len(distinct)] for g in range(1000)]
    cells = sum(r["cells"] for r in regions)
    prep = pkg.pairhmm.prepare_regions(regions)
    eng = pkg.PairHMMEngine(local_rank)
    eng.compute_regions(prepared=prep)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.compute_regions(prepared=prep); ts.append(time.perf_counter() - t0)
    one = float(np.median(ts))
    want = [x.copy() for x in prep["outs"]]
    q = pkg.PairHMMQueue(devices=(local_rank,), lanes_per_device=lanes, depth=2, batch_pairs=131072)
    q.run_regions(prepared=prep)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); got = q.run_regions(prepared=prep); ts.append(time.perf_counter() - t0)
    qt = float(np.median(ts))
    same = all(np.array_equal(a, b) for a, b in zip(got, want))
    q.close(); eng.close()
    return {"metric": "PairHMM GCUPS, 1000 regions of 40 reads x 25 haplotypes (row F1), host buffers in -> host results out",
            "cells": cells, "one_batch": {"ms": one * 1e3, "value": cells / one / 1e9, "unit": "GCUPS"},
            "queue": {"ms": qt * 1e3, "value": cells / qt / 1e9, "unit": "GCUPS", "lanes": lanes, "identical_to_one_batch": same},
            "note": "read U[20,128] x haplotype U[64,256]; includes flattening, H2D, kernels, D2H and the scatter into per-region outputs"}


def mixed_leg(pkg, synth, args, rank, local_rank, world, shard, barrier, max_over_ranks):
    """BASELINE.json configs[4]: the PairHMM work queue and the sort / mark-duplicate pipeline co-resident on every GPU,
    each driven by its own host thread on its own streams; measured alone and together over the same wall-clock window."""
    import threading
    total = args.total_pairs if world > 1 else 4 << 20
    lo, hi = shard.shard_bounds(total, rank, world)
    n_local = min(hi - lo, 4 << 20)                    # a window of the rank's shard keeps the leg short
    d = synth.gen_pairhmm_pairs_fast(n_local, 0x5EED0003, first_pair=lo, threads=host_threads(world))
    prepared = pkg.pairhmm.make_input(d)
    # four lanes keep the link as busy as eight (the queue is PCIe-bound) and leave the 4 x 3 PairHMM streams and the sort's 4 one
    # hardware queue each out of GPU_MAX_HW_QUEUES=16: streams that share a hardware queue serialise (sort 2 200 -> 4 900 Mrecords/s)
    lanes = args.queue_lanes or max(2, min(4, host_cores() // world))
    q = pkg.PairHMMQueue(devices=(local_rank,), lanes_per_device=lanes, depth=2, batch_pairs=65536)
    # "interleave config-3 batches with 8 coordinate-range shards of config 4": a GPU's share of
    # the record set is one eighth of configs[3] (25 M records), whatever the number of ranks of THIS run
    n_rec = max(args.sort_records // max(world, 8), 1_000_000) if args.sort_records else 25_000_000
    recs, L = synth.gen_sortdedup_packed_fast(n_rec, 0x5EED0004 + rank, threads=host_threads(world))
    sort_flags = int(os.environ.get("MGX_MIXED_SORT_FLAGS", str(1 << 16)), 0)      # MGX_STREAM_HIGH_PRIORITY (A/B: 0)
    eng = pkg.SortDedupEngine(local_rank, flags=sort_flags)
    eng.upload(L, recs)
    q.run(d, lo=0, hi=min(n_local, lanes * 2 * 65536), prepared=prepared); eng.run(); eng.stats()

    def run_for(seconds, do_hmm, do_sort):
        stop = time.perf_counter() + seconds
        counts = [0, 0]

        def hmm():
            while time.perf_counter() < stop:
                q.run(d, prepared=prepared); counts[0] += 1

        def srt():
            while time.perf_counter() < stop:
                eng.run(); eng.stats(); counts[1] += 1
        th = [threading.Thread(target=f) for f, on in ((hmm, do_hmm), (srt, do_sort)) if on]
        barrier()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        return counts[0] * d["cells"] / dt / 1e9, counts[1] * len(recs) / dt / 1e6
    hmm_alone, _ = run_for(args.mixed_seconds, True, False)
    _, sort_alone = run_for(args.mixed_seconds, False, True)
    hmm_mixed, sort_mixed = run_for(2 * args.mixed_seconds, True, True)
    q.close(); eng.close()
    if rank != 0:
        return None
    return {"metric": "BASELINE.json configs[4]: PairHMM queue + sort/mark-duplicate pipeline co-resident, per GPU (rank 0's figures)",
            "pairhmm_queue_gcups": {"alone": hmm_alone, "mixed": hmm_mixed}, "sortmardup_mrecords_s": {"alone": sort_alone, "mixed": sort_mixed},
            "combined_utilisation": hmm_mixed / max(hmm_alone, 1e-9) + sort_mixed / max(sort_alone, 1e-9),
            "config": {"pairs_per_gpu_window": n_local, "records_per_gpu": len(recs), "queue_lanes": lanes,
                       "sort_streams": "highest priority (MGX_STREAM_HIGH_PRIORITY)" if sort_flags & (1 << 16) else "default priority",
                       "window_s": {"alone": args.mixed_seconds, "mixed": 2 * args.mixed_seconds}},
            "note": "two host threads per GPU, separate contexts and streams; PairHMM streamed from host memory through the work queue, "
                    "sort records resident (independent per-GPU record sets in this leg)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=0, help="test cases per GPU (default: 1M at one GPU = BASELINE configs[1]; "
                                                         "--total-pairs / N at N GPUs = configs[2])")
    ap.add_argument("--total-pairs", type=int, default=64 << 20, help="configs[2]: test cases of the ONE stream the N ranks share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sort-records", type=int, default=200_000_000,
                    help="records of the sortmardup leg (BASELINE.json configs[3]; ONE data set, sharded over the ranks); 0 disables it")
    ap.add_argument("--sort-steps", type=int, default=5)
    ap.add_argument("--sw-pairs", type=int, default=20000, help="Smith-Waterman pairs (row F4 leg; 0 skips it)")
    ap.add_argument("--bgzf-mb", type=int, default=1024, help="MB of BAM bytes through the device BGZF compressor (row F3 leg; 0 skips it)")
    ap.add_argument("--cli-records", type=int, default=4_000_000, help="records through the sortmardup CLI end to end (0 skips the leg)")
    ap.add_argument("--queue-lanes", type=int, default=0, help="host lanes of the work queue (default: min(8, cores / ranks))")
    ap.add_argument("--no-ragged", action="store_true", help="skip sub-run 2b (ragged lengths)")
    ap.add_argument("--no-regions", action="store_true", help="skip the row-F1 leg (1000 regions of 40 x 25)")
    ap.add_argument("--no-queue", action="store_true", help="skip the host-work-queue leg (counter-collection runs)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the BASELINE.json configs[4] leg (PairHMM queue and sort/mark-duplicate "
                                                             "pipeline co-resident on every GPU)")
    ap.add_argument("--mixed", action="store_true", help="(default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--mixed-seconds", type=float, default=1.0, help="window of each solo phase of the configs[4] leg; the co-resident phase runs twice that")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N>1 path on one GPU
    n_dev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(n_dev, 1)          # ranks share the card in a rehearsal
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)

    os.environ.setdefault("MGX_ROUTE_THREADS", str(host_threads(world)))      # the router's threads, per rank
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    shard = importlib.import_module(PKG + ".shard")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    def max_over_ranks(x):
        return shard.max_over_ranks(x, dist, f"cuda:{local_rank}" if backend == "nccl" else "cpu")

    # ---- workload: N = 1 -> BASELINE configs[1] (1M test cases, sub-run 2a); N > 1 -> configs[2]: ONE stream of
    # --total-pairs test cases (2a distribution, seed 0x5EED0003), rank r owns test cases shard_bounds(total, r, N)
    if world == 1:
        seed, total = 0x5EED0002, args.pairs or (1 << 20)
        lo, hi = 0, total
        what = f"BASELINE.json configs[1], sub-run 2a: {total} synthetic read x haplotype test cases, read 128 x hap 256, fp32 + fp64 re-run of results < 1e-28"
    else:
        seed, total = 0x5EED0003, (args.pairs * world if args.pairs else args.total_pairs)
        lo, hi = shard.shard_bounds(total, rank, world)
        what = (f"BASELINE.json configs[2]: ONE stream of {total} synthetic test cases (read 128 x hap 256, seed 0x5EED0003) split over {world} ranks "
                f"({hi - lo} per GPU), fp32 + fp64 re-run; value = every rank's shard resident in HBM; 'queue' = the same shard streamed from host "
                "memory through the host work queue (65536-test-case batches, PCIe-inclusive)")
    d = synth.gen_pairhmm_pairs_fast(hi - lo, seed, first_pair=lo, threads=host_threads(world))      # 2a: fixed R=128, H=256
    eng = pkg.PairHMMEngine(local_rank, flags=pkg.pairhmm.TIMING)
    t0 = time.perf_counter()
    batch = eng.batch(d)                               # bins + uploads: resident in HBM from here on
    eng.sync()
    upload_s = time.perf_counter() - t0
    dt, st = timed_resident(eng, batch, args.steps, args.warmup, barrier)
    tmax = max_over_ranks(dt)
    value = total * 128 * 256 * args.steps / tmax / 1e9    # every test case of the stream is 128 x 256 cells
    batch.close()

    line = None
    if rank == 0:
        roof = pairhmm_rooflines(st, measured_traffic(st["dominant_kernel"]) if hi - lo == (1 << 20) else None)
        line = {
            "metric": "PairHMM GCUPS", "value": value, "unit": "GCUPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": tmax / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": what, "pairs_total": total, "pairs_per_gpu": hi - lo, "read_len": 128, "hap_len": 256, "seed": hex(seed),
                       "rerun_f64_per_step": st["n_rerun_f64"], "parallelism": f"shard{world}"},
            "roofline": roof, "upload_s": upload_s,
        }

    # ---- the same shard through the host work queue: host buffers in, results in host memory out (PCIe-inclusive;
    # never `value`).  Lanes pack / upload batch k+1 while the kernels of batch k run.
    if not args.no_queue:
        lanes = args.queue_lanes or max(2, min(8, host_cores() // world))      # the ranks of a node share its cores
        if world == 1:
            # at one GPU the resident batch is configs[1] (1M test cases); the queue is measured on a longer stream of the
            # same distribution (4M test cases = 64 batches) so that it reaches its steady state
            del d
            n_local = max(total, 4 << 20)
            d = synth.gen_pairhmm_pairs_fast(n_local, seed, threads=host_threads(world))
        else:
            n_local = hi - lo
        q = pkg.PairHMMQueue(devices=(local_rank,), lanes_per_device=lanes, depth=2, batch_pairs=65536)
        prepared = pkg.pairhmm.make_input(d)
        q.run(d, lo=0, hi=min(n_local, lanes * 2 * 65536), prepared=prepared)      # warm-up: pinned slabs are allocated on first use
        barrier()
        t0 = time.perf_counter()
        qout = q.run(d, prepared=prepared)
        barrier()
        qdt = max_over_ranks(time.perf_counter() - t0)
        qst = q.stats()
        q.close()
        q_total = n_local if world == 1 else total
        if rank == 0:
            line["queue"] = {"value": q_total * 128 * 256 / qdt / 1e9, "unit": "GCUPS", "seconds": qdt, "pcie_inclusive": True,
                             "test_cases": q_total, "lanes_per_gpu": lanes, "depth": 2, "batch_pairs": 65536, "batches_per_gpu": qst["n_batches"],
                             "h2d_GBps_per_gpu": qst["bytes_h2d"] / qst["seconds"] / 1e9, "pack_s_per_lane": qst["pack_seconds"] / lanes,
                             "wait_s_per_lane": qst["wait_seconds"] / lanes,
                             "note": "one pass of the host work queue over this rank's stream: pack (gather + bin) -> pinned slab -> H2D -> kernels "
                                     "-> D2H, results in host memory; bounded by the PCIe link at 900 bytes per 128x256 test case"}
            # the streamed results are the single call's results
            chk = eng.compute(pkg.pairhmm.pack_batch(d, n_local - 4096, n_local))
            line["queue"]["identical_to_single_call"] = bool(np.array_equal(chk, qout[n_local - 4096:]))
        del qout

    # ---- sub-run 2b: ragged lengths R in [32,128], H in [64,256]
    if not args.no_ragged:
        n2 = hi - lo if world == 1 else min(hi - lo, 1 << 20)
        d2 = synth.gen_pairhmm_pairs_fast(n2, 0x5EED0002 + 0x100 * rank, r_range=(32, 128), h_range=(64, 256), threads=host_threads(world))
        b2 = eng.batch(d2)
        dt2, st2 = timed_resident(eng, b2, args.steps, args.warmup, barrier)
        t2 = max_over_ranks(dt2)
        b2.close()
        if rank == 0:
            r2 = pairhmm_rooflines(st2, None)
            line["ragged"] = {"metric": "PairHMM GCUPS, sub-run 2b", "value": d2["cells"] * world * args.steps / t2 / 1e9, "unit": "GCUPS",
                              "ms_per_step": t2 / args.steps * 1e3, "scaling": "weak",
                              "config": {"workload": f"SURVEY.md 8d config 2b: {n2} test cases per GPU, read U[32,128] x hap U[64,256], resident",
                                         "cells_per_gpu": d2["cells"], "launches_f32": st2["n_launches_f32"], "rerun_f64_per_step": st2["n_rerun_f64"]},
                              "roofline": r2,
                              "valu_all_kernels": {"frac": 12.0 * d2["cells"] / (st2["ms_f32"] * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS,
                                                   "note": "all fp32 launches of a step (one per read-length class), 12 flop per cell"}}
        del d2
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(synth, 1 << 18, seed)
    d = prepared = None
    eng.close()

    # ---- second half of BASELINE.json's metric: sortmardup Mrecords/s (configs[3]) ------------
    sort_line = None
    if args.sort_records > 0:
        sort_line = sortmardup_leg(pkg, synth, args, rank, local_rank, world, dist, torch, backend)
    sw_line = smithwaterman_leg(pkg, synth, args, rank, local_rank) if args.sw_pairs > 0 else None
    bgzf_line = bgzf_leg(pkg, synth, args, rank, local_rank) if args.bgzf_mb > 0 else None
    cli_line = cli_leg(pkg, synth, args, rank) if args.cli_records > 0 else None
    mixed_line = mixed_leg(pkg, synth, args, rank, local_rank, world, shard, barrier, max_over_ranks) if not args.no_mixed else None
    if rank == 0:
        if not args.no_regions:
            line["regions"] = regions_leg(pkg, synth, local_rank, 3)      # three lanes, 131072 test cases per batch: the best of the sweep in profiles/r03_regions_queue_sweep.txt
        if mixed_line is not None:
            line["mixed"] = mixed_line
        if sort_line is not None:
            line["sortmardup"] = sort_line
        if sw_line is not None:
            line["smithwaterman"] = sw_line
        if bgzf_line is not None:
            line["bgzf"] = bgzf_line
        if cli_line is not None:
            line["cli"] = cli_line
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
