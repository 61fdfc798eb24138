"""Host-side handle on the PairHMM C ABI (include/mgx_pairhmm.h).

The argument layout is the flattened form of the reference's ``testcase`` list
(deepmutect/Mutect2Cpp-master/src/intel/pairhmm/pairhmm_common.h:45-57) as built by
VectorLoglessPairHMM::computeLog10Likelihoods (utils/pairhmm/VectorLoglessPairHMM.cpp:71-119).
"""
import ctypes as C

import numpy as np

from . import native

FORCE_DOUBLE = 1
TIMING = 2
PACKED_FP32 = 4      # the packed fp32 kernel for classes with an even number of rows per lane (bit-identical, slower: DESIGN.md 3.7)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def make_input(d):
    """dict of packed arrays (see synth.gen_pairhmm_pairs) -> (PairHMMInput, keepalive)."""
    keep = dict(
        read_off=_as(d["read_off"], np.uint64), bases=_as(d["bases"], np.uint8),
        qual=_as(d["qual"], np.uint8), ins=_as(d["ins"], np.uint8), dele=_as(d["dele"], np.uint8),
        gcp=_as(d["gcp"], np.uint8), hap_off=_as(d["hap_off"], np.uint64),
        hap_bases=_as(d["hap_bases"], np.uint8))
    cross = d.get("pair_read") is None          # every read x every haplotype, read-major
    if not cross:
        keep["pair_read"] = _as(d["pair_read"], np.uint32)
        keep["pair_hap"] = _as(d["pair_hap"], np.uint32)
    n_reads, n_haps = len(keep["read_off"]) - 1, len(keep["hap_off"]) - 1
    inp = native.PairHMMInput(
        n_reads=len(keep["read_off"]) - 1, read_off=_ptr(keep["read_off"]), bases=_ptr(keep["bases"]),
        qual=_ptr(keep["qual"]), ins=_ptr(keep["ins"]), del_=_ptr(keep["dele"]), gcp=_ptr(keep["gcp"]),
        n_haps=len(keep["hap_off"]) - 1, hap_off=_ptr(keep["hap_off"]), hap_bases=_ptr(keep["hap_bases"]),
        n_pairs=n_reads * n_haps if cross else len(keep["pair_read"]),
        pair_read=None if cross else _ptr(keep["pair_read"]), pair_hap=None if cross else _ptr(keep["pair_hap"]))
    return inp, keep


def prepare_regions(regions):
    """ctypes marshalling of a list of regions (dicts in the cross-product form) for compute_regions /
    run_regions, done once so that a timed call measures the library and not the Python veneer."""
    n = len(regions)
    arr = (native.PairHMMInput * n)()
    keeps, outs = [], []
    ptrs = (C.c_void_p * n)()
    for g, d in enumerate(regions):
        d = dict(d); d["pair_read"] = None; d["pair_hap"] = None
        inp, keep = make_input(d)
        arr[g] = inp; keeps.append(keep)
        o = np.empty((int(inp.n_reads), int(inp.n_haps)), dtype=np.float64)
        outs.append(o); ptrs[g] = o.ctypes.data
    return dict(n=n, arr=arr, keeps=keeps, outs=outs, ptrs=ptrs)


def pack_batch(d, lo, hi):
    """The work queue's host packer (no device): test cases [lo, hi) of the stream ``d`` as a
    self-contained dict in the same packed layout -- every referenced read / haplotype once, in
    first-use order, local indices."""
    lib = native.load()
    inp, keep = make_input(d)
    out = native.PairHMMInput()
    need = C.c_size_t()
    rc = lib.mgx_pairhmm_pack_batch(C.byref(inp), lo, hi, None, 0, C.byref(out), C.byref(need))
    if rc != -28:          # -ENOSPC is the sizing answer
        native.check(rc)
    buf = np.zeros(max(int(need.value), 1), dtype=np.uint8)
    native.check(lib.mgx_pairhmm_pack_batch(C.byref(inp), lo, hi, _ptr(buf), buf.nbytes, C.byref(out), C.byref(need)))
    base = buf.ctypes.data

    def view(ptr, count, dt):
        o = int(ptr or base) - base
        return buf[o:o + count * np.dtype(dt).itemsize].view(dt).copy()
    nr, nh, n = int(out.n_reads), int(out.n_haps), int(out.n_pairs)
    read_off = view(out.read_off, nr + 1, np.uint64); hap_off = view(out.hap_off, nh + 1, np.uint64)
    rb, hb = int(read_off[-1]), int(hap_off[-1])
    return dict(n_reads=nr, n_haps=nh, n_pairs=n, read_off=read_off, hap_off=hap_off,
                bases=view(out.bases, rb, np.uint8), qual=view(out.qual, rb, np.uint8), ins=view(out.ins, rb, np.uint8),
                dele=view(out.del_, rb, np.uint8), gcp=view(out.gcp, rb, np.uint8), hap_bases=view(out.hap_bases, hb, np.uint8),
                pair_read=view(out.pair_read, n, np.uint32), pair_hap=view(out.pair_hap, n, np.uint32))


class PairHMMQueue:
    """Host work queue over one or more devices (include/mgx_pairhmm.h, mgx_pairhmm_queue_*):
    BASELINE.json configs[2], the reference's worker threads pulling regions off an atomic index."""

    def __init__(self, devices=(0,), lanes_per_device=0, depth=0, batch_pairs=0, flags=0):
        self.lib = native.load()
        devs = (C.c_int * len(devices))(*devices)
        cfg = native.QueueConfig(n_devices=len(devices), devices=C.cast(devs, C.c_void_p), lanes_per_device=lanes_per_device,
                                 depth=depth, batch_pairs=batch_pairs, flags=flags)
        q = C.c_void_p()
        native.check(self.lib.mgx_pairhmm_queue_create(C.byref(cfg), C.byref(q)))
        self.q = q

    def run(self, d, lo=None, hi=None, with_flags=False, prepared=None):
        """Log10 likelihoods of test cases [lo, hi) of the stream (default: all of it)."""
        inp, keep = prepared if prepared is not None else make_input(d)
        total = int(inp.n_pairs)
        lo = 0 if lo is None else lo
        hi = total if hi is None else hi
        out = np.empty(hi - lo, dtype=np.float64)
        used = np.zeros(hi - lo, dtype=np.uint8) if with_flags else None
        native.check(self.lib.mgx_pairhmm_queue_run_range(self.q, C.byref(inp), lo, hi, _ptr(out), _ptr(used) if with_flags else None))
        return (out, used) if with_flags else out

    def run_regions(self, regions=None, prepared=None):
        """Row F1 through the queue: one [n_reads][n_haps] array per region."""
        p = prepared if prepared is not None else prepare_regions(regions)
        native.check(self.lib.mgx_pairhmm_queue_run_regions(self.q, p["n"], C.cast(p["arr"], C.c_void_p), C.cast(p["ptrs"], C.c_void_p)))
        return p["outs"]

    def stats(self):
        st = native.QueueStats()
        native.check(self.lib.mgx_pairhmm_queue_stats(self.q, C.byref(st)))
        d = {k: getattr(st, k) for k, _ in native.QueueStats._fields_ if k != "batches_per_device"}
        d["batches_per_device"] = list(st.batches_per_device)
        return d

    def close(self):
        if self.q:
            self.lib.mgx_pairhmm_queue_destroy(self.q)
            self.q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PairHMMBatch:
    def __init__(self, engine, d):
        self.engine = engine
        inp, keep = make_input(d)
        self.n_pairs = int(inp.n_pairs)
        h = C.c_void_p()
        native.check(engine.lib.mgx_pairhmm_batch_create(engine.ctx, C.byref(inp), C.byref(h)))
        self.h = h

    def run(self):
        native.check(self.engine.lib.mgx_pairhmm_batch_run(self.engine.ctx, self.h))

    def results(self, with_flags=False):
        out = np.empty(self.n_pairs, dtype=np.float64)
        used = np.zeros(self.n_pairs, dtype=np.uint8)
        native.check(self.engine.lib.mgx_pairhmm_batch_results(self.engine.ctx, self.h, _ptr(out), _ptr(used)))
        return (out, used) if with_flags else out

    def stats(self):
        st = native.PairHMMStats()
        native.check(self.engine.lib.mgx_pairhmm_batch_stats(self.engine.ctx, self.h, C.byref(st)))
        return {k: (getattr(st, k).decode() if k == "dominant_kernel" else getattr(st, k))
                for k, _ in native.PairHMMStats._fields_}

    def close(self):
        if self.h:
            self.engine.lib.mgx_pairhmm_batch_destroy(self.engine.ctx, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PairHMMEngine:
    """One context = one device + one compute stream (use one per worker thread)."""

    def __init__(self, device=0, flags=0):
        self.lib = native.load()
        ctx = C.c_void_p()
        native.check(self.lib.mgx_pairhmm_create(device, flags, C.byref(ctx)))
        self.ctx = ctx

    def compute(self, d):
        """One shot: returns log10 likelihoods, one per test case."""
        inp, keep = make_input(d)
        out = np.empty(inp.n_pairs, dtype=np.float64)
        native.check(self.lib.mgx_pairhmm_compute(self.ctx, C.byref(inp), _ptr(out)))
        return out

    def compute_regions(self, regions=None, prepared=None):
        """Row F1: several active regions (dicts in the cross-product form) in ONE device batch.
        Returns one [n_reads][n_haps] array per region."""
        p = prepared if prepared is not None else prepare_regions(regions)
        native.check(self.lib.mgx_pairhmm_compute_regions(self.ctx, p["n"], C.cast(p["arr"], C.c_void_p), C.cast(p["ptrs"], C.c_void_p)))
        return p["outs"]

    def regions(self, regions, mapqs, **model_overrides):
        """Rows F1 + F2: computeReadLikelihoods for several regions in one device batch.
        Returns [(log10 [n_reads][n_haps], keep [n_reads]) per region]."""
        n = len(regions)
        arr = (native.PairHMMInput * n)()
        keeps, outs, kept, mqs = [], [], [], []
        p_out, p_keep, p_mq = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for g, d in enumerate(regions):
            d = dict(d); d["pair_read"] = None; d["pair_hap"] = None
            inp, keep = make_input(d)
            arr[g] = inp; keeps.append(keep)
            o = np.empty((int(inp.n_reads), int(inp.n_haps)), dtype=np.float64); k = np.zeros(int(inp.n_reads), dtype=np.uint8)
            mq = np.ascontiguousarray(mapqs[g], dtype=np.uint8)
            outs.append(o); kept.append(k); mqs.append(mq)
            p_out[g], p_keep[g], p_mq[g] = o.ctypes.data, k.ctypes.data, mq.ctypes.data
        m = native.ReadModel()
        self.lib.mgx_read_model_defaults(C.byref(m))
        for k2, v in model_overrides.items():
            setattr(m, k2, v)
        native.check(self.lib.mgx_pairhmm_regions(self.ctx, n, C.cast(arr, C.c_void_p), C.cast(p_mq, C.c_void_p), C.byref(m),
                                                  C.cast(p_out, C.c_void_p), C.cast(p_keep, C.c_void_p)))
        return list(zip(outs, kept))

    def region(self, d, mapq, **model_overrides):
        """computeReadLikelihoods for one sample on the device: raw qualities in, normalised
        [n_reads][n_haps] log10 likelihoods and the keep mask of filterPoorlyModeledEvidence out."""
        d = dict(d); d["pair_read"] = None; d["pair_hap"] = None
        inp, keep = make_input(d)
        m = native.ReadModel()
        self.lib.mgx_read_model_defaults(C.byref(m))
        for k, v in model_overrides.items():
            setattr(m, k, v)
        mq = np.ascontiguousarray(mapq, dtype=np.uint8)
        n_reads, n_haps = len(keep["read_off"]) - 1, len(keep["hap_off"]) - 1
        out = np.empty((n_reads, n_haps), dtype=np.float64)
        kept = np.zeros(n_reads, dtype=np.uint8)
        native.check(self.lib.mgx_pairhmm_region(self.ctx, C.byref(inp), _ptr(mq), C.byref(m), _ptr(out), _ptr(kept)))
        return out, kept

    def batch(self, d):
        return PairHMMBatch(self, d)

    def sync(self):
        native.check(self.lib.mgx_pairhmm_sync(self.ctx))

    def close(self):
        if self.ctx:
            self.lib.mgx_pairhmm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
