"""Host-side handle on the sort / mark-duplicate C ABI (include/mgx_sortdedup.h).

``pack`` is the record-level half of sortmardup (mate discovery, key derivation: bam_parser.cpp,
bam_record.cpp, pair.cpp) and runs on the host next to the parser; ``SortDedupEngine`` is the
device half (sortmardup/main.cpp:235-388: pair sorts, duplicate search, coordinate sort).
"""
import ctypes as C

import numpy as np

from . import native
from .synth import REC_DTYPE


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(raw, score=None):
    """raw: dict of arrays in the layout of mgx_raw_records_t (see synth.RawRecords.arrays()).
    score: optional uint16 [n], BAMRecord::score per record from the caller's own pass over the qualities
    (mgx_sortdedup_pack_scored).  Returns (recs [arrival order, REC_DTYPE], input_index [arrival -> input], L)."""
    lib = native.load()
    n = int(raw["n_records"])
    keep = {k: np.ascontiguousarray(raw[k]) for k in
            ("flag", "tid", "pos", "cigar_off", "cigar", "qual_off", "qual", "qname_off", "qname", "target_len")}
    rr = native.RawRecords(n_records=n, flag=_ptr(keep["flag"]), tid=_ptr(keep["tid"]), pos=_ptr(keep["pos"]),
                           cigar_off=_ptr(keep["cigar_off"]), cigar=_ptr(keep["cigar"]),
                           qual_off=_ptr(keep["qual_off"]), qual=_ptr(keep["qual"]),
                           qname_off=_ptr(keep["qname_off"]), qname=_ptr(keep["qname"]),
                           n_targets=len(keep["target_len"]), target_len=_ptr(keep["target_len"]))
    recs = np.zeros(n, dtype=REC_DTYPE)
    idx = np.zeros(n, dtype=np.uint32)
    L = C.c_uint64()
    if score is None:
        native.check(lib.mgx_sortdedup_pack(C.byref(rr), _ptr(recs), _ptr(idx), C.byref(L)))
    else:
        score = np.ascontiguousarray(score, dtype=np.uint16)
        if len(score) != n:
            raise ValueError("score needs one entry per record")
        rr.qual_off = None; rr.qual = None              # mgx_sortdedup_pack_scored does not look at the qualities
        native.check(lib.mgx_sortdedup_pack_scored(C.byref(rr), _ptr(score), _ptr(recs), _ptr(idx), C.byref(L)))
    return recs, idx, int(L.value)


class Routed:
    """One record set cut into coordinate shards by the host router (mgx_sortdedup_route): the
    reference's range partitioners + shared bitmap, for GPUs that share nothing."""

    def __init__(self, L, recs, n_shards, only_shard=-1):
        self.lib = native.load()
        recs = np.ascontiguousarray(recs)
        assert recs.dtype == REC_DTYPE
        self.L, self.n_records, self.n_shards = int(L), len(recs), n_shards
        h = C.c_void_p()
        native.check(self.lib.mgx_sortdedup_route(self.L, self.n_records, _ptr(recs), n_shards, only_shard, C.byref(h)))
        self.h = h

    def shard(self, k):
        sh = native.SortDedupShard()
        native.check(self.lib.mgx_sortdedup_routed_shard(self.h, k, C.byref(sh)))
        return sh

    def shard_arrays(self, k):
        """Copies of shard k's arrays (tests / the CPU checker)."""
        sh = self.shard(k)

        def view(ptr, count, dt):
            if not count:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
        return dict(order_coord=view(sh.order_coord, sh.n_order, np.uint64), order_arrival=view(sh.order_arrival, sh.n_order, np.uint32),
                    mark_recs=view(sh.mark_recs, sh.n_mark, REC_DTYPE), mark_arrival=view(sh.mark_arrival, sh.n_mark, np.uint32),
                    marks=view(sh.marks, sh.n_marks, np.uint64), order_base=int(sh.order_base), coord_lo=int(sh.coord_lo), coord_hi=int(sh.coord_hi))

    def merge(self, k, shard_order, shard_dup, out_order, out_dup):
        native.check(self.lib.mgx_sortdedup_merge(self.h, k, _ptr(np.ascontiguousarray(shard_order)), _ptr(np.ascontiguousarray(shard_dup)),
                                                  _ptr(out_order), _ptr(out_dup)))

    def close(self):
        if self.h:
            self.lib.mgx_sortdedup_routed_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SortDedupEngine:
    def __init__(self, device=0, flags=0):
        self.lib = native.load()
        ctx = C.c_void_p()
        native.check(self.lib.mgx_sortdedup_create(device, flags, C.byref(ctx)))
        self.ctx = ctx
        self.n = 0

    def upload(self, L, recs):
        recs = np.ascontiguousarray(recs)
        assert recs.dtype == REC_DTYPE
        self.n = self.n_order = len(recs)
        native.check(self.lib.mgx_sortdedup_upload(self.ctx, L, self.n, _ptr(recs)))

    def upload_chunks(self, L, chunks, n_expected=0):
        """Streamed upload: ``chunks`` yields record arrays in arrival order (mate indices global)."""
        native.check(self.lib.mgx_sortdedup_upload_begin(self.ctx, L, n_expected))
        at = 0
        for ch in chunks:
            ch = np.ascontiguousarray(ch)
            assert ch.dtype == REC_DTYPE
            native.check(self.lib.mgx_sortdedup_upload_chunk(self.ctx, at, len(ch), _ptr(ch)))
            at += len(ch)
        native.check(self.lib.mgx_sortdedup_upload_end(self.ctx, at))
        self.n = self.n_order = at

    def upload_shard(self, routed, k):
        """One shard of a routed record set: results() then returns (global arrival indices of the shard's
        records in output order, duplicate flag per marking record)."""
        sh = routed.shard(k)
        self.n, self.n_order = int(sh.n_mark), int(sh.n_order)
        native.check(self.lib.mgx_sortdedup_upload_shard(self.ctx, routed.L, C.byref(sh)))

    def run(self):
        native.check(self.lib.mgx_sortdedup_run(self.ctx))

    def results(self):
        order = np.empty(getattr(self, "n_order", self.n), dtype=np.uint32)
        dup = np.empty(self.n, dtype=np.uint8)
        native.check(self.lib.mgx_sortdedup_results(self.ctx, _ptr(order), _ptr(dup)))
        return order, dup

    def stats(self):
        st = native.SortDedupStats()
        native.check(self.lib.mgx_sortdedup_stats(self.ctx, C.byref(st)))
        return {k: getattr(st, k) for k, _ in native.SortDedupStats._fields_}

    def sort_mark(self, L, recs):
        self.upload(L, recs)
        self.run()
        return self.results()

    def close(self):
        if self.ctx:
            self.lib.mgx_sortdedup_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
