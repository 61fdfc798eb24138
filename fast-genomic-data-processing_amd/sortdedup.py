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


def pack(raw):
    """raw: dict of arrays in the layout of mgx_raw_records_t (see synth.RawRecords.arrays()).
    Returns (recs [arrival order, REC_DTYPE], input_index [arrival -> input], L)."""
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
    native.check(lib.mgx_sortdedup_pack(C.byref(rr), _ptr(recs), _ptr(idx), C.byref(L)))
    return recs, idx, int(L.value)


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
        self.n = len(recs)
        native.check(self.lib.mgx_sortdedup_upload(self.ctx, L, self.n, _ptr(recs)))

    def run(self):
        native.check(self.lib.mgx_sortdedup_run(self.ctx))

    def results(self):
        order = np.empty(self.n, dtype=np.uint32)
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
