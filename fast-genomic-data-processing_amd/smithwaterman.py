"""Host-side handle on the Smith-Waterman C ABI (include/mgx_smithwaterman.h).

Mirrors the reference's aligner interface: ``SmithWatermanEngine.align(ref, alt, params, strategy)``
is ``IntelSmithWaterman::align`` (smithwaterman/IntelSmithWaterman.cpp:5-16: CIGAR text + offset),
``align_batch`` the same over many pairs in one device round trip.
"""
import ctypes as C

import numpy as np

from . import native

SOFTCLIP, INDEL, LEADING_INDEL, IGNORE = 9, 10, 11, 12
STANDARD_NGS = (25, -50, -110, -6)       # smithwaterman/SmithWatermanAligner.cpp:9
ORIGINAL_DEFAULT = (3, -1, -4, -3)       # smithwaterman/SmithWatermanAligner.cpp:8


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class SmithWatermanEngine:
    def __init__(self, device=0, flags=0):
        self.lib = native.load()
        ctx = C.c_void_p()
        native.check(self.lib.mgx_sw_create(device, flags, C.byref(ctx)))
        self.ctx = ctx

    def align_batch(self, ref_off, ref, alt_off, alt, strategy, params=STANDARD_NGS, stride=None, want_score=False):
        """Concatenated sequences with [n+1] offsets; strategy: uint8 [n].  Returns (cigars [list of bytes],
        offsets int32 [n]) and the scores when asked."""
        ref_off = np.ascontiguousarray(ref_off, dtype=np.uint64); alt_off = np.ascontiguousarray(alt_off, dtype=np.uint64)
        ref = np.ascontiguousarray(ref, dtype=np.uint8); alt = np.ascontiguousarray(alt, dtype=np.uint8)
        strategy = np.ascontiguousarray(strategy, dtype=np.uint8)
        n = len(strategy)
        if stride is None:
            l1 = np.diff(ref_off.astype(np.int64)); l2 = np.diff(alt_off.astype(np.int64))
            stride = int(2 * max(int(l1.max(initial=1)), int(l2.max(initial=1))) + 1) if n else 2
        P = native.SwParams(*params)
        inp = native.SwInput(n, _ptr(ref_off), _ptr(ref), _ptr(alt_off), _ptr(alt), _ptr(strategy))
        off = np.zeros(n, dtype=np.int32)
        cig = np.zeros((n, stride), dtype=np.uint8)
        score = np.zeros(n, dtype=np.int32)
        native.check(self.lib.mgx_sw_align_batch(self.ctx, C.byref(P), C.byref(inp), _ptr(off), _ptr(cig), stride,
                                                 _ptr(score) if want_score else None))
        lens = (cig != 0).sum(axis=1)                    # the text holds no NUL before its end
        flat = cig.tobytes()
        cigars = [flat[p * stride:p * stride + int(lens[p])] for p in range(n)]
        return (cigars, off, score) if want_score else (cigars, off)

    def align(self, ref, alt, params=STANDARD_NGS, strategy=SOFTCLIP, cigar_length=None):
        """One pair through the SmithWaterman_align-compatible entry point: (cigar bytes, offset)."""
        ref = np.frombuffer(bytes(ref), dtype=np.uint8); alt = np.frombuffer(bytes(alt), dtype=np.uint8)
        cap = 2 * max(len(ref), len(alt)) if cigar_length is None else cigar_length
        buf = np.zeros(cap + 1, dtype=np.uint8)
        rc = self.lib.mgx_sw_align(self.ctx, _ptr(ref), len(ref), _ptr(alt), len(alt), _ptr(buf), cap,
                                   params[0], params[1], params[2], params[3], strategy)
        if rc <= -1000000:
            native.check(-(rc + 1000000) or 1)
        return bytes(buf[:int(np.argmax(buf == 0))]), rc

    def stats(self):
        st = native.SwStats()
        native.check(self.lib.mgx_sw_stats(self.ctx, C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in st._fields_}

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mgx_sw_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
