"""Host-side handle on the BGZF compressor's C ABI (include/mgx_bgzf.h).  The compression runs on the device only."""
import ctypes as C

import numpy as np

from . import native

MAX_BLOCK_IN = 0xff00
EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class BgzfCompressor:
    """mgx_bgzf_t: pieces of a byte stream -> BGZF blocks (bgzf_compress, htslib bgzf.c:610, on the device)."""

    def __init__(self, device=0, flags=0):
        self.lib = native.load()
        h = C.c_void_p()
        native.check(self.lib.mgx_bgzf_create(device, flags, C.byref(h)))
        self.h = h

    def compress(self, data, offsets=None, block=MAX_BLOCK_IN):
        """data: bytes-like; offsets: block boundaries (default: every `block` bytes).
        Returns (blocks back to back as a uint8 array, offsets of the blocks in it)."""
        data = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is None:
            offsets = np.arange(0, len(data) + block, block, dtype=np.uint64)
            offsets[-1] = len(data)
            if len(offsets) >= 2 and offsets[-2] == offsets[-1] and len(data):
                offsets = offsets[:-1]
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nb = len(offsets) - 1
        cap = int(self.lib.mgx_bgzf_bound(int(offsets[-1] - offsets[0]), nb))
        out = np.empty(max(cap, 1), dtype=np.uint8)
        out_off = np.zeros(nb + 1, dtype=np.uint64)
        native.check(self.lib.mgx_bgzf_compress(self.h, _ptr(data), _ptr(offsets), nb, _ptr(out), cap, _ptr(out_off)))
        return out[: int(out_off[-1])], out_off

    def stats(self):
        st = native.BgzfStats()
        native.check(self.lib.mgx_bgzf_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in native.BgzfStats._fields_}

    def close(self):
        if self.h:
            self.lib.mgx_bgzf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BgzfBatch:
    """mgx_bgzf_batch_t: a pinned input buffer filled in place, submitted, waited for."""

    def __init__(self, comp, in_capacity, max_blocks):
        self.comp = comp
        self.lib = comp.lib
        b = C.c_void_p()
        native.check(self.lib.mgx_bgzf_batch_create(comp.h, in_capacity, max_blocks, C.byref(b)))
        self.b = b
        self.in_capacity, self.max_blocks = in_capacity, max_blocks
        self.input = np.ctypeslib.as_array(C.cast(self.lib.mgx_bgzf_batch_input(b), C.POINTER(C.c_uint8)), shape=(in_capacity,))
        self.offsets = np.ctypeslib.as_array(C.cast(self.lib.mgx_bgzf_batch_offsets(b), C.POINTER(C.c_uint64)), shape=(max_blocks + 1,))
        self.n_blocks = 0

    def submit(self, n_blocks):
        native.check(self.lib.mgx_bgzf_batch_submit(self.comp.h, self.b, n_blocks))
        self.n_blocks = n_blocks

    def wait(self):
        out, off = C.c_void_p(), C.c_void_p()
        native.check(self.lib.mgx_bgzf_batch_wait(self.comp.h, self.b, C.byref(out), C.byref(off)))
        o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(self.n_blocks + 1,)).copy()
        total = int(o[-1])
        data = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy()
        return data, o

    def close(self):
        if self.b:
            self.lib.mgx_bgzf_batch_destroy(self.comp.h, self.b)
            self.b = None


class BgzfStore:
    """mgx_bgzf_store_t: records resident in HBM (put), emitted as a sorted, duplicate-marked BGZF stream (emit)."""

    def __init__(self, comp):
        self.comp = comp
        self.lib = comp.lib
        h = C.c_void_p()
        native.check(self.lib.mgx_bgzf_store_create(comp.h, C.byref(h)))
        self.h = h

    def put(self, data):
        data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        addr = C.c_uint64()
        native.check(self.lib.mgx_bgzf_store_put(self.h, _ptr(data), len(data), C.byref(addr)))
        return int(addr.value)

    def emit(self, order, dup, addr, length):
        """Returns (all blocks back to back as bytes, compressed offset of every block, uoff[n + 1])."""
        order = np.ascontiguousarray(order, dtype=np.uint32); dup = np.ascontiguousarray(dup, dtype=np.uint8)
        addr = np.ascontiguousarray(addr, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(order)
        uoff = np.zeros(n + 1, dtype=np.uint64)
        parts, block_at = [], []
        total = [0]

        def sink(_user, blocks, n_bytes, n_blocks, block_off):
            parts.append(C.string_at(blocks, n_bytes))
            oo = np.ctypeslib.as_array(C.cast(block_off, C.POINTER(C.c_uint64)), shape=(n_blocks + 1,))
            block_at.extend(int(total[0] + x) for x in oo[:n_blocks])
            total[0] += int(n_bytes)
            return 0
        cb = native.BGZF_SINK(sink)
        native.check(self.lib.mgx_bgzf_store_emit(self.h, n, _ptr(order), _ptr(dup), _ptr(addr), _ptr(length), cb, None, _ptr(uoff)))
        return b"".join(parts), np.array(block_at + [total[0]], dtype=np.uint64), uoff

    def close(self):
        if self.h:
            self.lib.mgx_bgzf_store_destroy(self.h)
            self.h = None
