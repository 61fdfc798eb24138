"""ctypes binding of libmgx.so -- mirrors include/*.h one to one."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MgxError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libmgx.so")


class PairHMMInput(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("read_off", C.c_void_p), ("bases", C.c_void_p),
        ("qual", C.c_void_p), ("ins", C.c_void_p), ("del_", C.c_void_p), ("gcp", C.c_void_p),
        ("n_haps", C.c_uint64), ("hap_off", C.c_void_p), ("hap_bases", C.c_void_p),
        ("n_pairs", C.c_uint64), ("pair_read", C.c_void_p), ("pair_hap", C.c_void_p),
    ]


class PairHMMStats(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint64), ("cells", C.c_uint64), ("alg_bytes", C.c_uint64),
        ("n_rerun_f64", C.c_uint64), ("n_launches_f32", C.c_uint32), ("n_launches_f64", C.c_uint32),
        ("n_runs_timed", C.c_uint32), ("ms_f32", C.c_float), ("ms_f64", C.c_float), ("ms_f32_dominant", C.c_float),
        ("dominant_cells", C.c_uint64), ("dominant_alg_bytes", C.c_uint64),
        ("dominant_kernel", C.c_char * 64), ("n_exact", C.c_uint64),
    ]


class QueueConfig(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("devices", C.c_void_p), ("lanes_per_device", C.c_uint32),
                ("depth", C.c_uint32), ("batch_pairs", C.c_uint32), ("flags", C.c_uint)]


class QueueStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_batches", C.c_uint64), ("cells", C.c_uint64),
                ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64), ("seconds", C.c_double),
                ("pack_seconds", C.c_double), ("wait_seconds", C.c_double), ("n_lanes", C.c_uint32),
                ("batches_per_device", C.c_uint64 * 16)]


class ReadModel(C.Structure):
    _fields_ = [("pcr_rate_factor", C.c_int), ("base_quality_threshold", C.c_int), ("constant_gcp", C.c_int),
                ("log10_mismapping_rate", C.c_double), ("max_error_per_base", C.c_double)]


# every symbol include/mgx_pairhmm.h declares: name -> (restype, argtypes)
PAIRHMM_SYMBOLS = {
    "mgx_last_error": (C.c_char_p, []),
    "mgx_pairhmm_create": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "mgx_pairhmm_destroy": (None, [C.c_void_p]),
    "mgx_pairhmm_compute": (C.c_int, [C.c_void_p, C.POINTER(PairHMMInput), C.c_void_p]),
    "mgx_pairhmm_compute_regions": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_batch_create": (C.c_int, [C.c_void_p, C.POINTER(PairHMMInput), C.POINTER(C.c_void_p)]),
    "mgx_pairhmm_batch_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_batch_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_batch_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PairHMMStats)]),
    "mgx_pairhmm_batch_destroy": (None, [C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_sync": (C.c_int, [C.c_void_p]),
    "mgx_read_model_defaults": (None, [C.c_void_p]),
    "mgx_pairhmm_region": (C.c_int, [C.c_void_p, C.POINTER(PairHMMInput), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_regions": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_queue_create": (C.c_int, [C.POINTER(QueueConfig), C.POINTER(C.c_void_p)]),
    "mgx_pairhmm_queue_destroy": (None, [C.c_void_p]),
    "mgx_pairhmm_queue_run": (C.c_int, [C.c_void_p, C.POINTER(PairHMMInput), C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_queue_run_range": (C.c_int, [C.c_void_p, C.POINTER(PairHMMInput), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_queue_run_regions": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mgx_pairhmm_queue_stats": (C.c_int, [C.c_void_p, C.POINTER(QueueStats)]),
    "mgx_pairhmm_pack_batch": (C.c_int, [C.POINTER(PairHMMInput), C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t,
                                         C.POINTER(PairHMMInput), C.POINTER(C.c_size_t)]),
    "mgx_pairhmm_table_f32": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mgx_pairhmm_table_f64": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
}

SYMBOLS = dict(PAIRHMM_SYMBOLS)


def load():
    """Loads libmgx.so (never falls back to anything else)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise MgxError(
            f"{path} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # The sort pipeline overlaps its three sorts on three HIP streams; the runtime maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on one queue serialise, which is
    # what happens once another context (PairHMM, torch) holds streams in the same process.  Only takes
    # effect if the HIP runtime has not been initialised yet; an explicit setting is respected.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the ABI and the header diverge
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().mgx_last_error()
        raise MgxError(f"libmgx error {rc}: {msg.decode() if msg else ''}")


# ---- include/mgx_sortdedup.h -------------------------------------------------------------------
class RawRecords(C.Structure):
    _fields_ = [
        ("n_records", C.c_uint64), ("flag", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p),
        ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("qual_off", C.c_void_p), ("qual", C.c_void_p),
        ("qname_off", C.c_void_p), ("qname", C.c_void_p), ("n_targets", C.c_uint32), ("target_len", C.c_void_p),
    ]


class SortDedupStats(C.Structure):
    _fields_ = [
        ("n_records", C.c_uint64), ("n_double", C.c_uint64), ("n_single", C.c_uint64), ("n_dup_records", C.c_uint64),
        ("key_bits_coord", C.c_uint32), ("key_bits_pair1", C.c_uint32), ("key_bits_pair2", C.c_uint32),
        ("n_radix_passes", C.c_uint32), ("ms_total", C.c_float), ("ms_radix_scatter", C.c_float),
        ("radix_scatter_bytes", C.c_uint64), ("alg_bytes", C.c_uint64),
        ("ms_scatter_records", C.c_float), ("n_scatter_records", C.c_uint32), ("scatter_records_bytes", C.c_uint64),
        ("n_key_hist_launches", C.c_uint32), ("pad_", C.c_uint32),
    ]


class SortDedupShard(C.Structure):
    _fields_ = [("n_order", C.c_uint64), ("order_coord", C.c_void_p), ("order_arrival", C.c_void_p),
                ("n_mark", C.c_uint64), ("mark_recs", C.c_void_p), ("mark_arrival", C.c_void_p),
                ("n_marks", C.c_uint64), ("marks", C.c_void_p), ("order_base", C.c_uint64),
                ("coord_lo", C.c_uint64), ("coord_hi", C.c_uint64)]


SORTDEDUP_SYMBOLS = {
    "mgx_sortdedup_route": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "mgx_sortdedup_routed_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(SortDedupShard)]),
    "mgx_sortdedup_routed_free": (None, [C.c_void_p]),
    "mgx_sortdedup_upload_shard": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(SortDedupShard)]),
    "mgx_sortdedup_merge": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_sortdedup_pack": (C.c_int, [C.POINTER(RawRecords), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "mgx_sortdedup_pack_scored": (C.c_int, [C.POINTER(RawRecords), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "mgx_sortdedup_create": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "mgx_sortdedup_destroy": (None, [C.c_void_p]),
    "mgx_sortdedup_upload": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "mgx_sortdedup_upload_begin": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "mgx_sortdedup_upload_chunk": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "mgx_sortdedup_upload_end": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mgx_sortdedup_run": (C.c_int, [C.c_void_p]),
    "mgx_sortdedup_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_sortdedup_stats": (C.c_int, [C.c_void_p, C.POINTER(SortDedupStats)]),
    "mgx_sortdedup_sort_mark": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
}
SYMBOLS.update(SORTDEDUP_SYMBOLS)


# ---- include/mgx_smithwaterman.h ---------------------------------------------------------------
class SwParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32)]


class SwInput(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("ref_off", C.c_void_p), ("ref", C.c_void_p), ("alt_off", C.c_void_p),
                ("alt", C.c_void_p), ("strategy", C.c_void_p)]


class SwStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("cells", C.c_uint64), ("n_launches", C.c_uint32), ("ms_fill", C.c_float),
                ("ms_trace", C.c_float), ("backtrace_bytes", C.c_uint64), ("n_pairs_i16", C.c_uint64)]


SMITHWATERMAN_SYMBOLS = {
    "mgx_sw_create": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "mgx_sw_destroy": (None, [C.c_void_p]),
    "mgx_sw_align_batch": (C.c_int, [C.c_void_p, C.POINTER(SwParams), C.POINTER(SwInput), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mgx_sw_align": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint8]),
    "mgx_sw_stats": (C.c_int, [C.c_void_p, C.POINTER(SwStats)]),
}
SYMBOLS.update(SMITHWATERMAN_SYMBOLS)


# ---- include/mgx_bgzf.h ---------------------------------------------------------------------------
class BgzfStats(C.Structure):
    _fields_ = [("n_blocks", C.c_uint64), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64), ("n_stored", C.c_uint64),
                ("ms_kernels", C.c_float), ("ms_pack", C.c_float)]


BGZF_SYMBOLS = {
    "mgx_bgzf_create": (C.c_int, [C.c_int, C.c_uint, C.POINTER(C.c_void_p)]),
    "mgx_bgzf_destroy": (None, [C.c_void_p]),
    "mgx_bgzf_batch_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mgx_bgzf_batch_destroy": (None, [C.c_void_p, C.c_void_p]),
    "mgx_bgzf_batch_input": (C.c_void_p, [C.c_void_p]),
    "mgx_bgzf_batch_offsets": (C.c_void_p, [C.c_void_p]),
    "mgx_bgzf_batch_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "mgx_bgzf_batch_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mgx_bgzf_bound": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "mgx_bgzf_compress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mgx_bgzf_stats": (C.c_int, [C.c_void_p, C.POINTER(BgzfStats)]),
    "mgx_bgzf_prepare": (C.c_int, [C.c_void_p]),
    "mgx_bgzf_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgx_bgzf_store_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mgx_bgzf_store_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgx_bgzf_store_destroy": (None, [C.c_void_p]),
    "mgx_bgzf_store_put": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mgx_bgzf_store_emit": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}
BGZF_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p)
SYMBOLS.update(BGZF_SYMBOLS)
