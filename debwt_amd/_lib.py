"""ctypes loader of libdebwt_hip.so (include/debwt_hip.h).  There is no CPU fallback: if the HIP
library is missing or cannot be loaded, every entry point raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEBWT_HIP_LIB", os.path.join(_HERE, "libdebwt_hip.so"))   # override: kernel experiments only
_lib = None


class DebwtConfig(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int), ("device", ctypes.c_int), ("sort_algo", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class DebwtPackedText(ctypes.Structure):
    _fields_ = [("words", ctypes.POINTER(ctypes.c_uint64)), ("nwords", ctypes.c_uint64), ("n", ctypes.c_uint64),
                ("sep", ctypes.POINTER(ctypes.c_uint64)), ("nrec", ctypes.c_uint64),
                ("seconds_read", ctypes.c_double), ("seconds_pack", ctypes.c_double)]


class DebwtStats(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_uint64) for n in (
        "n", "nrec", "red_capacity", "blue_capacity", "blue_bound_num", "case3num", "sp_len",
        "special_branch_num", "n_main", "distinct_keys", "blue_large_blocks", "blue_max_block")] +
        [(n, ctypes.c_float) for n in (
            "ms_extract", "ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total",
            "ms_host_special")] +
        [("radix_pass_launches", ctypes.c_uint32), ("radix_pass_ms", ctypes.c_float),
         ("radix_pass_keys", ctypes.c_uint64), ("special_path", ctypes.c_uint32), ("special_threads", ctypes.c_uint32),
         ("sort_unfit_stretches", ctypes.c_uint64), ("sort_unfit_network", ctypes.c_uint64),
         ("sort_over_stretches", ctypes.c_uint64)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class DebwtVerifyReport(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_uint64) for n in ("segments", "steps", "mismatches", "broken_links", "search_failures",
                                                "search_steps")] +
                [(n, ctypes.c_float) for n in ("ms_index", "ms_search", "ms_walk")] + [("ok", ctypes.c_int)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


MULTI_STEPS = 24


class DebwtShardReport(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_float * MULTI_STEPS), ("bytes_in", ctypes.c_uint64 * 5), ("bytes_out", ctypes.c_uint64 * 5),
                ("keys", ctypes.c_uint64), ("key_ranges", ctypes.c_uint64), ("blocks", ctypes.c_uint64),
                ("blue_rows", ctypes.c_uint64), ("rows", ctypes.c_uint64), ("bin_lo", ctypes.c_uint32), ("bin_hi", ctypes.c_uint32)]


class DebwtMultiStats(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("nrec", ctypes.c_uint64), ("ngpus", ctypes.c_uint32), ("rounds", ctypes.c_uint32),
                ("key_bytes_in", ctypes.c_uint64), ("blue_bytes_in", ctypes.c_uint64), ("ms_build", ctypes.c_float),
                ("key_mode", ctypes.c_uint32), ("exchange_backend", ctypes.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/debwt_hip.h declares
SYMBOLS = [
    "debwt_create", "debwt_destroy", "debwt_strerror", "debwt_last_error", "debwt_load_text",
    "debwt_load_ascii", "debwt_kmer_sort_rle", "debwt_classify", "debwt_sp_generate", "debwt_blue_sort",
    "debwt_bwt_assemble", "debwt_build", "debwt_fetch_bwt", "debwt_bwt_device_ptr", "debwt_get_stats",
    "debwt_fetch_array", "debwt_kmer_count_sorted", "debwt_radix_sort_u64", "debwt_verify_inverse",
    "debwt_shard_begin", "debwt_shard_histogram", "debwt_shard_set_range", "debwt_shard_classify_local",
    "debwt_shard_facts_export", "debwt_shard_classify_global", "debwt_shard_info", "debwt_shard_fetch",
    "debwt_shard_partition_keys", "debwt_shard_plan", "debwt_shard_ranges", "debwt_shard_sort_begin",
    "debwt_shard_sort_range", "debwt_shard_sort_end", "debwt_concat_rows", "debwt_shard_export", "debwt_census_words", "debwt_shard_sp_flags", "debwt_shard_sp_emit",
    "debwt_shard_sp_import", "debwt_shard_blue_route", "debwt_shard_blue_place", "debwt_set_range_cap", "debwt_pack_fasta", "debwt_free_packed", "debwt_load_fasta", "debwt_pack_fasta_opts", "debwt_load_fasta_opts", "debwt_fasta_text_bound", "debwt_host_release_hold", "debwt_special_digest", "debwt_bwt_census", "debwt_fetch_rows", "debwt_verify_device", "debwt_multi_create", "debwt_multi_destroy", "debwt_multi_last_error",
    "debwt_multi_load_text", "debwt_multi_load_fasta", "debwt_multi_build", "debwt_multi_fetch_bwt", "debwt_multi_get_stats",
    "debwt_multi_verify", "debwt_multi_shard", "debwt_pinned_alloc", "debwt_pinned_free", "debwt_shard_key_mode",
    "debwt_multi_set_key_mode", "debwt_special_compare", "debwt_build_to_host", "debwt_multi_set_exchange", "debwt_reserve",
    "debwt_multi_set_serial", "debwt_multi_get_shard_report", "debwt_multi_step_name", "debwt_get_config",
    "debwt_dump_reference_files", "debwt_shard_scratch",
]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src_dir]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C debwt_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (its bundled HIP runtime must be the one in the process, see __graft_entry__.build)
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)
    L.debwt_create.restype = ctypes.c_int
    L.debwt_create.argtypes = [ctypes.POINTER(DebwtConfig), ctypes.POINTER(vp)]
    L.debwt_destroy.restype = None
    L.debwt_destroy.argtypes = [vp]
    L.debwt_strerror.restype = ctypes.c_char_p
    L.debwt_strerror.argtypes = [ctypes.c_int]
    L.debwt_last_error.restype = ctypes.c_char_p
    L.debwt_last_error.argtypes = [vp]
    L.debwt_load_text.restype = ctypes.c_int
    L.debwt_load_text.argtypes = [vp, u64p, ctypes.c_uint64, u64p, ctypes.c_uint64]
    L.debwt_load_ascii.restype = ctypes.c_int
    L.debwt_load_ascii.argtypes = [vp, ctypes.c_char_p, u64p, ctypes.c_uint64]
    for name in ("debwt_kmer_sort_rle", "debwt_classify", "debwt_sp_generate", "debwt_blue_sort",
                 "debwt_bwt_assemble", "debwt_build"):
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp]
    L.debwt_fetch_bwt.restype = ctypes.c_int
    L.debwt_fetch_bwt.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_reserve.restype = ctypes.c_int
    L.debwt_reserve.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint]
    L.debwt_build_to_host.restype = ctypes.c_int
    L.debwt_build_to_host.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_bwt_device_ptr.restype = ctypes.c_int
    L.debwt_bwt_device_ptr.argtypes = [vp, ctypes.POINTER(vp)]
    L.debwt_get_stats.restype = ctypes.c_int
    L.debwt_get_stats.argtypes = [vp, ctypes.POINTER(DebwtStats)]
    L.debwt_fetch_array.restype = ctypes.c_int
    L.debwt_fetch_array.argtypes = [vp, ctypes.c_int, vp, ctypes.c_uint64, u64p]
    L.debwt_kmer_count_sorted.restype = ctypes.c_int
    L.debwt_kmer_count_sorted.argtypes = [vp, u64p, u64p, ctypes.c_uint64, u64p]
    L.debwt_radix_sort_u64.restype = ctypes.c_int
    L.debwt_radix_sort_u64.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.debwt_verify_inverse.restype = ctypes.c_int
    L.debwt_verify_inverse.argtypes = [u64p, ctypes.c_uint64, u64p, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint8)]
    L.debwt_shard_begin.restype = ctypes.c_int
    L.debwt_shard_begin.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.debwt_shard_histogram.restype = ctypes.c_int
    L.debwt_shard_histogram.argtypes = [vp, u64p]
    L.debwt_shard_set_range.restype = ctypes.c_int
    L.debwt_shard_set_range.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64]
    L.debwt_shard_classify_local.restype = ctypes.c_int
    L.debwt_shard_classify_local.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_shard_facts_export.restype = ctypes.c_int
    L.debwt_shard_facts_export.argtypes = [vp, vp, ctypes.c_uint64]
    L.debwt_shard_classify_global.restype = ctypes.c_int
    L.debwt_shard_classify_global.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    u8p, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32)
    L.debwt_shard_partition_keys.restype = ctypes.c_int
    L.debwt_shard_partition_keys.argtypes = [vp, u8p, vp, ctypes.c_uint64, u64p]
    L.debwt_shard_plan.restype = ctypes.c_int
    L.debwt_shard_plan.argtypes = [vp, u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int,
                                   ctypes.c_uint64, u32p]
    L.debwt_shard_key_mode.restype = ctypes.c_int
    L.debwt_shard_key_mode.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double)]
    L.debwt_multi_set_key_mode.restype = ctypes.c_int
    L.debwt_multi_set_key_mode.argtypes = [vp, ctypes.c_int]
    L.debwt_multi_set_exchange.restype = ctypes.c_int
    L.debwt_multi_set_exchange.argtypes = [vp, ctypes.c_int]
    L.debwt_shard_ranges.restype = ctypes.c_int
    L.debwt_shard_ranges.argtypes = [vp, u32p, u64p, ctypes.c_uint32]
    L.debwt_shard_sort_begin.restype = ctypes.c_int
    L.debwt_shard_sort_begin.argtypes = [vp]
    L.debwt_shard_sort_range.restype = ctypes.c_int
    L.debwt_shard_sort_range.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint64]
    L.debwt_shard_sort_end.restype = ctypes.c_int
    L.debwt_shard_sort_end.argtypes = [vp]
    L.debwt_shard_export.restype = ctypes.c_int
    L.debwt_shard_export.argtypes = [vp, vp, ctypes.c_uint64]
    L.debwt_census_words.restype = ctypes.c_int
    L.debwt_census_words.argtypes = [vp, vp, ctypes.c_uint64, u64p]
    L.debwt_concat_rows.restype = ctypes.c_int
    L.debwt_concat_rows.argtypes = [vp, vp, ctypes.c_uint32, u64p, u64p, u64p, ctypes.c_uint64, vp]
    L.debwt_shard_sp_flags.restype = ctypes.c_int
    L.debwt_shard_sp_flags.argtypes = [vp, u64p, u64p]
    L.debwt_shard_sp_emit.restype = ctypes.c_int
    L.debwt_shard_sp_emit.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    L.debwt_shard_sp_import.restype = ctypes.c_int
    L.debwt_shard_sp_import.argtypes = [vp, vp, ctypes.c_uint64]
    L.debwt_shard_blue_route.restype = ctypes.c_int
    L.debwt_shard_blue_route.argtypes = [vp, u32p, vp, ctypes.c_uint64, u64p]
    L.debwt_shard_blue_place.restype = ctypes.c_int
    L.debwt_shard_blue_place.argtypes = [vp, vp, ctypes.c_uint64]
    L.debwt_shard_info.restype = ctypes.c_int
    L.debwt_shard_info.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_shard_fetch.restype = ctypes.c_int
    L.debwt_shard_fetch.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_pack_fasta.restype = ctypes.c_int
    L.debwt_pack_fasta.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(DebwtPackedText), ctypes.c_char_p,
                                   ctypes.c_size_t]
    L.debwt_free_packed.restype = None
    L.debwt_free_packed.argtypes = [ctypes.POINTER(DebwtPackedText)]
    L.debwt_load_fasta.restype = ctypes.c_int
    L.debwt_load_fasta.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    L.debwt_host_release_hold.restype = None
    L.debwt_host_release_hold.argtypes = [ctypes.c_int]
    L.debwt_fasta_text_bound.restype = ctypes.c_uint64
    L.debwt_fasta_text_bound.argtypes = [ctypes.c_char_p]
    L.debwt_pack_fasta_opts.restype = ctypes.c_int
    L.debwt_pack_fasta_opts.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64,
                                        ctypes.POINTER(DebwtPackedText), ctypes.c_char_p, ctypes.c_size_t]
    L.debwt_load_fasta_opts.restype = ctypes.c_int
    L.debwt_load_fasta_opts.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64]
    L.debwt_special_compare.restype = ctypes.c_int
    L.debwt_special_compare.argtypes = [vp, u64p]
    L.debwt_special_digest.restype = ctypes.c_int
    L.debwt_special_digest.argtypes = [u64p, ctypes.c_uint64, u64p, ctypes.c_uint64, ctypes.c_int, u64p]
    L.debwt_multi_create.restype = ctypes.c_int
    L.debwt_multi_create.argtypes = [ctypes.POINTER(DebwtConfig), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.debwt_multi_destroy.restype = None
    L.debwt_multi_destroy.argtypes = [vp]
    L.debwt_multi_last_error.restype = ctypes.c_char_p
    L.debwt_multi_last_error.argtypes = [vp]
    L.debwt_multi_load_text.restype = ctypes.c_int
    L.debwt_multi_load_text.argtypes = [vp, u64p, ctypes.c_uint64, u64p, ctypes.c_uint64]
    L.debwt_multi_load_fasta.restype = ctypes.c_int
    L.debwt_multi_load_fasta.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64]
    L.debwt_multi_build.restype = ctypes.c_int
    L.debwt_multi_build.argtypes = [vp]
    L.debwt_multi_fetch_bwt.restype = ctypes.c_int
    L.debwt_multi_fetch_bwt.argtypes = [vp, u64p, u64p, u64p]
    L.debwt_multi_get_stats.restype = ctypes.c_int
    L.debwt_multi_get_stats.argtypes = [vp, ctypes.POINTER(DebwtMultiStats), ctypes.POINTER(DebwtStats)]
    L.debwt_multi_shard.restype = vp
    L.debwt_multi_shard.argtypes = [vp, ctypes.c_int]
    L.debwt_multi_set_serial.restype = ctypes.c_int
    L.debwt_multi_set_serial.argtypes = [vp, ctypes.c_int]
    L.debwt_multi_get_shard_report.restype = ctypes.c_int
    L.debwt_multi_get_shard_report.argtypes = [vp, ctypes.c_int, ctypes.POINTER(DebwtShardReport)]
    L.debwt_multi_step_name.restype = ctypes.c_char_p
    L.debwt_multi_step_name.argtypes = [ctypes.c_int]
    L.debwt_multi_verify.restype = ctypes.c_int
    L.debwt_multi_verify.argtypes = [vp, ctypes.POINTER(DebwtVerifyReport)]
    L.debwt_verify_device.restype = ctypes.c_int
    L.debwt_verify_device.argtypes = [vp, vp, u64p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(DebwtVerifyReport)]
    L.debwt_fetch_rows.restype = ctypes.c_int
    L.debwt_fetch_rows.argtypes = [vp, u64p, u64p]
    L.debwt_bwt_census.restype = ctypes.c_int
    L.debwt_bwt_census.argtypes = [vp, u64p]
    L.debwt_set_range_cap.restype = ctypes.c_int
    L.debwt_set_range_cap.argtypes = [vp, ctypes.c_uint64]
    L.debwt_shard_scratch.restype = ctypes.c_int
    L.debwt_shard_scratch.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), u64p]
    L.debwt_get_config.restype = ctypes.c_int
    L.debwt_get_config.argtypes = [vp, ctypes.POINTER(DebwtConfig)]
    L.debwt_dump_reference_files.restype = ctypes.c_int
    L.debwt_dump_reference_files.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    _lib = L
    return L
