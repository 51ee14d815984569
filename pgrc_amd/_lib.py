"""ctypes binding of libpgrc_match.so (include/pgrc_match.h).

The shared library is the product: HIP kernels + C ABI.  This module only loads
it and declares the prototypes.  There is no Python / CPU fallback: if the
library is missing, import fails loudly with the build hint.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGRC_MATCH_LIB: another build of the same library (A/B runs of compile-time variants, tools/variants.sh)
LIB_PATH = os.environ.get("PGRC_MATCH_LIB") or os.path.join(_HERE, "libpgrc_match.so")

NOT_MATCHED_POS = 0xFFFFFFFFFFFFFFFF  # DefaultReadsMatcher::NOT_MATCHED_POSITION (ReadsMatchers.cpp:69)
NOT_MATCHED_CNT = 255                 # NOT_MATCHED_COUNT (ReadsMatchers.h:17)

ERR_NAMES = {0: "OK", 1: "E_PARAM", 2: "E_SEED_SHORT", 3: "E_NO_DEVICE", 4: "E_ALLOC",
             5: "E_SYMBOL", 6: "E_STATE", 7: "E_MODE", 8: "E_DEVICE"}


class PgrcMatchError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pgrc_match error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class MatchParams(C.Structure):
    _fields_ = [("read_len", C.c_uint32), ("seed_len", C.c_uint32), ("max_mismatches", C.c_uint8),
                ("min_mismatches", C.c_uint8), ("mode", C.c_char), ("device", C.c_int32)]


class CopmemParams(C.Structure):
    _fields_ = [("K", C.c_int32), ("k1", C.c_int32), ("k2", C.c_int32), ("hash_size", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("searched", C.c_uint64 * 2), ("candidates", C.c_uint64 * 2), ("probes", C.c_uint64 * 2),
                ("entry_fetches", C.c_uint64 * 2), ("verifies", C.c_uint64 * 2), ("index_entries", C.c_uint64 * 2), ("ms_index", C.c_float * 2), ("ms_match", C.c_float * 2),
                ("ms_other", C.c_float), ("ms_total", C.c_float), ("ms_allgather", C.c_float),
                ("screened", C.c_uint32), ("ms_screen", C.c_float), ("redo_reads", C.c_uint64), ("dual", C.c_uint64 * 5),
                ("schedule_downgraded", C.c_uint32), ("dual_seed_probes", C.c_uint64)]


class SynthPg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("pg_len", C.c_uint64), ("grid", C.c_uint32), ("plant_len", C.c_uint32),
                ("pool_div", C.c_uint32), ("tandem_every", C.c_uint32)]


class SynthReads(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n", C.c_uint64), ("read_len", C.c_uint32), ("paired", C.c_uint32),
                ("n_with_n", C.c_uint64)]


class ExportStreams(C.Structure):   # pgrc_export_streams
    _fields_ = [("n_entries", C.c_uint64), ("n_mismatches", C.c_uint64), ("off_width", C.c_uint32), ("off", C.POINTER(C.c_uint8)),
                ("org_idx", C.POINTER(C.c_uint32)), ("rev_comp", C.POINTER(C.c_uint8)), ("mis_cnt", C.POINTER(C.c_uint8)),
                ("mis_sym", C.POINTER(C.c_uint8)), ("mis_rev_off", C.POINTER(C.c_uint8)), ("last_pos", C.c_uint64)]


class ExportPgOrderArgs(C.Structure):   # pgrc_export_pg_order_args
    _fields_ = [("order", C.c_void_p), ("n_matched", C.c_uint64), ("read_org_idx", C.c_void_p), ("list_off", C.c_void_p),
                ("list_org_idx", C.c_void_p), ("list_rev_comp", C.c_void_p), ("list_count", C.c_uint64),
                ("rev_compl_pair_file", C.c_int32), ("byte_per_read_length", C.c_int32), ("order_on_device", C.c_int32)]


class ExportOriginalOrderArgs(C.Structure):   # pgrc_export_original_order_args
    _fields_ = [("read_org_idx", C.c_void_p), ("reads_total_count", C.c_uint64), ("pair_file_mode", C.c_int32),
                ("rev_compl_pair_file", C.c_int32), ("byte_per_read_length", C.c_int32)]


class TextMatch(C.Structure):     # pgrc_text_match (include/pgrc_mem.h) = TextMatch, matching/TextMatchers.h:10-16
    _fields_ = [("pos_src", C.c_uint64), ("length", C.c_uint64), ("pos_dest", C.c_uint64)]


class DivideParams(C.Structure):
    _fields_ = [("read_len", C.c_uint32), ("error_limit", C.c_double), ("simplified_suffix_mode", C.c_int32),
                ("separate_n_reads_set", C.c_int32), ("n_reads_lq", C.c_int32), ("device", C.c_int32)]


class DividedReads(C.Structure):
    _fields_ = [("n_hq", C.c_uint64), ("n_lq", C.c_uint64), ("n_n", C.c_uint64),
                ("hq_symbols", C.c_uint32), ("lq_symbols", C.c_uint32), ("n_symbols", C.c_uint32),
                ("hq_row_bytes", C.c_uint32), ("lq_row_bytes", C.c_uint32), ("n_row_bytes", C.c_uint32),
                ("hq_rows", C.c_void_p), ("lq_rows", C.c_void_p), ("n_rows", C.c_void_p),
                ("lq_index", C.c_void_p), ("n_index", C.c_void_p)]


class MemCounters(C.Structure):
    _fields_ = [("probes", C.c_uint64), ("events", C.c_uint64), ("stale_lookups", C.c_uint64), ("ms_index", C.c_float),
                ("ms_probe", C.c_float), ("ms_sort", C.c_float), ("ms_extend", C.c_float), ("ms_host", C.c_float),
                ("ms_replay", C.c_float), ("replay_rounds", C.c_uint32), ("event_blocks", C.c_uint32)]


# every symbol include/pgrc_match.h and include/pgrc_mem.h declare: (name, restype, argtypes)
_P = C.c_void_p
_PROTOS = [
    ("pgrc_match_version", C.c_char_p, []),
    ("pgrc_match_derive_params", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_char, C.POINTER(MatchParams)]),
    ("pgrc_match_create", C.c_int, [C.POINTER(MatchParams), C.POINTER(_P)]),
    ("pgrc_match_create_multi", C.c_int, [C.POINTER(MatchParams), C.c_int32, C.POINTER(C.c_int32), C.POINTER(_P)]),
    ("pgrc_match_device_count", C.c_int, [C.POINTER(C.c_int32)]),
    ("pgrc_match_shard_count", C.c_int32, [_P]),
    ("pgrc_match_shard_info", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pgrc_match_destroy", None, [_P]),
    ("pgrc_match_last_error", C.c_char_p, [_P]),
    ("pgrc_match_set_stream", C.c_int, [_P, _P]),
    ("pgrc_match_set_pg_ascii", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_match_set_pg_packed_device", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_match_pack_pg_slice", C.c_int, [_P, _P, C.c_uint64, _P]),
    ("pgrc_match_set_reads_ascii", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_match_begin_reads", C.c_int, [_P, C.c_uint64]),
    ("pgrc_match_append_reads_ascii", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_match_end_reads", C.c_int, [_P]),
    ("pgrc_match_append_reads_packed", C.c_int, [_P, _P, C.c_uint64, C.c_int32]),
    ("pgrc_match_set_reads_packed", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_match_set_reads_device", C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    ("pgrc_match_words_per_read", C.c_uint32, [C.c_uint32]),
    ("pgrc_match_init_results", C.c_int, [_P]),
    ("pgrc_match_set_results", C.c_int, [_P, _P, _P, _P]),
    ("pgrc_match_run", C.c_int, [_P, C.c_int]),
    ("pgrc_match_run_pass", C.c_int, [_P, C.c_int]),
    ("pgrc_match_get_results", C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_uint64)]),
    ("pgrc_match_get_results_device", C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("pgrc_match_extract_mismatches", C.c_int, [_P, _P, _P, _P, _P]),
    ("pgrc_match_get_redo_flags", C.c_int, [_P, _P]),
    ("pgrc_match_trim_device_memory", C.c_uint64, []),
    ("pgrc_match_prepare_index", C.c_int, [_P, C.c_int32]),
    ("pgrc_match_stream_begin", C.c_int, [_P, _P, _P, _P]),
    ("pgrc_match_stream_end", C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    ("pgrc_match_export_pg_order", C.c_int, [_P, C.POINTER(ExportPgOrderArgs), C.POINTER(ExportStreams)]),
    ("pgrc_match_export_entries", C.c_int, [_P, _P, _P, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(ExportStreams)]),
    ("pgrc_match_export_original_order", C.c_int, [_P, C.POINTER(ExportOriginalOrderArgs), C.POINTER(ExportStreams)]),
    ("pgrc_match_free_export", None, [C.POINTER(ExportStreams)]),
    ("pgrc_match_copmem_params", C.c_int, [C.c_uint32, C.c_uint64, C.POINTER(CopmemParams)]),
    ("pgrc_match_export_index", C.c_int, [_P, C.c_int, _P, _P, C.POINTER(C.c_uint64)]),
    ("pgrc_match_export_pg", C.c_int, [_P, C.c_int, _P]),
    ("pgrc_match_reload_options", C.c_int, [_P]),
    ("pgrc_match_set_profiling", C.c_int, [_P, C.c_int]),
    ("pgrc_match_get_counters", C.c_int, [_P, C.POINTER(Counters)]),
    ("pgrc_match_get_counters_sized", C.c_int, [_P, _P, C.c_size_t]),
    ("pgrc_synth_pg_host", None, [C.POINTER(SynthPg), _P]),
    ("pgrc_synth_reads_host", None, [C.POINTER(SynthPg), _P, C.POINTER(SynthReads), C.c_uint64, C.c_uint64, _P]),
    ("pgrc_synth_pg_device", C.c_int, [C.POINTER(SynthPg), _P, _P]),
    ("pgrc_synth_reads_device", C.c_int, [C.POINTER(SynthPg), _P, C.POINTER(SynthReads), C.c_uint64, C.c_uint64,
                                          _P, C.c_uint64, _P]),
    # include/pgrc_mem.h
    ("pgrc_mem_create", C.c_int, [C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(_P)]),
    ("pgrc_mem_destroy", None, [_P]),
    ("pgrc_mem_last_error", C.c_char_p, [_P]),
    ("pgrc_mem_set_src_ascii", C.c_int, [_P, _P, C.c_uint64]),
    ("pgrc_mem_match_texts", C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.POINTER(TextMatch)),
                                       C.POINTER(C.c_uint64)]),
    ("pgrc_mem_free_matches", None, [C.POINTER(TextMatch)]),
    ("pgrc_mem_get_counters", C.c_int, [_P, C.POINTER(MemCounters)]),
    # include/pgrc_reads.h
    ("pgrc_divider_create", C.c_int, [C.POINTER(DivideParams), C.POINTER(_P)]),
    ("pgrc_divider_destroy", None, [_P]),
    ("pgrc_divider_last_error", C.c_char_p, [_P]),
    ("pgrc_divider_run", C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(DividedReads)]),
    ("pgrc_divider_run_fastq", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(DividedReads)]),
    ("pgrc_divider_last_ms", C.c_int, [_P, C.POINTER(C.c_float * 3)]),
    ("pgrc_divider_last_was_terminal", C.c_int, [_P]),
]

EXPORTED_SYMBOLS = [p[0] for p in _PROTOS]


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's).  If our library pulled in the system copy first and
    torch its bundled copy later, two runtimes would fight over the device (the second one reports no
    devices).  Loading torch's copy first -- without importing torch -- makes both resolve to it.
    Processes without torch (e.g. the C++ reference with the adapter) use the system runtime."""
    if os.environ.get("PGRC_USE_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:  # torch absent or unloadable: fall through to the system runtime
        pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product and there is no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pgrc_amd/csrc`.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, res, args in _PROTOS:
        fn = getattr(lib, name)  # AttributeError here = header / library out of sync: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
