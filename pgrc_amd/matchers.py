"""Host-side mirror of the reference's matcher classes over the C ABI.

Names, argument meaning and error behaviour follow matching/ReadsMatchers.h:
`DefaultReadsMatcher.matchConstantLengthReads()` (ReadsMatchers.cpp:162-172), the result fields
`readMatchPos / readMatchRC / readMismatchesCount / matchedReadsCount /
matchedCountPerMismatches` (ReadsMatchers.h:32-35,115-116), `getMatchedReadsBitmap`
(:677-691), `continueMatchingConstantLengthReads` (:174-184) and the factory `mapReadsIntoPg`
(:693-796).  All compute happens in libpgrc_match.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import Counters, CopmemParams, MatchParams, PgrcMatchError, lib

DISABLED_PREFIX_MODE = 0xFFFF  # (uint_read_len_max) -1, ReadsMatchers.cpp:68


def _as_ascii_2d(reads, read_len: Optional[int] = None) -> np.ndarray:
    """reads -> contiguous uint8 [n, L] (accepts list[str], bytes rows, or an ndarray)."""
    if isinstance(reads, np.ndarray):
        a = np.ascontiguousarray(reads, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("reads array must be [n, read_len] uint8 ASCII")
        return a
    rows = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    if not rows:
        return np.zeros((0, read_len or 0), dtype=np.uint8)
    L = len(rows[0])
    if any(len(r) != L for r in rows):
        raise ValueError("Unsupported variable length reads.")  # PackedConstantLengthReadsSet.cpp:37-40
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), L).copy()


def _as_ascii_1d(pg) -> np.ndarray:
    if isinstance(pg, np.ndarray):
        return np.ascontiguousarray(pg, dtype=np.uint8).reshape(-1)
    if isinstance(pg, str):
        pg = pg.encode()
    return np.frombuffer(bytes(pg), dtype=np.uint8).copy()


def copmem_params(seed_len: int, pg_len: int) -> dict:
    """CopMEMMatcher::initParams (CopMEMMatcher.cpp:69-96, :111-137)."""
    p = CopmemParams()
    rc = lib.pgrc_match_copmem_params(seed_len, pg_len, C.byref(p))
    if rc:
        raise PgrcMatchError(rc, "copMEM parameter derivation")
    return {"K": p.K, "k1": p.k1, "k2": p.k2, "hash_size": p.hash_size}


class MatchContext:
    """RAII wrapper of `pgrc_match_ctx*`."""

    def __init__(self, read_len: int, seed_len: int, max_mismatches: int, min_mismatches: int, mode: str,
                 device: int = -1, devices=None):
        """devices: a list of HIP device ordinals -> ONE matcher over several GPUs in this process
        (pgrc_match_create_multi: reads sharded, text all-gathered); a device may be listed twice to rehearse the
        sharded path on a smaller box."""
        prm = MatchParams(read_len, seed_len, max_mismatches, min_mismatches, mode.encode(), device)
        self._h = C.c_void_p()
        if devices is None:
            rc = lib.pgrc_match_create(C.byref(prm), C.byref(self._h))
        else:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = lib.pgrc_match_create_multi(C.byref(prm), len(devices), arr, C.byref(self._h))
        if rc:
            self._h = C.c_void_p()
            raise PgrcMatchError(rc, "pgrc_match_create: " + (lib.pgrc_match_last_error(None) or b"").decode())
        self.read_len, self.seed_len, self.mode = read_len, seed_len, mode
        self.n = 0
        self.pg_len = 0
        self._keep = []  # keeps borrowed device tensors alive

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib.pgrc_match_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _ck(self, rc: int):
        if rc:
            raise PgrcMatchError(rc, (lib.pgrc_match_last_error(self._h) or b"").decode())

    # ---- inputs
    def set_stream(self, stream_ptr: int):
        self._ck(lib.pgrc_match_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_pg_ascii(self, pg):
        a = _as_ascii_1d(pg)
        self._ck(lib.pgrc_match_set_pg_ascii(self._h, a.ctypes.data_as(C.c_void_p), a.size))
        self.pg_len = int(a.size)

    def set_pg_packed_device(self, dev_ptr: int, pg_len: int, keep=None):
        self._ck(lib.pgrc_match_set_pg_packed_device(self._h, C.c_void_p(dev_ptr), pg_len))
        self.pg_len = int(pg_len)

    def pack_pg_slice(self, pg_slice, dev_out_ptr: int):
        a = _as_ascii_1d(pg_slice)
        self._ck(lib.pgrc_match_pack_pg_slice(self._h, a.ctypes.data_as(C.c_void_p), a.size, C.c_void_p(dev_out_ptr)))

    def set_reads_ascii(self, reads):
        a = _as_ascii_2d(reads, self.read_len)
        if a.shape[0] and a.shape[1] != self.read_len:
            raise ValueError("Unsupported variable length reads.")
        self._ck(lib.pgrc_match_set_reads_ascii(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0]))
        self.n = int(a.shape[0])

    def set_reads_packed(self, packed: np.ndarray, n: int):
        a = np.ascontiguousarray(packed, dtype=np.uint8)
        self._ck(lib.pgrc_match_set_reads_packed(self._h, a.ctypes.data_as(C.c_void_p), n))
        self.n = int(n)

    def set_reads_packed_sets(self, sets):
        """sets: [(packed uint8 rows, n, symbols)], symbols 4 = an ACGT set, 5 = an ACGNT set, in read order -- e.g. the
        LQ + N sum set of pgrc-encoder.cpp:349-352 as [(lq_rows, n_lq, 4), (n_rows, n_n, 5)]."""
        total = sum(int(n) for _, n, _ in sets)
        self._ck(lib.pgrc_match_begin_reads(self._h, total))
        for rows, n, symbols in sets:
            a = np.ascontiguousarray(rows, dtype=np.uint8)
            self._ck(lib.pgrc_match_append_reads_packed(self._h, a.ctypes.data_as(C.c_void_p), int(n), int(symbols)))
        self._ck(lib.pgrc_match_end_reads(self._h))
        self.n = total

    # ---- pipelined hand-over (pgrc_amd/csrc/stream.hip)
    def prepare_index(self, both_strands: bool = True):
        """start both strands' index builds now (beneath the upload of the reads); the next two-strand run uses them"""
        self._ck(lib.pgrc_match_prepare_index(self._h, 1 if both_strands else 0))

    def match_streamed(self, sets, blocks: int = 1, out=None):
        """The whole job with its steps overlapped: `sets` as in set_reads_packed_sets, or [(ascii rows, n, 0)]; every
        appended block is matched while the next one is copied, its results land in the returned arrays as they
        exist.  blocks > 1 cuts every set into that many append calls.  Returns (pos, rc, mism, hist, matched) -- what
        init_results() + run(True) + get_results() give."""
        total = sum(int(n) for _, n, _ in sets)
        pos, rc, mism = out if out is not None else (np.empty(total, dtype=np.uint64), np.empty(total, dtype=np.uint8), np.empty(total, dtype=np.uint8))
        assert pos.size == total and rc.size == total and mism.size == total and pos.dtype == np.uint64 and rc.dtype == np.uint8 and mism.dtype == np.uint8
        self._ck(lib.pgrc_match_begin_reads(self._h, total))
        self._ck(lib.pgrc_match_stream_begin(self._h, pos.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p),
                                             mism.ctypes.data_as(C.c_void_p)))
        for rows, n, symbols in sets:
            a = np.ascontiguousarray(rows, dtype=np.uint8)
            n = int(n)
            rb = a.size // n if n else 0
            a = a.reshape(n, rb) if n else a
            step = max(1, -(-n // blocks))
            for lo in range(0, n, step):
                part = np.ascontiguousarray(a[lo:lo + step])
                if symbols:
                    self._ck(lib.pgrc_match_append_reads_packed(self._h, part.ctypes.data_as(C.c_void_p), part.shape[0], int(symbols)))
                else:
                    self._ck(lib.pgrc_match_append_reads_ascii(self._h, part.ctypes.data_as(C.c_void_p), part.shape[0]))
        self._ck(lib.pgrc_match_end_reads(self._h))
        self.n = total
        hist = (C.c_uint64 * 256)()
        matched = C.c_uint64()
        self._ck(lib.pgrc_match_stream_end(self._h, hist, C.byref(matched)))
        return pos, rc, mism, np.array(hist, dtype=np.uint64), int(matched.value)

    def shards(self):
        """[(device, first_read, n_reads)] -- one entry for a single-device context"""
        out = []
        for k in range(lib.pgrc_match_shard_count(self._h)):
            d, a, b = C.c_int32(), C.c_uint64(), C.c_uint64()
            self._ck(lib.pgrc_match_shard_info(self._h, k, C.byref(d), C.byref(a), C.byref(b)))
            out.append((d.value, a.value, b.value))
        return out

    def set_reads_device(self, dev_ptr: int, n: int, stride: int, keep=None):
        self._ck(lib.pgrc_match_set_reads_device(self._h, C.c_void_p(dev_ptr), n, stride))
        self.n = int(n)
        if keep is not None:
            self._keep.append(keep)

    # ---- matching
    def init_results(self):
        self._ck(lib.pgrc_match_init_results(self._h))

    def set_results(self, pos, rc, mism):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        rc = np.ascontiguousarray(rc, dtype=np.uint8)
        mism = np.ascontiguousarray(mism, dtype=np.uint8)
        self._ck(lib.pgrc_match_set_results(self._h, pos.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p),
                                            mism.ctypes.data_as(C.c_void_p)))

    def run(self, rev_compl_pg: bool = True):
        self._ck(lib.pgrc_match_run(self._h, 1 if rev_compl_pg else 0))

    def run_pass(self, strand: int):
        """one executeMatching(revCompMode) (ReadsMatchers.h:46)"""
        self._ck(lib.pgrc_match_run_pass(self._h, int(strand)))

    def get_results(self, arrays: bool = True, out=None):
        """out = (pos, rc, mism): existing arrays to fill (the reference's matcher holds its result vectors before the run)"""
        n = self.n
        hist = np.zeros(256, dtype=np.uint64)
        matched = C.c_uint64(0)
        if arrays:
            pos, rc, mism = out if out is not None else (np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8))
            assert pos.size == n and rc.size == n and mism.size == n and pos.dtype == np.uint64 and rc.dtype == np.uint8 and mism.dtype == np.uint8
            self._ck(lib.pgrc_match_get_results(self._h, pos.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p),
                                                mism.ctypes.data_as(C.c_void_p), hist.ctypes.data_as(C.c_void_p),
                                                C.byref(matched)))
        else:
            pos = rc = mism = None
            self._ck(lib.pgrc_match_get_results(self._h, None, None, None, hist.ctypes.data_as(C.c_void_p),
                                                C.byref(matched)))
        return pos, rc, mism, hist, int(matched.value)

    def results_device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(lib.pgrc_match_get_results_device(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def extract_mismatches(self, reversed_flags=None):
        """-> (cum[n+1] u64, codes u8, offsets u16), see pgrc_match_extract_mismatches."""
        n = self.n
        cum = np.zeros(n + 1, dtype=np.uint64)
        fl = None
        flp = None
        if reversed_flags is not None:
            fl = np.ascontiguousarray(reversed_flags, dtype=np.uint8)
            flp = fl.ctypes.data_as(C.c_void_p)
        self._ck(lib.pgrc_match_extract_mismatches(self._h, flp, cum.ctypes.data_as(C.c_void_p), None, None))
        total = int(cum[n])
        codes = np.zeros(total, dtype=np.uint8)
        offs = np.zeros(total, dtype=np.uint16)
        if total:
            self._ck(lib.pgrc_match_extract_mismatches(self._h, flp, cum.ctypes.data_as(C.c_void_p),
                                                       codes.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p)))
        return cum, codes, offs

    # ---- export of the matches as reads-list streams (row f1)
    @staticmethod
    def _streams(st):
        n, m, w = int(st.n_entries), int(st.n_mismatches), int(st.off_width)
        ot = np.uint8 if w == 1 else np.uint16

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,))
            return raw.view(dtype).copy()
        out = {"off": arr(st.off, n, ot), "org_idx": arr(st.org_idx, n, np.uint32), "rev_comp": arr(st.rev_comp, n, np.uint8),
               "mis_cnt": arr(st.mis_cnt, n, np.uint8), "mis_sym": arr(st.mis_sym, m, np.uint8),
               "mis_rev_off": arr(st.mis_rev_off, m, ot), "last_pos": int(st.last_pos)}
        lib.pgrc_match_free_export(C.byref(st))
        return out

    def export_pg_order(self, order, list_off, list_org_idx, list_rev_comp=None, read_org_idx=None,
                        rev_compl_pair_file: bool = False, byte_per_read_length: bool = True) -> dict:
        """exportMatchesInPgOrder's streams (see pgrc_match_export_pg_order).  order=None: the library makes the order on the
        device (ascending position, reads at one position by ascending index)."""
        keep = [np.ascontiguousarray(order if order is not None else [], dtype=np.uint32), np.ascontiguousarray(list_off, dtype=np.uint8),
                np.ascontiguousarray(list_org_idx, dtype=np.uint32)]
        a = _lib.ExportPgOrderArgs()
        if order is not None:
            a.order, a.n_matched = (keep[0].ctypes.data if keep[0].size else None), keep[0].size
        else:
            a.order_on_device = 1
        a.list_off, a.list_org_idx, a.list_count = keep[1].ctypes.data, keep[2].ctypes.data, keep[1].size
        if list_rev_comp is not None:
            keep.append(np.ascontiguousarray(list_rev_comp, dtype=np.uint8))
            a.list_rev_comp = keep[-1].ctypes.data
        if read_org_idx is not None:
            keep.append(np.ascontiguousarray(read_org_idx, dtype=np.uint32))
            a.read_org_idx = keep[-1].ctypes.data
        a.rev_compl_pair_file, a.byte_per_read_length = int(rev_compl_pair_file), int(byte_per_read_length)
        st = _lib.ExportStreams()
        self._ck(lib.pgrc_match_export_pg_order(self._h, C.byref(a), C.byref(st)))
        return self._streams(st)

    def export_entries(self, entry_read, entry_org_idx, rev_compl_pair_file: bool = False,
                       byte_per_read_length: bool = True) -> dict:
        """exportMatchesInOriginalOrder's streams for a caller-made entry list (see pgrc_match_export_entries)."""
        er = np.ascontiguousarray(entry_read, dtype=np.uint32)
        eo = np.ascontiguousarray(entry_org_idx, dtype=np.uint32)
        st = _lib.ExportStreams()
        self._ck(lib.pgrc_match_export_entries(self._h, er.ctypes.data_as(C.c_void_p), eo.ctypes.data_as(C.c_void_p), er.size,
                                               int(rev_compl_pair_file), int(byte_per_read_length), C.byref(st)))
        return self._streams(st)

    def export_original_order(self, read_org_idx, reads_total_count: int, pair_file_mode: bool = False,
                              rev_compl_pair_file: bool = False, byte_per_read_length: bool = True) -> dict:
        """exportMatchesInOriginalOrder's streams, entry list made on the device (see pgrc_match_export_original_order)."""
        ro = np.ascontiguousarray(read_org_idx, dtype=np.uint32)
        a = _lib.ExportOriginalOrderArgs()
        a.read_org_idx, a.reads_total_count = ro.ctypes.data, int(reads_total_count)
        a.pair_file_mode, a.rev_compl_pair_file = int(pair_file_mode), int(rev_compl_pair_file)
        a.byte_per_read_length = int(byte_per_read_length)
        st = _lib.ExportStreams()
        self._ck(lib.pgrc_match_export_original_order(self._h, C.byref(a), C.byref(st)))
        return self._streams(st)

    # ---- introspection
    def export_index(self, strand: int = 0):
        cnt = C.c_uint64(0)
        self._ck(lib.pgrc_match_export_index(self._h, strand, None, None, C.byref(cnt)))
        hs = copmem_params(self.seed_len, self.pg_len)["hash_size"]
        cumm = np.empty(hs + 2, dtype=np.uint32)
        positions = np.empty(int(cnt.value), dtype=np.uint32)
        self._ck(lib.pgrc_match_export_index(self._h, strand, cumm.ctypes.data_as(C.c_void_p),
                                             positions.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return cumm, positions

    def export_pg(self, strand: int = 0) -> np.ndarray:
        w = np.empty((self.pg_len + 15) // 16, dtype=np.uint32)
        self._ck(lib.pgrc_match_export_pg(self._h, strand, w.ctypes.data_as(C.c_void_p)))
        return w

    def reload_options(self):
        """The library reads its PGRC_* run-time options from the environment ONCE, when a context is created; tests and A/B
        tools that change one for a live context ask it to read them again (pgrc_match_reload_options)."""
        self._ck(lib.pgrc_match_reload_options(self._h))

    def set_profiling(self, on: bool = True):
        self._ck(lib.pgrc_match_set_profiling(self._h, 1 if on else 0))

    def redo_flags(self) -> np.ndarray:
        """per read: did the dual kernel of the last run do it again in the reference's order (pgrc_match_get_redo_flags)"""
        f = np.zeros(self.n, dtype=np.uint8)
        self._ck(lib.pgrc_match_get_redo_flags(self._h, f.ctypes.data_as(C.c_void_p)))
        return f

    def counters(self) -> dict:
        c = Counters()
        self._ck(lib.pgrc_match_get_counters_sized(self._h, C.byref(c), C.sizeof(c)))
        return {"searched": list(c.searched), "candidates": list(c.candidates), "probes": list(c.probes),
                "entry_fetches": list(c.entry_fetches), "verifies": list(c.verifies), "index_entries": list(c.index_entries), "ms_index": list(c.ms_index), "ms_match": list(c.ms_match),
                "ms_other": c.ms_other, "ms_total": c.ms_total, "ms_allgather": c.ms_allgather,
                "screened": c.screened, "ms_screen": c.ms_screen, "redo_reads": c.redo_reads,
                "schedule_downgraded": c.schedule_downgraded, "dual_seed_probes": c.dual_seed_probes,
                "dual": dict(zip(("searched", "candidates", "probes", "entry_fetches", "verifies"), list(c.dual)))}


class DefaultReadsMatcher:
    """matching/ReadsMatchers.h:24-83.  Subclasses fix the matcher kind ('mode')."""

    MODE = None
    NOT_MATCHED_POSITION = _lib.NOT_MATCHED_POS
    DISABLED_PREFIX_MODE = DISABLED_PREFIX_MODE

    def __init__(self, pg, revComplPg: bool, readsSet, matchPrefixLength: int = DISABLED_PREFIX_MODE,
                 readsExactMatchingChars: Optional[int] = None, maxMismatches: int = 0, minMismatches: int = 0,
                 device: int = -1):
        if matchPrefixLength != DISABLED_PREFIX_MODE:
            # the hot path is only ever entered with DISABLED_PREFIX_MODE (pgrc-encoder.cpp:362)
            raise PgrcMatchError(1, "prefix matching mode is not part of the accelerated path")
        self.reads = _as_ascii_2d(readsSet)
        self.readsCount, self.readLength = int(self.reads.shape[0]), int(self.reads.shape[1])
        self.matchingLength = self.readLength
        self.pg = _as_ascii_1d(pg)
        self.pgLength = int(self.pg.size)
        self.revComplPg = bool(revComplPg)
        seed = self.readLength if readsExactMatchingChars is None else int(readsExactMatchingChars)
        self.maxMismatches, self.minMismatches = int(maxMismatches), int(minMismatches)
        self.ctx = MatchContext(self.readLength, seed, self.maxMismatches, self.minMismatches, self.MODE, device)
        self.ctx.set_pg_ascii(self.pg)
        self.ctx.set_reads_ascii(self.reads)
        self.readMatchPos = self.readMatchRC = self.readMismatchesCount = None
        self.matchedReadsCount = 0
        self.matchedCountPerMismatches = np.zeros(256, dtype=np.uint64)

    def _fetch(self):
        (self.readMatchPos, rc, self.readMismatchesCount, self.matchedCountPerMismatches,
         self.matchedReadsCount) = self.ctx.get_results()
        self.readMatchRC = rc.astype(bool)

    def matchConstantLengthReads(self):
        """initMatching(); executeMatching(false); [RC(pg); executeMatching(true); RC(pg)]"""
        self.ctx.init_results()
        self.ctx.run(self.revComplPg)
        self._fetch()

    def getMatchedReadsBitmap(self, maxMismatches: int = _lib.NOT_MATCHED_CNT - 1) -> np.ndarray:
        return self.readMatchPos != np.uint64(_lib.NOT_MATCHED_POS)  # ReadsMatchers.cpp:677-683


class DefaultReadsExactMatcher(DefaultReadsMatcher):
    """ReadsMatchers.h:85-107 (mode 'e': whole-read seed, first hit in scan order wins)."""
    MODE = "e"

    def __init__(self, pg, revComplPg, readsSet, matchPrefixLength=DISABLED_PREFIX_MODE, device=-1):
        super().__init__(pg, revComplPg, readsSet, matchPrefixLength, None, 0, 0, device)


class AbstractReadsApproxMatcher(DefaultReadsMatcher):
    """ReadsMatchers.h:109-143."""

    def __init__(self, pg, revComplPg, readsSet, matchPrefixLength, readsExactMatchingChars, maxMismatches,
                 minMismatches=0, device=-1):
        super().__init__(pg, revComplPg, readsSet, matchPrefixLength, readsExactMatchingChars, maxMismatches,
                         minMismatches, device)
        self.targetMismatches = self.readLength // int(readsExactMatchingChars) - 1

    def getMatchedReadsBitmap(self, maxMismatches: int = _lib.NOT_MATCHED_CNT - 1) -> np.ndarray:
        return self.readMismatchesCount <= maxMismatches  # ReadsMatchers.cpp:685-691

    def continueMatchingConstantLengthReads(self, pMatcher: DefaultReadsMatcher):
        """Second phase (ReadsMatchers.cpp:174-184): take over pMatcher's results, then both passes."""
        mism = pMatcher.readMismatchesCount
        self.ctx.set_results(pMatcher.readMatchPos, pMatcher.readMatchRC.astype(np.uint8), mism)
        self.ctx.run(self.revComplPg)
        self._fetch()

    def getMismatches(self, reversed_flags=None):
        """updateEntry for every matched read (ReadsMatchers.cpp:548-559): (cum, codes, offsets)."""
        return self.ctx.extract_mismatches(reversed_flags)


class DefaultReadsApproxMatcher(AbstractReadsApproxMatcher):
    MODE = "d"


class InterleavedReadsApproxMatcher(AbstractReadsApproxMatcher):
    MODE = "i"


class CopMEMReadsApproxMatcher(AbstractReadsApproxMatcher):
    MODE = "c"


def mapReadsIntoPg(pg, revComplPg: bool, readsSet, readsExactMatchingChars: int, minCharsPerMismatch: int,
                   matchingMode: str = "c", preReadsExactMatchingChars: int = 0, preMatchingMode: str = "c",
                   matchPrefixLength: int = DISABLED_PREFIX_MODE, device: int = -1):
    """PgTools::mapReadsIntoPg (ReadsMatchers.cpp:693-796) up to the export step.

    Returns (matched bitmap, matcher).  The export into reads-list streams (:785-792) is the
    reference's own code consuming the matcher's result fields.
    """
    reads = _as_ascii_2d(readsSet)
    readLength = int(reads.shape[1])
    prm = MatchParams()

    def derive(seed, mode):
        rc = lib.pgrc_match_derive_params(readLength, seed, minCharsPerMismatch, mode.encode(), C.byref(prm))
        if rc == 7:
            raise PgrcMatchError(rc, f"Unknown matching mode: {mode}.")  # ReadsMatchers.cpp:738
        if rc:
            raise PgrcMatchError(rc, "bad matching parameters")
        return prm.seed_len, prm.max_mismatches, prm.min_mismatches, prm.mode.decode()

    def build(seed, kmax, kmin, kind):
        if kind == "e":
            return DefaultReadsExactMatcher(pg, revComplPg, reads, matchPrefixLength, device)
        cls = {"c": CopMEMReadsApproxMatcher, "d": DefaultReadsApproxMatcher, "i": InterleavedReadsApproxMatcher}[kind]
        return cls(pg, revComplPg, reads, matchPrefixLength, seed, kmax, kmin, device)

    if preReadsExactMatchingChars > 0:
        seed, kmax, kmin, kind = derive(preReadsExactMatchingChars, preMatchingMode)
    else:
        seed, kmax, kmin, kind = derive(readsExactMatchingChars, matchingMode)
    matcher = build(seed, kmax, kmin, kind)
    matcher.matchConstantLengthReads()
    if preReadsExactMatchingChars > 0:
        # 2nd phase, ReadsMatchers.cpp:749-779: minMismatches = shortcut ? max : targetMismatches+1, where
        # targetMismatches is still the FIRST phase's value (:713; it is only recomputed at :770, after its use)
        seed2 = min(readsExactMatchingChars, readLength)
        shortcut = matchingMode.upper() == matchingMode
        target = readLength // seed - 1
        kmin2 = kmax if shortcut else target + 1
        kind2 = matchingMode.lower()
        if kind2 not in "cdi":
            raise PgrcMatchError(7, f"Unknown mismatches mode: {matchingMode}.")
        if isinstance(matcher, DefaultReadsExactMatcher):
            matcher.readMismatchesCount = np.where(matcher.readMatchPos == np.uint64(_lib.NOT_MATCHED_POS), 255, 0).astype(np.uint8)
        approx = build(seed2, kmax, kmin2, kind2)
        approx.continueMatchingConstantLengthReads(matcher)
        matcher = approx
    return matcher.getMatchedReadsBitmap(), matcher
