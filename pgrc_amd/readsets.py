"""Host-side mirror of the reference's read-set division (readsset/DividedPCLReadsSets.h) on top of
include/pgrc_reads.h: computes nothing itself."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import PgrcMatchError, lib


class DividedPCLReadsSets:
    """DividedPCLReadsSets::getQualityDivisionBasedReadsSets (DividedPCLReadsSets.cpp:59-100) over batches of FASTQ
    records given as row arrays.  `divide(reads, quals)` returns the packed rows of the HQ / LQ / N sets of the batch
    (PackedConstantLengthReadsSet::packedReads layout) and the batch-local indexes of the LQ / N reads."""

    def __init__(self, readLength: int, error_limit: float = 1.0, simplified_suffix_mode: bool = True,
                 separateNReadsSet: bool = False, nReadsLQ: bool = False, device: int = -1):
        prm = _lib.DivideParams(int(readLength), float(error_limit), int(bool(simplified_suffix_mode)),
                                int(bool(separateNReadsSet)), int(bool(nReadsLQ)), int(device))
        self._h = C.c_void_p()
        code = lib.pgrc_divider_create(C.byref(prm), C.byref(self._h))
        if code:
            raise PgrcMatchError(code, (lib.pgrc_divider_last_error(None) or b"").decode())
        self.readLength = int(readLength)
        self.needs_quality = error_limit < 1

    def _ck(self, code: int) -> None:
        if code:
            raise PgrcMatchError(code, (lib.pgrc_divider_last_error(self._h) or b"").decode())

    def divide(self, reads: np.ndarray, quals: Optional[np.ndarray] = None) -> dict:
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        n = reads.shape[0] if reads.ndim == 2 else reads.size // self.readLength
        assert reads.size == n * self.readLength
        qp = None
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
            assert quals.size == reads.size
            qp = quals.ctypes.data_as(C.c_void_p)
        out = _lib.DividedReads()
        self._ck(lib.pgrc_divider_run(self._h, reads.ctypes.data_as(C.c_void_p), qp, n, C.byref(out)))

        def arr(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype=dtype)
            a = np.empty(count, dtype=dtype)                 # (the library's arrays live until the next run: copy out)
            C.memmove(a.ctypes.data, ptr, a.nbytes)
            return a
        res = {"n_hq": int(out.n_hq), "n_lq": int(out.n_lq), "n_n": int(out.n_n),
               "symbols": (int(out.hq_symbols), int(out.lq_symbols), int(out.n_symbols)),
               "row_bytes": (int(out.hq_row_bytes), int(out.lq_row_bytes), int(out.n_row_bytes)),
               "hq_rows": arr(out.hq_rows, out.n_hq * out.hq_row_bytes, np.uint8),
               "lq_rows": arr(out.lq_rows, out.n_lq * out.lq_row_bytes, np.uint8),
               "n_rows": arr(out.n_rows, out.n_n * out.n_row_bytes, np.uint8),
               "lq_index": arr(out.lq_index, out.n_lq, np.uint32), "n_index": arr(out.n_index, out.n_n, np.uint32)}
        return res

    def _result(self, out) -> dict:
        def arr(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype=dtype)
            a = np.empty(count, dtype=dtype)
            C.memmove(a.ctypes.data, ptr, a.nbytes)
            return a
        return {"n_hq": int(out.n_hq), "n_lq": int(out.n_lq), "n_n": int(out.n_n),
                "symbols": (int(out.hq_symbols), int(out.lq_symbols), int(out.n_symbols)),
                "row_bytes": (int(out.hq_row_bytes), int(out.lq_row_bytes), int(out.n_row_bytes)),
                "hq_rows": arr(out.hq_rows, out.n_hq * out.hq_row_bytes, np.uint8),
                "lq_rows": arr(out.lq_rows, out.n_lq * out.lq_row_bytes, np.uint8),
                "n_rows": arr(out.n_rows, out.n_n * out.n_row_bytes, np.uint8),
                "lq_index": arr(out.lq_index, out.n_lq, np.uint32), "n_index": arr(out.n_index, out.n_n, np.uint32)}

    def divide_fastq(self, text: bytes, pair_text: Optional[bytes] = None, rev_compl_pair: bool = False, final=True):
        """One piece of FASTQ text (and of the pair file's): (result dict, records taken, bytes consumed, pair bytes consumed).
        final: True / 1 = nothing follows in either text; 2 / 4 = only the first / the second text ends here."""
        out = _lib.DividedReads()
        used, pused, nrec = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        t = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, np.uint8)
        pt = None
        if pair_text is not None:
            pt = np.frombuffer(pair_text, dtype=np.uint8) if len(pair_text) else np.zeros(1, np.uint8)
        self._ck(lib.pgrc_divider_run_fastq(self._h, t.ctypes.data_as(C.c_void_p), len(text),
                                            pt.ctypes.data_as(C.c_void_p) if pt is not None else None,
                                            len(pair_text) if pair_text is not None else 0, int(bool(rev_compl_pair)), int(final),
                                            C.byref(used), C.byref(pused), C.byref(nrec), C.byref(out)))
        return self._result(out), int(nrec.value), int(used.value), int(pused.value)

    def last_was_terminal(self) -> bool:
        """did the last divide_fastq take the last records the reference's iteration would take (pgrc_divider_last_was_terminal)"""
        return bool(lib.pgrc_divider_last_was_terminal(self._h))

    def last_ms(self):
        ms = (C.c_float * 3)()
        self._ck(lib.pgrc_divider_last_ms(self._h, C.byref(ms)))
        return {"upload": ms[0], "kernels": ms[1], "download": ms[2]}

    def close(self) -> None:
        if self._h:
            lib.pgrc_divider_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
