"""Host-side mirror of the reference's TextMatcher seam (matching/TextMatchers.h:53-61) over include/pgrc_mem.h:
Pg-vs-Pg exact matching (SimplePgMatcher's CopMEMMatcher) on the MI355X.  No compute here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PgrcMatchError, lib

UINT32_MAX = 0xFFFFFFFF


def _ascii(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, str)):
        a = np.frombuffer(a.encode() if isinstance(a, str) else bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class CopMEMMatcher:
    """CopMEMMatcher(srcText, srcLength, targetMatchLength, minMatchLength) as a TextMatcher
    (matching/copmem/CopMEMMatcher.cpp:571-591, :604-622)."""

    def __init__(self, srcText, targetMatchLength: int, minMatchLength: int = UINT32_MAX, device: int = -1):
        self._h = C.c_void_p()
        rc = lib.pgrc_mem_create(int(targetMatchLength), int(minMatchLength), int(device), C.byref(self._h))
        if rc:
            raise PgrcMatchError(rc, (lib.pgrc_mem_last_error(None) or b"").decode())
        self.targetMatchLength = int(targetMatchLength)
        self._src = _ascii(srcText)          # the library borrows the text: keep it alive
        self._ck(lib.pgrc_mem_set_src_ascii(self._h, self._src.ctypes.data_as(C.c_void_p), self._src.size))

    def _ck(self, rc: int) -> None:
        if rc:
            raise PgrcMatchError(rc, (lib.pgrc_mem_last_error(self._h) or b"").decode())

    def matchTexts(self, destText, destIsSrc: bool, revComplMatching: bool, minMatchLength: int | None = None) -> np.ndarray:
        """-> uint64 array [count, 3] of (posSrcText, length, posDestText), discovery order.  destText is the text as
        SimplePgMatcher hands it over (already reverse-complemented when revComplMatching)."""
        d = _ascii(destText)
        out = C.POINTER(_lib.TextMatch)()
        cnt = C.c_uint64(0)
        self._ck(lib.pgrc_mem_match_texts(self._h, d.ctypes.data_as(C.c_void_p), d.size, int(bool(destIsSrc)),
                                          int(bool(revComplMatching)),
                                          self.targetMatchLength if minMatchLength is None else int(minMatchLength),
                                          C.byref(out), C.byref(cnt)))
        n = cnt.value
        res = np.zeros((n, 3), dtype=np.uint64)
        if n:
            res[:] = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n * 3,)).reshape(n, 3)
            lib.pgrc_mem_free_matches(out)
        return res

    def counters(self) -> dict:
        c = _lib.MemCounters()
        self._ck(lib.pgrc_mem_get_counters(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in c._fields_}

    def close(self) -> None:
        if self._h:
            lib.pgrc_mem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
