"""ctypes bindings of the TEST-ONLY checkers: oracle/libpgrc_oracle.so (our CPU restatement) and,
when present, oracle/_ref/libpgrc_ref.so (the real reference compiled by oracle/Makefile).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libpgrc_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpgrc_ref.so")

NOT_MATCHED_POS = 0xFFFFFFFFFFFFFFFF
_P = C.c_void_p


class CopmemParams(C.Structure):
    _fields_ = [("L", C.c_int32), ("K", C.c_int32), ("k1", C.c_int32), ("k2", C.c_int32), ("hash_size", C.c_uint32)]


class Index(C.Structure):
    _fields_ = [("p", CopmemParams), ("pg_len", C.c_uint64), ("cumm", C.POINTER(C.c_uint32)),
                ("positions", C.POINTER(C.c_uint32)), ("count", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("pos", _P), ("rc", _P), ("mism", _P), ("hist", C.c_uint64 * 256), ("matched", C.c_uint64),
                ("searched", C.c_uint64 * 2), ("candidates", C.c_uint64 * 2), ("falses", C.c_uint64 * 2)]


class MapParams(C.Structure):
    _fields_ = [("kmax", C.c_uint8), ("kmin", C.c_uint8), ("seed_len", C.c_uint32), ("parts", C.c_uint8),
                ("matcher", C.c_char)]


_or = None


def oracle():
    global _or
    if _or is None:
        lib = C.CDLL(ORACLE_SO)
        lib.pgrc_or_copmem_derive.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(CopmemParams)]
        lib.pgrc_or_copmem_hash.argtypes = [C.c_int, C.c_char_p]
        lib.pgrc_or_copmem_hash.restype = C.c_uint32
        lib.pgrc_or_index_build.argtypes = [_P, C.c_uint64, C.c_uint32, C.POINTER(Index)]
        lib.pgrc_or_index_free.argtypes = [C.POINTER(Index)]
        lib.pgrc_or_copmem_match_read.argtypes = [C.POINTER(Index), _P, _P, C.c_uint32, C.c_uint8, C.c_uint8,
                                                  C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.pgrc_or_copmem_match_read.restype = C.c_uint64
        lib.pgrc_or_set_early_stop.argtypes = [C.c_int]
        lib.pgrc_or_set_dual_spec.argtypes = [C.c_int]
        lib.pgrc_or_match_copmem_dual.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8,
                                                  C.c_uint8, C.c_int, C.c_int, C.POINTER(Result), C.POINTER(C.c_uint64)]
        lib.pgrc_or_match_copmem_screened.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8,
                                                      C.c_uint8, C.c_int, C.c_int, C.POINTER(Result)]
        lib.pgrc_or_probe_count.argtypes = [C.c_int]
        lib.pgrc_or_probe_count.restype = C.c_uint64
        lib.pgrc_or_match_copmem.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8,
                                             C.c_uint8, C.c_int, C.c_int, C.c_int, C.POINTER(Result)]
        lib.pgrc_or_match_seedindex.argtypes = [C.c_char, _P, C.c_uint64, _P, C.c_uint64, C.c_uint32, C.c_uint32,
                                                C.c_uint8, C.c_uint8, C.c_int, C.POINTER(Result)]
        lib.pgrc_or_map_derive.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_char, C.POINTER(MapParams)]
        lib.pgrc_or_extract_mismatches.argtypes = [_P, C.c_uint64, _P, C.c_uint32, C.c_int, C.c_int, C.c_uint8, _P, _P]
        lib.pgrc_or_revcomp.argtypes = [_P, C.c_uint64]
        lib.pgrc_or_pack_read.argtypes = [_P, C.c_uint32, C.c_char_p, _P]
        lib.pgrc_or_unpack_read.argtypes = [_P, C.c_uint32, C.c_char_p, _P]
        _or = lib
    return _or


_ref = None


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        # libpgrc_ref.so links the product library (for the adapter test): load that first so that both
        # resolve to ONE HIP runtime (see pgrc_amd/_lib.py)
        import pgrc_amd  # noqa: F401
        lib = C.CDLL(REF_SO)
        if hasattr(lib, "pgrc_ref_match_via_adapter"):
            lib.pgrc_ref_match_via_adapter.argtypes = [C.c_char, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint32,
                                                       C.c_uint32, C.c_uint8, C.c_uint8, C.c_int, C.c_int, _P, _P, _P, _P,
                                                       C.POINTER(C.c_uint64)]
        lib.pgrc_ref_match.argtypes = [C.c_char, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_uint8, C.c_uint8, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P,
                                       C.POINTER(C.c_uint64), _P]
        lib.pgrc_ref_copmem_index.argtypes = [_P, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                              C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)),
                                              C.POINTER(C.c_uint64)]
        lib.pgrc_ref_copmem_hash.argtypes = [C.c_uint32, C.c_char_p, C.POINTER(C.c_int32)]
        lib.pgrc_ref_copmem_hash.restype = C.c_uint32
        lib.pgrc_ref_copmem_match_read.argtypes = [_P, C.c_uint64, C.c_uint32, _P, C.c_uint32, C.c_uint8, C.c_uint8,
                                                   C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
        lib.pgrc_ref_copmem_match_read.restype = C.c_uint64
        lib.pgrc_ref_extract.argtypes = [_P, C.c_uint64, _P, C.c_uint32, C.c_uint64, C.c_int, C.c_uint8, C.c_uint32,
                                         C.c_int, _P, _P]
        lib.pgrc_ref_pack_read.argtypes = [_P, C.c_uint32, C.c_char_p, _P]
        lib.pgrc_ref_revcomp.argtypes = [_P, C.c_uint64]
        lib.pgrc_ref_free.argtypes = [_P]
        if hasattr(lib, "pgrc_ref_mem_match"):
            lib.pgrc_ref_mem_match.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        _ref = lib
    return _ref


def _ptr(a):
    return a.ctypes.data_as(_P)


def _new_result(n):
    pos = np.empty(n, dtype=np.uint64)
    rc = np.empty(n, dtype=np.uint8)
    mism = np.empty(n, dtype=np.uint8)
    r = Result()
    r.pos, r.rc, r.mism = _ptr(pos), _ptr(rc), _ptr(mism)
    return r, pos, rc, mism


def _pack_result(r, pos, rc, mism):
    return {"pos": pos, "rc": rc, "mism": mism, "hist": np.array(r.hist[:], dtype=np.uint64), "matched": int(r.matched),
            "searched": list(r.searched), "candidates": list(r.candidates), "falses": list(r.falses)}


def oracle_match(mode, pg, reads, seed_len, kmax, kmin, rev_compl=True, threads=8, state=None, early_stop=False):
    """mode in 'c','d','i','e'. pg: uint8[G] ASCII, reads: uint8[n, L] ASCII.
    early_stop (mode c): the HIP kernel's early-stop rule instead of the reference's full loops -- same results, the
    work counters (candidates) are then the kernel's."""
    if early_stop:
        oracle().pgrc_or_set_early_stop(1)
        try:
            return oracle_match(mode, pg, reads, seed_len, kmax, kmin, rev_compl, threads, state)
        finally:
            oracle().pgrc_or_set_early_stop(0)
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    r, pos, rc, mism = _new_result(n)
    if mode == "c":
        init = 1
        if state is not None:
            pos[:], rc[:], mism[:] = state
            r.hist = (C.c_uint64 * 256)(*[int(x) for x in np.bincount(mism, minlength=256)])
            r.matched = int((mism != 255).sum())
            init = 0
        e = oracle().pgrc_or_match_copmem(_ptr(pg), pg.size, _ptr(reads), n, L, seed_len, kmax, kmin,
                                          1 if rev_compl else 0, threads, init, C.byref(r))
    else:
        e = oracle().pgrc_or_match_seedindex(mode.encode(), _ptr(pg), pg.size, _ptr(reads), n, L, seed_len, kmax, kmin,
                                             1 if rev_compl else 0, C.byref(r))
    if e:
        raise RuntimeError(f"oracle returned {e}")
    return _pack_result(r, pos, rc, mism)


def oracle_match_screened(pg, reads, seed_len, kmax, kmin, threads=8, state=None):
    """Mode c, both strands, in the HIP path's schedule (exact-match screen first): see pgrc_or_match_copmem_screened."""
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    r, pos, rc, mism = _new_result(n)
    init = 1
    if state is not None:
        pos[:], rc[:], mism[:] = state
        init = 0
    e = oracle().pgrc_or_match_copmem_screened(_ptr(pg), pg.size, _ptr(reads), n, L, seed_len, kmax, kmin, threads, init, C.byref(r))
    if e:
        raise RuntimeError(f"oracle returned {e}")
    return _pack_result(r, pos, rc, mism)


def oracle_match_dual(pg, reads, seed_len, kmax, threads=8, state=None):
    """Mode c, both strands, kmin 0, as ONE query over both strands per read (pgrc_or_match_copmem_dual).
    The returned dict also has 'aborted': reads that fell back to the reference's order."""
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    r, pos, rc, mism = _new_result(n)
    init = 1
    if state is not None:
        pos[:], rc[:], mism[:] = state
        init = 0
    ab = C.c_uint64(0)
    e = oracle().pgrc_or_match_copmem_dual(_ptr(pg), pg.size, _ptr(reads), n, L, seed_len, kmax, 0, threads, init, C.byref(r), C.byref(ab))
    if e:
        raise RuntimeError(f"oracle returned {e}")
    out = _pack_result(r, pos, rc, mism)
    out["aborted"] = int(ab.value)
    return out


def oracle_index(pg, seed_len):
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    idx = Index()
    e = oracle().pgrc_or_index_build(_ptr(pg), pg.size, seed_len, C.byref(idx))
    if e:
        raise RuntimeError(f"oracle index returned {e}")
    hs = idx.p.hash_size
    cumm = np.ctypeslib.as_array(idx.cumm, shape=(hs + 2,)).copy()
    positions = np.ctypeslib.as_array(idx.positions, shape=(max(int(idx.count), 1),))[: int(idx.count)].copy()
    prm = {"K": idx.p.K, "k1": idx.p.k1, "k2": idx.p.k2, "hash_size": hs}
    oracle().pgrc_or_index_free(C.byref(idx))
    return prm, cumm, positions


def oracle_extract(pg, pos, read, rc, reversed_, cnt):
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    codes = np.zeros(max(cnt, 1), dtype=np.uint8)
    offs = np.zeros(max(cnt, 1), dtype=np.uint16)
    oracle().pgrc_or_extract_mismatches(_ptr(pg), int(pos), _ptr(read), read.size, int(rc), int(reversed_), int(cnt),
                                        _ptr(codes), _ptr(offs))
    return codes[:cnt], offs[:cnt]


def ref_match(mode, pg, reads, seed_len, kmax, kmin, rev_compl=True, n_nset=0, index_threads=1, omp_threads=1):
    """The compiled reference.  index_threads=1 selects its serial (deterministic) copMEM index build.  omp_threads is
    the width of its per-read loop: it defaults to 1 because that loop stores readMatchRC -- a vector<bool> -- from
    several threads (ReadsMatchers.cpp:426-446), so reads whose flags share a 64-bit word across two threads' chunks
    occasionally lose an update (seen once: one RC flag of 10 000 reads; DESIGN.md, reference quirk 7)."""
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    pos = np.empty(n, dtype=np.uint64)
    rc = np.empty(n, dtype=np.uint8)
    mism = np.empty(n, dtype=np.uint8)
    hist = np.zeros(256, dtype=np.uint64)
    stats = np.zeros(2, dtype=np.uint64)
    matched = C.c_uint64(0)
    e = ref().pgrc_ref_match(mode.encode(), _ptr(pg), pg.size, _ptr(reads), n - n_nset, n_nset, L, seed_len, kmax, kmin,
                             1 if rev_compl else 0, index_threads, omp_threads, _ptr(pos), _ptr(rc), _ptr(mism),
                             _ptr(hist), C.byref(matched), _ptr(stats))
    if e:
        raise RuntimeError(f"ref returned {e}")
    return {"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": int(matched.value),
            "better": int(stats[0]), "falses": int(stats[1])}


def ref_index(pg, seed_len, index_threads=1):
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    K, k1, k2 = C.c_int32(), C.c_int32(), C.c_int32()
    hs = C.c_uint32()
    cumm = C.POINTER(C.c_uint32)()
    positions = C.POINTER(C.c_uint32)()
    cnt = C.c_uint64()
    e = ref().pgrc_ref_copmem_index(_ptr(pg), pg.size, seed_len, index_threads, C.byref(K), C.byref(k1), C.byref(k2),
                                    C.byref(hs), C.byref(cumm), C.byref(positions), C.byref(cnt))
    if e:
        raise RuntimeError(f"ref index returned {e}")
    c = np.ctypeslib.as_array(cumm, shape=(hs.value + 2,)).copy()
    p = np.ctypeslib.as_array(positions, shape=(max(int(cnt.value), 1),))[: int(cnt.value)].copy()
    ref().pgrc_ref_free(cumm)
    ref().pgrc_ref_free(positions)
    return {"K": K.value, "k1": k1.value, "k2": k2.value, "hash_size": hs.value}, c, p


def ref_extract(pg, pos, read, rc, cnt, org_idx=0, rev_compl_pair_file=False):
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    codes = np.zeros(256, dtype=np.uint8)
    offs = np.zeros(256, dtype=np.uint16)
    k = ref().pgrc_ref_extract(_ptr(pg), pg.size, _ptr(read), read.size, int(pos), int(rc), int(cnt), int(org_idx),
                               1 if rev_compl_pair_file else 0, _ptr(codes), _ptr(offs))
    return codes[:k], offs[:k]


def have_adapter() -> bool:
    return have_ref() and hasattr(ref(), "pgrc_ref_match_via_adapter")


def ref_match_via_adapter(mode, pg, reads, seed_len, kmax, kmin, rev_compl=True, n_nset=0, entry=1):
    """The reference's own flow with integration/HipReadsMatcher plugged into the matcher seam (needs a GPU)."""
    pg = np.ascontiguousarray(pg, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    pos = np.empty(n, dtype=np.uint64)
    rc = np.empty(n, dtype=np.uint8)
    mism = np.empty(n, dtype=np.uint8)
    hist = np.zeros(256, dtype=np.uint64)
    matched = C.c_uint64(0)
    e = ref().pgrc_ref_match_via_adapter(mode.encode(), _ptr(pg), pg.size, _ptr(reads), n - n_nset, n_nset, L, seed_len,
                                         kmax, kmin, 1 if rev_compl else 0, entry, _ptr(pos), _ptr(rc), _ptr(mism),
                                         _ptr(hist), C.byref(matched))
    if e:
        raise RuntimeError(f"adapter returned {e}")
    return {"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": int(matched.value)}


# ---- Pg-vs-Pg exact matching

def revcomp_ascii(a):
    """reverse complement of an ASCII uint8 array (N stays N)"""
    lut = np.arange(256, dtype=np.uint8)
    for x, y in (b"AT", b"TA", b"CG", b"GC"):
        lut[x] = y
    return lut[np.ascontiguousarray(a, dtype=np.uint8)[::-1]].copy()


def mem_dest(src, dest, dest_is_src, rev_compl):
    """the text SimplePgMatcher::exactMatchPg hands to matchTexts (SimplePgMatcher.cpp:31-43)"""
    base = src if dest_is_src else dest
    return revcomp_ascii(base) if rev_compl else np.ascontiguousarray(base, dtype=np.uint8)


def oracle_mem_match(src, dest_text, dest_is_src, rev_compl, target_len=45, min_len=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    d = np.ascontiguousarray(dest_text, dtype=np.uint8)
    lib = oracle()
    lib.pgrc_or_mem_match.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    lib.pgrc_or_mem_free.argtypes = [C.POINTER(C.c_uint64)]
    out = C.POINTER(C.c_uint64)()
    cnt = C.c_uint64(0)
    e = lib.pgrc_or_mem_match(_ptr(src), src.size, _ptr(d), d.size, int(dest_is_src), int(rev_compl), target_len,
                              target_len if min_len is None else min_len, C.byref(out), C.byref(cnt))
    if e:
        raise RuntimeError(f"oracle mem_match returned {e}")
    res = np.ctypeslib.as_array(out, shape=(cnt.value * 3,)).reshape(-1, 3).copy() if cnt.value else np.zeros((0, 3), np.uint64)
    lib.pgrc_or_mem_free(out)
    return res


def ref_mem_match_via_adapter(src, dest_text, dest_is_src, rev_compl, target_len=45, min_len=None):
    """CopMEMMatcher::matchTexts through integration/HipTextMatcher compiled against the reference (needs a GPU)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    d = np.ascontiguousarray(dest_text, dtype=np.uint8)
    f = ref().pgrc_ref_mem_match_via_adapter
    f.argtypes = [_P, C.c_uint64, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                  C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    out = C.POINTER(C.c_uint64)()
    cnt = C.c_uint64(0)
    e = f(_ptr(src), src.size, _ptr(d), d.size, int(dest_is_src), int(rev_compl), target_len,
          target_len if min_len is None else min_len, C.byref(out), C.byref(cnt))
    if e:
        raise RuntimeError(f"adapter mem_match returned {e}")
    res = np.ctypeslib.as_array(out, shape=(cnt.value * 3,)).reshape(-1, 3).copy() if cnt.value else np.zeros((0, 3), np.uint64)
    ref().pgrc_ref_free(C.cast(out, _P))
    return res


def ref_mem_match(src, dest_text, dest_is_src, rev_compl, target_len=45, min_len=None, threads=1):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    d = np.ascontiguousarray(dest_text, dtype=np.uint8)
    out = C.POINTER(C.c_uint64)()
    cnt = C.c_uint64(0)
    e = ref().pgrc_ref_mem_match(_ptr(src), src.size, _ptr(d), d.size, int(dest_is_src), int(rev_compl), target_len,
                                 0xFFFFFFFF, target_len if min_len is None else min_len, threads, C.byref(out), C.byref(cnt))
    if e:
        raise RuntimeError(f"ref mem_match returned {e}")
    res = np.ctypeslib.as_array(out, shape=(cnt.value * 3,)).reshape(-1, 3).copy() if cnt.value else np.zeros((0, 3), np.uint64)
    ref().pgrc_ref_free(C.cast(out, _P))
    return res
