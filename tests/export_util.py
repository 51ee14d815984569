"""Helpers of the export tests (row f1): synthetic cases with a reads list already on the pseudogenome, the checkers'
bindings (oracle restatement; the compiled reference's export into stream files), stream comparison."""
import ctypes as C
import os

import numpy as np

import oracle as orc
from util import make_inputs

_P = C.c_void_p
STREAMS = ("off", "org_idx", "rev_comp", "mis_cnt", "mis_sym", "mis_rev_off")
FILES = {"off": "_rl_off.pg", "org_idx": "_rl_idx.pg", "rev_comp": "_rl_rc.pg", "mis_cnt": "_rl_mis_cnt.pg",
         "mis_sym": "_rl_mis_sym.pg", "mis_rev_off": "_rl_mis_roff.pg"}


class OrStreams(C.Structure):   # pgrc_or_export_streams
    _fields_ = [("n_entries", C.c_uint64), ("n_mismatches", C.c_uint64), ("off_width", C.c_uint32), ("off", _P),
                ("org_idx", _P), ("rev_comp", _P), ("mis_cnt", _P), ("mis_sym", _P), ("mis_rev_off", _P), ("last_pos", C.c_uint64)]


def export_case(seed, G=300_000, n=20_000, L=100, n_with_n=400, paired=False, dups=300, list_gap=60, empty_list=False,
                short_list=False):
    """pg, reads (the last n_with_n carry N: the N set), a reads list on the Pg (offset deltas < 256, original indexes,
    RC flags), original indexes of the reads (LQ part ascending, N part ascending: SumOfMappings), the total."""
    pg, reads = make_inputs(G, n, L, seed, n_with_n=n_with_n, paired=paired)
    rng = np.random.default_rng(seed)
    n_lq = n - n_with_n
    if dups:     # identical reads match at identical positions: ties in the position sort
        src = rng.integers(0, n_lq, size=dups)
        dst = rng.integers(0, n_lq, size=dups)
        reads[dst] = reads[src]
    if empty_list:
        h = 0
    else:
        h = (G - L) // list_gap
        if short_list:
            h //= 3                      # the list ends early: matches beyond its last entry (the -1 quirk)
    off = rng.integers(1, 2 * list_gap - 1, size=h).astype(np.uint8)
    if h:
        off[0] = 0                       # an entry at position 0
        while int(off.astype(np.int64).sum()) > G - L:
            off = off[: off.size - 100]
        h = off.size
    total = h + n
    perm = rng.permutation(total).astype(np.uint32)
    list_org = perm[:h].copy()
    read_org = np.concatenate([np.sort(perm[h:h + n_lq]), np.sort(perm[h + n_lq:])]).astype(np.uint32)
    list_rc = (rng.random(h) < 0.4).astype(np.uint8)
    return {"pg": pg, "reads": reads, "n_n": n_with_n, "L": L, "list_off": off, "list_org": list_org, "list_rc": list_rc,
            "read_org": read_org, "total": total}


def position_order(pos, threads=1):
    """matched reads by ascending position in the reference's tie order (the adapter's positionOrder, oracle/_ref)"""
    r = orc.ref()
    r.pgrc_ref_position_order.restype = C.c_uint64
    r.pgrc_ref_position_order.argtypes = [_P, C.c_uint64, C.c_int, _P]
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    out = np.empty(pos.size, dtype=np.uint32)
    k = r.pgrc_ref_position_order(pos.ctypes.data_as(_P), pos.size, threads, out.ctypes.data_as(_P))
    return out[:k].copy()


def stable_order(pos):
    idx = np.flatnonzero(pos != np.uint64(2**64 - 1))
    return idx[np.argsort(pos[idx], kind="stable")].astype(np.uint32)


def _or_streams(ne_max, nm_max):
    bufs = {"off": np.zeros(2 * ne_max + 2, np.uint8), "org_idx": np.zeros(ne_max + 1, np.uint32),
            "rev_comp": np.zeros(ne_max + 1, np.uint8), "mis_cnt": np.zeros(ne_max + 1, np.uint8),
            "mis_sym": np.zeros(nm_max + 1, np.uint8), "mis_rev_off": np.zeros(2 * nm_max + 2, np.uint8)}
    s = OrStreams()
    for k, v in bufs.items():
        setattr(s, k, v.ctypes.data)
    return s, bufs


def _or_result(s, bufs):
    ne, nm, w = int(s.n_entries), int(s.n_mismatches), int(s.off_width)
    ot = np.uint8 if w == 1 else np.uint16
    return {"off": bufs["off"][: ne * w].view(ot).copy(), "org_idx": bufs["org_idx"][:ne].copy(),
            "rev_comp": bufs["rev_comp"][:ne].copy(), "mis_cnt": bufs["mis_cnt"][:ne].copy(),
            "mis_sym": bufs["mis_sym"][:nm].copy(), "mis_rev_off": bufs["mis_rev_off"][: nm * w].view(ot).copy(),
            "last_pos": int(s.last_pos)}


def oracle_export_pg_order(case, res, order, pair_file=False, byte_mode=True, with_read_org=True):
    lib = orc.oracle()
    f = lib.pgrc_or_export_pg_order
    f.argtypes = [_P, _P, C.c_uint32, _P, _P, _P, _P, C.c_uint64, _P, _P, _P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(OrStreams)]
    reads = np.ascontiguousarray(case["reads"])
    order = np.ascontiguousarray(order, dtype=np.uint32)
    mism = np.ascontiguousarray(res["mism"])
    nm = int(mism[mism != 255].astype(np.int64).sum())
    s, bufs = _or_streams(order.size + case["list_off"].size, nm)
    ro = np.ascontiguousarray(case["read_org"]) if with_read_org else None
    f(case["pg"].ctypes.data_as(_P), reads.ctypes.data_as(_P), case["L"], np.ascontiguousarray(res["pos"]).ctypes.data_as(_P),
      np.ascontiguousarray(res["rc"]).ctypes.data_as(_P), mism.ctypes.data_as(_P), order.ctypes.data_as(_P), order.size,
      ro.ctypes.data_as(_P) if ro is not None else None, case["list_off"].ctypes.data_as(_P), case["list_org"].ctypes.data_as(_P),
      case["list_rc"].ctypes.data_as(_P) if case["list_rc"] is not None else None, case["list_off"].size, int(pair_file),
      int(byte_mode), C.byref(s))
    return _or_result(s, bufs)


def oracle_export_entries(case, res, entry_read, entry_org, pair_file=False, byte_mode=True):
    lib = orc.oracle()
    f = lib.pgrc_or_export_entries
    f.argtypes = [_P, _P, C.c_uint32, _P, _P, _P, _P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(OrStreams)]
    reads = np.ascontiguousarray(case["reads"])
    er = np.ascontiguousarray(entry_read, dtype=np.uint32)
    eo = np.ascontiguousarray(entry_org, dtype=np.uint32)
    mism = np.ascontiguousarray(res["mism"])
    nm = int(mism[mism != 255].astype(np.int64).sum())
    s, bufs = _or_streams(er.size, nm)
    f(case["pg"].ctypes.data_as(_P), reads.ctypes.data_as(_P), case["L"], np.ascontiguousarray(res["pos"]).ctypes.data_as(_P),
      np.ascontiguousarray(res["rc"]).ctypes.data_as(_P), mism.ctypes.data_as(_P), er.ctypes.data_as(_P), eo.ctypes.data_as(_P),
      er.size, int(pair_file), int(byte_mode), C.byref(s))
    return _or_result(s, bufs)


def ref_export_run(case, prefix, use_adapter, seed_len=38, kmax=33, preserve_order=False, pair_file_mode=False,
                   rev_compl_pair_file=False, threads=1, with_read_org=True):
    """matching + export inside the compiled reference (CPU matcher + its own export, or HipReadsMatcher + its device
    export); returns the stream files' bytes and the bytes written to the archive stream"""
    r = orc.ref()
    f = r.pgrc_ref_export_run
    f.argtypes = [C.c_int, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, _P, _P, _P,
                  C.c_uint64, _P, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    reads = np.ascontiguousarray(case["reads"])
    n = reads.shape[0]
    ro = np.ascontiguousarray(case["read_org"]) if with_read_org else None
    arch = prefix + ".archive"
    e = f(int(use_adapter), case["pg"].ctypes.data_as(_P), case["pg"].size, reads.ctypes.data_as(_P), n - case["n_n"], case["n_n"],
          case["L"], seed_len, kmax, 0, case["list_off"].ctypes.data_as(_P), case["list_org"].ctypes.data_as(_P),
          case["list_rc"].ctypes.data_as(_P) if case["list_rc"] is not None else None, case["list_off"].size,
          ro.ctypes.data_as(_P) if ro is not None else None, case["total"] if with_read_org else n, int(preserve_order),
          int(pair_file_mode), int(rev_compl_pair_file), threads, prefix.encode(), arch.encode())
    assert e == 0
    out = {k: open(prefix + suf, "rb").read() for k, suf in FILES.items()}
    out["archive"] = open(arch, "rb").read()
    return out


def stream_bytes(st):
    return {k: np.ascontiguousarray(st[k]).tobytes() for k in STREAMS}


def original_order_entries(read_org, matched, total, pair_file_mode, n_lq):
    """the entry list of exportMatchesInOriginalOrder (ReadsMatchers.cpp:616-667), restated with numpy: per parity
    class, all original indexes in ascending order; an index that belongs to the read set yields an entry only if the
    read matched, every other index a filler"""
    owner = np.full(total, -1, dtype=np.int64)
    owner[read_org] = np.arange(read_org.size)
    er, eo = [], []
    parts = 2 if pair_file_mode else 1
    for p in range(parts):
        for o in range(p, total, parts):
            i = owner[o]
            if i < 0:
                er.append(0xFFFFFFFF)
                eo.append(o)
            elif matched[i]:
                er.append(i)
                eo.append(o)
    return np.array(er, dtype=np.uint32), np.array(eo, dtype=np.uint32)


# ---- committed golden fixtures of the export (tests/golden/make_golden_export.py)

def load_export_golden(name):
    """-> (case, paired-file rule, reference results, reference order, {pg|org: stream bytes})"""
    import hashlib
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    m = json.load(open(os.path.join(gold, "manifest_export.json")))[name]
    case = export_case(**m["kw"])
    h = hashlib.sha256()
    for k in ("pg", "reads", "list_off", "list_org", "list_rc", "read_org"):
        h.update(np.ascontiguousarray(case[k]).tobytes())
    assert h.hexdigest() == m["inputs_sha256"], "generator drifted"
    z = np.load(os.path.join(gold, name + ".npz"))
    res = {"pos": z["pos"], "rc": z["rc"], "mism": z["mism"]}
    streams = {tag: {k: z[f"{tag}_{k}"].tobytes() for k in STREAMS} for tag in ("pg", "org")}
    return case, m["paired_file_rule"], m["kmax"], res, z["order"], streams


EXPORT_GOLDEN = ("export_se", "export_pe_pairfile", "export_short_list", "export_L250")
