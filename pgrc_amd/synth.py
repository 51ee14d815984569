"""Synthetic pseudogenome / read sets (include/pgrc_synth.h) -- host loops and HIP generators."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import SynthPg, SynthReads, lib


def pg_params(pg_len: int, seed: int = 12345, grid: int = 20000, plant_len: int = 3000, pool_div: int = 8,
              tandem_every: int = 64) -> SynthPg:
    return SynthPg(seed, pg_len, grid, plant_len, pool_div, tandem_every)


def reads_params(n: int, read_len: int, seed: int = 12345, paired: bool = False, n_with_n: int = 0) -> SynthReads:
    return SynthReads(seed ^ 0x5EED5EED, n, read_len, 1 if paired else 0, n_with_n)


def pg_host(g: SynthPg) -> np.ndarray:
    out = np.empty(g.pg_len, dtype=np.uint8)
    lib.pgrc_synth_pg_host(C.byref(g), out.ctypes.data_as(C.c_void_p))
    return out


def reads_host(g: SynthPg, pg_ascii: np.ndarray, rs: SynthReads, first: int = 0, count: int | None = None) -> np.ndarray:
    count = rs.n - first if count is None else count
    out = np.empty((count, rs.read_len), dtype=np.uint8)
    pgp = pg_ascii.ctypes.data_as(C.c_void_p) if pg_ascii is not None else None
    lib.pgrc_synth_reads_host(C.byref(g), pgp, C.byref(rs), first, count, out.ctypes.data_as(C.c_void_p))
    return out


def pg_device(g: SynthPg, dev_ptr: int, stream: int = 0):
    rc = lib.pgrc_synth_pg_device(C.byref(g), C.c_void_p(dev_ptr), C.c_void_p(stream))
    if rc:
        raise RuntimeError(f"pgrc_synth_pg_device failed ({rc})")


def reads_device(g: SynthPg, pg_dev_ptr: int, rs: SynthReads, first: int, count: int, out_dev_ptr: int, stride: int,
                 stream: int = 0):
    rc = lib.pgrc_synth_reads_device(C.byref(g), C.c_void_p(pg_dev_ptr), C.byref(rs), first, count,
                                     C.c_void_p(out_dev_ptr), stride, C.c_void_p(stream))
    if rc:
        raise RuntimeError(f"pgrc_synth_reads_device failed ({rc})")
